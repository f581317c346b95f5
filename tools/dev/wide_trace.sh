cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for m in "$@"; do
  o=gpurun_out/wt_$m; rm -rf $o
  CENO_HIP_GEN_SPLIT_MLES=$m timeout 300 rocprofv3 --kernel-trace --output-format csv -d $o -- python3 tools/bench_batched_wide.py --reps 2 > $o.log 2>&1
  echo "== cap $m"; tail -1 $o.log | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms'])"
  python3 - $o <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=[x for x in csv.DictReader(open(f)) if 'k_gen_eq' in x['Kernel_Name'] or 'k_eq_base0' in x['Kernel_Name']]
rows.sort(key=lambda x:int(x['Start_Timestamp']))
n=len(rows)//2
for x in rows[n:n+6]:
    print(x['Kernel_Name'][:30], (int(x['End_Timestamp'])-int(x['Start_Timestamp']))/1e3,'us wgs',int(x['Grid_Size_X'])//256,'vgpr',x['VGPR_Count'],'lds',x.get('LDS_Block_Size'))
PY
done
