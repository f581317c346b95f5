# the multi-rank opening on 8 virtual ranks of ONE device, repeated: pipelined rounds (CENO_DIST_OPEN_PIPELINE=1) against serial rounds (the default
# where ranks may share a device).  A stalled run shows as a wall time of minutes (queued round kernels give up after CENO_HIP_PIPE_TIMEOUT_S).
export CENO_HIP_PIPE_TIMEOUT_S=8
for mode in 1 0; do
  bad=0; slow=0
  for i in $(seq 1 ${REPS:-25}); do
    t0=$(date +%s.%N)
    CENO_DIST_OPEN_PIPELINE=$mode timeout 300 python3 -m pytest tests/test_gpu_dist_gkr.py -x -q -k "opening_equals and 8-" > /tmp/ors.log 2>&1 || bad=$((bad+1))
    t1=$(date +%s.%N)
    if [ $(python3 -c "print(int($t1-$t0 > 15))") = 1 ]; then slow=$((slow+1)); fi
  done
  echo "CENO_DIST_OPEN_PIPELINE=$mode: ${REPS:-25} runs, $bad failed, $slow slower than 15 s"
done
