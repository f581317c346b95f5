for v in 0 1 0 1; do echo "== CENO_HIP_DENSE_LDS=$v"; CENO_HIP_DENSE_LDS=$v python3 bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.read()); print(json.dumps({'ms_per_step': round(r['ms_per_step'],4), 'stub': round(r['ms_per_step_stub'],4), 'kernel_ms': round(r['roofline']['kernel_ms_per_sumcheck'],4), 'frac': round(r['roofline']['frac'],4)}))"; done
CENO_HIP_DENSE_LDS=1 python3 -m pytest tests/test_gpu_parity.py -x -q -k "nv26 or nv22 or full_size or dense" 2>&1 | tail -3
