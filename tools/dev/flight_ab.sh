set -x
FLIGHT_PREFILL=1 python tools/dev/flight_ab.py 26 poseidon2 2>&1 | grep -v PARITY
FLIGHT_PREFILL=1 CENO_HIP_POOL_TRACE=1 python tools/dev/flight_ab.py 26 poseidon2 2>&1 | grep -v PARITY | grep -c hipMalloc
FLIGHT_PREFILL=1 CENO_HIP_HOST_TIMING=1 python tools/dev/flight_ab.py 26 poseidon2 2>&1 | grep -v PARITY | tail -40
