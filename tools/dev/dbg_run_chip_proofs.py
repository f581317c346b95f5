import sys, os, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
faulthandler.dump_traceback_later(60, exit=True)
from ceno_amd import Device, prover, synthetic
dev = Device(0)
flow = synthetic.ShardFlowWide(dev, prover, log_cycles=int(os.environ.get("LOG_CYCLES", "10")), n_queries=8, pow_bits=4)
kind = os.environ.get("TR", "stub")
mk = (lambda: prover.Transcript.stub(0x5A)) if kind == "stub" else (lambda: prover.Transcript.poseidon2(b"riscv"))
fk = (lambda: prover.Transcript.stub(0xF0)) if kind == "stub" else (lambda: prover.Transcript.poseidon2(b"fork"))
for rep in range(int(os.environ.get("REPS", "2"))):
    try:
        r = flow.run(mk, fk, lanes=int(os.environ.get("LANES", "4")))
        print("ok", r["total_ms"], flush=True)
    except Exception as e:
        print("ERR", e, flush=True)
