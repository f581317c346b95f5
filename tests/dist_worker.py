"""Worker for tests/test_dist_cpu.py: world_size-N gloo run of the hypercube-sharded sumcheck driver
(ceno_amd/dist.py) with the CPU oracle standing in for the per-shard device engine (test infrastructure:
the product engine is HipShardEngine)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ceno_amd import dist as cdist  # noqa: E402
from ceno_amd import prover  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


class OracleShardEngine(cdist.ShardEngine):
    def __init__(self, tables):
        self.tables = [np.ascontiguousarray(t) for t in tables]
        self.k = len(tables)

    def _first_msg(self, tabs, degree):
        nv = int(tabs[0].shape[0]).bit_length() - 1
        msgs, _, _ = po.sumcheck_prove(tabs, po.ext([1]), [list(range(self.k))], nv, degree, po.StubTranscript(1))
        return msgs[0]

    def begin(self, n_local, degree):
        return {"tabs": list(self.tables), "deg": degree}

    def round_partial(self, state, challenge):
        if challenge is not None:
            state["tabs"] = [po.mle_fix_variable(t, challenge) for t in state["tabs"]]
        return self._first_msg(state["tabs"], state["deg"])

    def finish(self, state, last_challenge):
        if last_challenge is not None:
            state["tabs"] = [po.mle_fix_variable(t, last_challenge) for t in state["tabs"]]
        return np.stack([t[0] for t in state["tabs"]])

    def tail(self, tables, degree, transcript, msgs_out, chal_out, first_round):
        tabs = list(tables)
        nv = int(tabs[0].shape[0]).bit_length() - 1
        ch = None
        for r in range(nv):
            if ch is not None:
                tabs = [po.mle_fix_variable(t, ch) for t in tabs]
            msg = self._first_msg(tabs, degree)
            ch = cdist._absorb_round(transcript, msg)
            msgs_out[first_round + r] = msg
            chal_out[first_round + r] = ch
        tabs = [po.mle_fix_variable(t, ch) for t in tabs]
        return np.stack([t[0] for t in tabs])


def main():
    import torch.distributed as dist

    out_dir = sys.argv[1]
    n_local = int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    k = 3
    # shard `rank` of table j = words [rank * 2 * 2^n_local, ...) of the SplitMix stream (same rule as bench.py)
    tables = [po.fill_splitmix(2 << n_local, 0xCE10 + j, rank * 2 * (1 << n_local)).reshape(-1, 2) for j in range(k)]
    eng = OracleShardEngine(tables)
    n_total = n_local + world.bit_length() - 1
    msgs, chal, fin = cdist.sharded_sumcheck_prove(eng, n_total, k, prover.Transcript.stub(0xF5), dist=dist, world=world, rank=rank)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), msgs=msgs, chal=chal, fin=fin)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
