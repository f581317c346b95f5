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


class OracleBatchedEngine(cdist.BatchedEngine):
    """round-granular stand-in for the device engine: the message of engine round j is the FIRST message of a
    fresh oracle sumcheck over the tables folded so far (exhausted tables carry their front-load tail product)"""

    def __init__(self, tables, coeffs, terms, max_num_vars, degree):
        self.tabs = [np.ascontiguousarray(np.asarray(t, dtype=np.uint64).reshape(-1, 2)) for t in tables]
        self.pure = [None] * len(self.tabs)
        self.coeffs, self.terms, self.left, self.deg = np.asarray(coeffs, dtype=np.uint64), terms, max_num_vars, degree

    def _bind(self, ch):
        for j, t in enumerate(self.tabs):
            if t.shape[0] > 1:
                self.tabs[j] = po.mle_fix_variable(t, ch)
                if self.tabs[j].shape[0] == 1:
                    self.pure[j] = self.tabs[j][0].copy()
            else:
                if self.pure[j] is None:
                    self.pure[j] = t[0].copy()
                self.tabs[j] = np.array([po.e2_mul((int(t[0, 0]), int(t[0, 1])), (int(ch[0]), int(ch[1])))], dtype=np.uint64)
        self.left -= 1

    def round(self, challenge):
        if challenge is not None:
            self._bind(challenge)
        msgs, _, _ = po.sumcheck_prove(self.tabs, self.coeffs, self.terms, self.left, self.deg, po.StubTranscript(1))
        return msgs[0]

    def finish(self, last_challenge):
        if last_challenge is not None:
            self._bind(last_challenge)
        return np.stack([self.pure[j] if self.pure[j] is not None else t[0] for j, t in enumerate(self.tabs)])


def batched_case(n_total):
    """global description of a mixed-size plan: (num_vars, k tables, terms) per class; tables from SplitMix streams"""
    spec = [(n_total, 3, [[0, 1, 2], [0, 1]]), (n_total - 2, 2, [[0, 1], [1]]), (2, 2, [[0, 1, 1]]), (n_total - 1, 1, [[0]])]
    classes = []
    for ci, (nv, k, terms) in enumerate(spec):
        tabs = [po.fill_splitmix(2 << nv, 0xBA7C + 16 * ci + j, 0).reshape(-1, 2) for j in range(k)]
        coeffs = po.fill_splitmix(2 * len(terms), 0xC0EF + ci, 0).reshape(-1, 2)
        classes.append({"num_vars": nv, "tables": tabs, "terms": terms, "coeffs": coeffs})
    return classes


def main_batched(out_dir, n_total):
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    log_w = world.bit_length() - 1
    classes = []
    for c in batched_case(n_total):
        sharded = c["num_vars"] >= log_w + 1 and c["num_vars"] != 2  # the 2-variable class stays replicated
        if c["num_vars"] == n_total - 1:
            sharded = True
        if sharded:
            m = 1 << (c["num_vars"] - log_w)
            tabs = [t[rank * m:(rank + 1) * m] for t in c["tables"]]
        else:
            tabs = c["tables"]
        classes.append(dict(c, tables=tabs, sharded=sharded))
    factory = lambda tables, coeffs, terms, max_nv, degree: OracleBatchedEngine(tables, coeffs, terms, max_nv, degree)
    msgs, chal, fins = cdist.sharded_batched_sumcheck_prove(factory, classes, n_total, 3, prover.Transcript.stub(0xF5), dist=dist,
                                                          world=world, rank=rank)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), msgs=msgs, chal=chal, fin=np.concatenate(fins))
    dist.barrier()
    dist.destroy_process_group()


def main_shm(out_dir, iters):
    """host shared-memory exchange of the C++ sharded driver (ceno_amd/host/dist.cpp), no GPU involved"""
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    comm = prover.ShmComm(world, rank, dist)
    rc = comm.selftest(iters)
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write(str(rc))
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


class FileRendezvous:
    """the three collectives ShmComm needs from a launcher (segment name broadcast, two barriers), through files of the test's
    temporary directory: the GPU workers of tests/test_gpu_dist.py then start without importing torch (dozens of processes)"""

    def __init__(self, directory, rank, world):
        self.d, self.rank, self.world, self.seq = directory, rank, world, 0

    def all_gather_object(self, out, obj):
        import pickle
        import time

        self.seq += 1
        mine = os.path.join(self.d, f"rv{self.seq}_{self.rank}.pkl")
        with open(mine + ".tmp", "wb") as f:
            pickle.dump(obj, f)
        os.rename(mine + ".tmp", mine)  # atomic: a reader never sees a partial file
        t0 = time.time()
        for r in range(self.world):
            path = os.path.join(self.d, f"rv{self.seq}_{r}.pkl")
            while not os.path.exists(path):
                if time.time() - t0 > 300:
                    raise TimeoutError(f"rank {r} did not reach rendezvous {self.seq}")
                time.sleep(0.002)
            with open(path, "rb") as f:
                out[r] = pickle.load(f)

    def broadcast_object_list(self, objs, src=0):
        got = [None] * self.world
        self.all_gather_object(got, list(objs))
        objs[:] = got[src]

    def barrier(self):
        self.all_gather_object([None] * self.world, 0)

    def destroy_process_group(self):
        pass


def main_shm_gpu(out_dir, n_local):
    """the C++ sharded driver with the shared-memory exchange, `world` PROCESSES sharing GPU 0 (no RCCL on this path,
    so several ranks may use the same device): real HIP engine, real cross-process exchange"""
    from ceno_amd import Device

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist = FileRendezvous(out_dir, rank, world)
    dev = Device(0)
    k = 3
    tables = [po.fill_splitmix(2 << n_local, 0xCE10 + j, rank * 2 * (1 << n_local)).reshape(-1, 2) for j in range(k)]
    mles = [dev.upload(t) for t in tables]
    comm = prover.ShmComm(world, rank, dist)
    stream = dev.stream_create()
    n_total = n_local + world.bit_length() - 1
    msgs, chal, fin = prover.dist_sumcheck_prove(dev, comm, mles, po.ext([1]), [list(range(k))], n_total, k, prover.Transcript.stub(0xF5), stream)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), msgs=msgs, chal=chal, fin=fin)
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


def main_shm_gpu_batched(out_dir, n_total):
    """C++ mixed-size batched sharded driver, `world` processes sharing GPU 0"""
    from ceno_amd import Device

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist = FileRendezvous(out_dir, rank, world)
    log_w = world.bit_length() - 1
    dev = Device(0)
    classes = []
    for c in batched_case(n_total):
        sharded = c["num_vars"] > log_w and c["num_vars"] != 2
        tabs = c["tables"]
        if sharded:
            m = 1 << (c["num_vars"] - log_w)
            tabs = [t[rank * m:(rank + 1) * m] for t in tabs]
        classes.append(dict(num_vars=c["num_vars"], sharded=sharded, mles=[dev.upload(np.ascontiguousarray(t)) for t in tabs],
                            coeffs=c["coeffs"], terms=c["terms"]))
    comm = prover.ShmComm(world, rank, dist)
    stream = dev.stream_create()
    msgs, chal, fins = prover.dist_batched_sumcheck_prove(dev, comm, classes, n_total, 3, prover.Transcript.stub(0xF5), stream)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), msgs=msgs, chal=chal, fin=np.concatenate(fins))
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


def chip_case(log2_n, w=9, shape=(4, 4, 0, 8)):
    """the row-sharded chip proof's test case: columns, record plan, challenges (deterministic; shared by the workers and the test)"""
    num_reads, num_writes, num_lk_tables, num_lk = shape
    n_rec = num_reads + num_writes + num_lk_tables + (num_lk_tables if num_lk_tables else num_lk)
    alpha, beta = (0x1234567, 0x89ABCDE), (0x13579B, 0x2468AC)
    cols = [po.rand_base(1 << log2_n, 700 + j) for j in range(w)]
    b2 = po.e2_mul(beta, beta)
    terms, coeffs, out_terms = [], [], []
    for k in range(n_rec):
        base = len(terms)
        terms += [[(2 * k) % w], [(2 * k + 1) % w], [(3 * k + 5) % w, (k + 7) % w]]
        coeffs += [beta, b2, alpha]
        out_terms.append([base, base + 1, base + 2])
    return cols, po.ext(coeffs), terms, out_terms, [alpha, beta], shape


def main_shm_gpu_chip(out_dir, log2_n):
    """ceno_dist_create_chip_proof with the shared-memory exchange: `world` PROCESSES sharing GPU 0, every rank its rows of every column"""
    from ceno_amd import Device

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    q = int(os.environ.get("CENO_TEST_ROW_BLOCK_LOG", "3"))
    dist = FileRendezvous(out_dir, rank, world)
    dev = Device(0)
    cols, coeffs, terms, out_terms, challenges, shape = chip_case(log2_n)
    local = [dev.upload(prover.shard_rows(c, world, rank, q)) for c in cols]
    comm = prover.ShmComm(world, rank, dist)
    stream = dev.stream_create()
    task = dict(mles=local, n_witin=len(cols), n_fixed=0, n_structural=0, num_instances=(1 << log2_n) - 5, log2_num_instances=log2_n - (world.bit_length() - 1),
                num_reads=shape[0], num_writes=shape[1], num_lk_tables=shape[2], num_lk=shape[3], record_coeffs=coeffs, record_terms=terms,
                record_out_terms=out_terms)
    pr = prover.dist_create_chip_proof(dev, comm.h, task, log2_n, q, challenges, prover.Transcript.stub(21), stream)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), msgs=pr.tower_msgs, point=pr.tower_point, prod=pr.tower_prod_evals, logup=pr.tower_logup_evals,
             r_out=pr.r_out_evals, w_out=pr.w_out_evals, lk_out=pr.lk_out_evals, rt_main=pr.rt_main)
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


def main():
    if len(sys.argv) > 3 and sys.argv[3] == "shm_gpu_chip":
        return main_shm_gpu_chip(sys.argv[1], int(sys.argv[2]))
    if len(sys.argv) > 3 and sys.argv[3] == "shm_gpu":
        return main_shm_gpu(sys.argv[1], int(sys.argv[2]))
    if len(sys.argv) > 3 and sys.argv[3] == "shm_gpu_batched":
        return main_shm_gpu_batched(sys.argv[1], int(sys.argv[2]))
    import torch.distributed as dist

    if len(sys.argv) > 3 and sys.argv[3] == "batched":
        return main_batched(sys.argv[1], int(sys.argv[2]))
    if len(sys.argv) > 3 and sys.argv[3] == "shm":
        return main_shm(sys.argv[1], int(sys.argv[2]))
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
