"""Worker for tests/test_dist_cpu.py: world_size-N gloo run of the hypercube-sharded sumcheck driver
(ceno_amd/dist.py) with the CPU oracle standing in for the per-shard device engine (test infrastructure:
the product engine is HipShardEngine)."""
import os
import sys
import time

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
    if n_local >= 16:  # at size: shard `rank` of the SplitMix stream generated on the device (the same words as fill_splitmix at this offset)
        mles = [dev.synthetic(n_local, True, 0xCE10 + j, word_offset=rank * 2 * (1 << n_local)) for j in range(k)]
        dev.sync()
    else:
        tables = [po.fill_splitmix(2 << n_local, 0xCE10 + j, rank * 2 * (1 << n_local)).reshape(-1, 2) for j in range(k)]
        mles = [dev.upload(t) for t in tables]
    comm = prover.ShmComm(world, rank, dist)
    stream = dev.stream_create()
    n_total = n_local + world.bit_length() - 1
    reps = int(os.environ.get("CENO_TEST_DIST_REPS", "1"))
    walls = []
    for _ in range(reps):
        dist.barrier()
        t0 = time.time()
        msgs, chal, fin = prover.dist_sumcheck_prove(dev, comm, mles, po.ext([1]), [list(range(k))], n_total, k, prover.Transcript.stub(0xF5), stream)
        walls.append(time.time() - t0)
    import ctypes as C

    L = prover.plib()
    L.ceno_dist_comm_stats.restype = C.c_int
    L.ceno_dist_comm_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
    st4 = (C.c_uint64 * 4)()
    L.ceno_dist_comm_stats(comm.h, st4, 0)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), msgs=msgs, chal=chal, fin=fin, wall_s=np.array(walls), wire=np.array(list(st4), dtype=np.uint64) // reps)
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


def chip_main_job(tables, log2_n, w, rt_main):
    """the main-constraint job of the process test's chip: products of two and three columns and every column alone under one Prefix selector at
    the chip proof's rt_main (`tables`: the device tables — whole, or one rank's rows)"""
    terms = [[w, j, (j + 1) % w] for j in range(w)] + [[w, j, (j + 2) % w, (j + 4) % w] for j in range(0, w, 2)] + [[w, j] for j in range(w)]
    scalars = [[((3 + 5 * t, 11 * t + 1), [2 + (t % 2)])] for t in range(len(terms))]
    sel = (po.SEL_PREFIX, 0, (1 << log2_n) - 5, 0, (), 0, np.ascontiguousarray(rt_main))
    return dict(num_vars=log2_n, mles=list(tables) + [None], n_witin=w, n_fixed=0, n_structural=1, selectors=[sel], n_exprs=2, max_degree=4, terms=terms,
                scalars=scalars)


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
    tr = prover.Transcript.stub(21)
    pr = prover.dist_create_chip_proof(dev, comm.h, task, log2_n, q, challenges, tr, stream)
    # ... and the chip's main-constraint sumcheck on the same row shards, the same transcript (ceno_dist_prove_batched_main_constraints)
    mjob = chip_main_job(local, log2_n, len(cols), pr.rt_main)
    mc, mm, mrt, mev = prover.dist_prove_batched_main_constraints(dev, comm.h, [mjob], challenges, tr, q, stream)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), msgs=pr.tower_msgs, point=pr.tower_point, prod=pr.tower_prod_evals, logup=pr.tower_logup_evals,
             r_out=pr.r_out_evals, w_out=pr.w_out_evals, lk_out=pr.lk_out_evals, rt_main=pr.rt_main, main_claim=np.array(mc, dtype=np.uint64), main_msgs=mm,
             main_rt=mrt, main_evals=mev)
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


def open_case(world):
    """matrices of a commitment for the process test of the multi-rank opening: (heights, column split per matrix and rank, full matrices)"""
    heights = [9, 7, 9] if world <= 2 else [10, 8, 10]
    col_split = [[1 + ((m + g) % 2) for g in range(world)] for m in range(len(heights))]
    fulls = [po.rand_base((1 << h) * sum(ws), 2100 + 5 * i).reshape(1 << h, sum(ws)) for i, (h, ws) in enumerate(zip(heights, col_split))]
    return heights, col_split, fulls


def main_shm_gpu_open(out_dir, _n):
    """commit across `world` PROCESSES sharing GPU 0 (ceno_dist_commit_traces_mmcs) and open across them (ceno_dist_basefold_open_mmcs), every
    exchange through the shared segment (no RCCL between ranks of one device): root and proof written per rank"""
    import torch

    from ceno_amd import Device
    from ceno_amd import dist as cdist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist = FileRendezvous(out_dir, rank, world)
    dev = Device(0)
    heights, col_split, fulls = open_case(world)
    comm = prover.ShmComm(world, rank, dist)
    stream = dev.stream_create()
    keep, ptrs = [], []
    for ws, full in zip(col_split, fulls):
        c0 = sum(ws[:rank])
        cols = np.ascontiguousarray(full[:, c0:c0 + ws[rank]].T)
        t = torch.from_numpy(cols.view(np.int64).copy()).to("cuda:0")
        keep.append(t)
        ptrs.append(t.data_ptr())
    torch.cuda.synchronize()
    com = cdist.sharded_commit_mmcs_native(dev, comm.h, ptrs, col_split, heights, 1, rank, stream)
    dev.sync(stream)
    point = np.array([[(i * 7919 + 13) % po.P, (i * 104729 + 17) % po.P] for i in range(max(heights))], dtype=np.uint64)
    points = [point[:h] for h in heights]
    evals = [np.array([po.mle_evaluate(np.ascontiguousarray(full[:, c]), points[m]) for c in range(full.shape[1])], dtype=np.uint64)
             for m, full in enumerate(fulls)]
    proof = prover.dist_basefold_open(dev, comm.h, heights, col_split, 1, ptrs, [t.data_ptr() for t in com["codeword_rows"]], com["subtree"], com["top"],
                                      points, evals, 10, 3, prover.Transcript.poseidon2(b"open"), stream)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), root=np.asarray(com["root"]), proof=proof)
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


# ---- the row-sharded tower proof a SECOND time, in Python over the oracle's primitives and torch.distributed (gloo): an independent
# statement of the algorithm of ceno_amd/host/dist_gkr.cpp (block-cyclic rows, local towers = shards of the large layers, replicated tops,
# local rounds with exchanged partial sums, interleaved gather, replicated tail).  Test infrastructure: everything arithmetic is the oracle's. ----
def _usize(tr, v):
    tr.append_label(int(v).to_bytes(8, "little"))


def _pows(tr, n):
    tr.append_label(b"combine subset evals")
    a = tr.sample_ext()
    out, acc = [], (1, 0)
    for _ in range(n):
        out.append(acc)
        acc = po.e2_mul(acc, a)
    return out


def _plan(n_prod_active, n_logup_active, a_prod, a_num, a_den):
    """tables [eq, (a, b) per product tower, (p1, p2, q1, q2) per LogUp tower] -> monomial plan of the layer sumcheck"""
    coeffs, terms, t = [], [], 1
    for i in range(n_prod_active):
        coeffs.append(a_prod[i]); terms.append([0, t, t + 1]); t += 2
    for i in range(n_logup_active):
        coeffs += [a_num[i], a_num[i], a_den[i]]
        terms += [[0, t, t + 3], [0, t + 1, t + 2], [0, t + 2, t + 3]]
        t += 4
    return po.ext(coeffs), terms


def _first_msg(tabs, coeffs, terms):
    nv = int(tabs[0].shape[0]).bit_length() - 1
    msgs, _, _ = po.sumcheck_prove(tabs, coeffs, terms, nv, 3, po.StubTranscript(1))
    return msgs[0]


def _rounds(tabs, coeffs, terms, n_rounds, tr, msgs_out, chal_out, combine=None):
    """n_rounds rounds on `tabs` (folded in place); combine: sums a partial message over the ranks"""
    for _ in range(n_rounds):
        m = _first_msg(tabs, coeffs, terms)
        if combine is not None:
            m = combine(m)
        for e in range(3):
            tr.append_ext((int(m[e][0]), int(m[e][1])))
        tr.append_label(b"Internal round")
        ch = tr.sample_ext()
        msgs_out.append(m)
        chal_out.append(ch)
        for j in range(len(tabs)):
            tabs[j] = po.mle_fix_variable(tabs[j], ch)


def _interleave(parts, lo_bits, k):
    """per-rank tables -> the global table: index (hi, rank, lo)"""
    world, n_loc = len(parts), parts[0].shape[0]
    out = np.zeros((n_loc * world, 2), dtype=np.uint64)
    j = np.arange(n_loc)
    lo, hi = j & ((1 << lo_bits) - 1), j >> lo_bits
    for g in range(world):
        out[(hi << (k + lo_bits)) | (g << lo_bits) | lo] = parts[g]
    return out


def rotation_case(n, log2):
    """full columns, rotation pairs, subgroup size and the point of the rotation argument for the CPU second implementation"""
    src = po.rand_base(1 << n, 41)
    cols = [src, po.rotation_next_base_mle(src, log2), po.rand_base(1 << n, 42), po.rand_base(1 << n, 43)]
    return cols, [(0, 1), (2, 3)], 23 if log2 == 5 else 45, po.rand_ext(n, 44)


def main_rotation_gloo(out_dir, n):
    """the rotation argument over ROW-SHARDED columns a second time (independent of prover.cpp prover_prove_rotation_sharded), in Python over the
    oracle's primitives and gloo: local rotations, the local selector at the point without the rank coordinates, q local rounds with the two
    partial evaluations summed over the ranks, the folded tables gathered with the rank bits lowest, the tail replicated, the left evaluations
    as eq-weighted sums of per-rank evaluations.  Test infrastructure: every piece of arithmetic is the oracle's."""
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    k, q, log2 = world.bit_length() - 1, int(os.environ.get("CENO_TEST_ROW_BLOCK_LOG", "5")), int(os.environ.get("CENO_TEST_ROT_LOG", "5"))
    cols, pairs, subgroup, rt = rotation_case(n, log2)
    rt = [(int(x[0]), int(x[1])) for x in rt]
    local = [prover.shard_rows(c, world, rank, q) for c in cols]
    one = (1, 0)

    def gather(x):
        got = [None] * world
        dist.all_gather_object(got, x)
        return got

    def eq_rank(pt):
        v = one
        for j in range(k):
            c_ = pt[q + j]
            v = po.e2_mul(v, c_ if (rank >> j) & 1 else po.e2_sub(one, c_))
        return v

    def without_rank(pt):
        return [pt[j] for j in range(len(pt)) if j < q or j >= q + k]

    tr = po.StubTranscript(8)
    alphas = _pows(tr, len(pairs))
    eq_g = eq_rank(rt)
    sel = po.rotation_selector(po.build_eq(po.ext(without_rank(rt))), subgroup, log2)
    tabs, cf_loc, cf_glob, terms = [], [], [], []
    for j, (s_, t_) in enumerate(pairs):
        tabs += [po.rotation_next_base_mle(local[s_], log2), local[t_]]
        na = po.e2_sub((0, 0), alphas[j])
        cf_glob += [alphas[j], na]
        cf_loc += [po.e2_mul(alphas[j], eq_g), po.e2_mul(na, eq_g)]
        terms += [[2 * j, 2 * len(pairs)], [2 * j + 1, 2 * len(pairs)]]
    tabs.append(sel)
    _usize(tr, n)
    _usize(tr, 2)
    msgs, origin = [], []

    def first(tabs_, cf_):
        nv = int(tabs_[0].shape[0]).bit_length() - 1
        m, _, _ = po.sumcheck_prove(tabs_, po.ext(cf_), terms, nv, 2, po.StubTranscript(1))
        return [(int(m[0][e][0]), int(m[0][e][1])) for e in range(2)]

    def publish(m):
        for e in range(2):
            tr.append_ext(m[e])
        tr.append_label(b"Internal round")
        ch = tr.sample_ext()
        msgs.append(m)
        origin.append(ch)
        return ch

    for _ in range(q):
        tot = [(0, 0), (0, 0)]
        for part in gather(first(tabs, cf_loc)):
            tot = [po.e2_add(tot[e], part[e]) for e in range(2)]
        ch = publish(tot)
        tabs = [po.mle_fix_variable(t_, ch) for t_ in tabs]
    # the selector's rank factor went into the coefficients: the global selector carries it
    tabs[-1] = po.ext([po.e2_mul((int(v[0]), int(v[1])), eq_g) for v in tabs[-1]])
    tabs = [_interleave(gather(np.ascontiguousarray(t_)), 0, k) for t_ in tabs]
    for _ in range(n - q):
        ch = publish(first(tabs, cf_glob))
        tabs = [po.mle_fix_variable(t_, ch) for t_ in tabs]
    fin = [(int(t_[0][0]), int(t_[0][1])) for t_ in tabs]
    left, right = po.rotation_points(po.ext(origin), log2)
    left_l = [(int(x[0]), int(x[1])) for x in left]
    eq_left = eq_rank(left_l)
    rk = origin[log2 - 1]
    rk_inv = po.e2_inv(rk)
    evals = []
    parts = gather([po.e2_mul(po.mle_evaluate(local[s_], po.ext(without_rank(left_l))), eq_left) for (s_, _t) in pairs])
    for j in range(len(pairs)):
        lv = (0, 0)
        for g in range(world):
            lv = po.e2_add(lv, parts[g][j])
        rot, target = fin[2 * j], fin[2 * j + 1]
        rv = po.e2_mul(po.e2_sub(rot, po.e2_mul(po.e2_sub(one, rk), lv)), rk_inv)
        evals += [lv, rv, target]
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), msgs=po.ext([m[e] for m in msgs for e in range(2)]).reshape(n, 2, 2), evals=po.ext(evals),
             origin=po.ext(origin), left=left, right=right)
    dist.barrier()
    dist.destroy_process_group()


def main_case(world):
    """chips of a batched main-constraint sumcheck for the CPU second implementation of its row-sharded form: per chip (num_vars, columns, Prefix
    range, terms over the columns 0 .. w - 1 and the selector w, coefficients), and the selector points"""
    w, nvs = 4, ((8, 6, 3) if world <= 2 else (9, 7, 4))  # (the last chip is too small to be sharded at q = 3: it rides along replicated)
    chips = []
    for c, nv in enumerate(nvs):
        cols = [po.rand_base(1 << nv, 5100 + 11 * c + j) for j in range(w)]
        terms = [[w, 0, 1], [w, 1, 2, 3], [w, 2], [w, 3], [w, 0]]
        coeffs = [(7 + 3 * t + c, 2 + t) for t in range(len(terms))]
        off = min(3 * c, (1 << nv) // 4)
        chips.append(dict(nv=nv, cols=cols, off=off, n=max(1, (1 << nv) - 7 - 5 * c) if nv >= 6 else (1 << nv) - 1 - off, terms=terms, coeffs=coeffs,
                          point=po.rand_ext(nv, 5200 + c)))
    return chips


def main_sharded_gloo(out_dir, _n):
    """prove_batched_main_constraints' sumcheck over ROW-SHARDED tables a second time (independent of main_constraints.cpp
    prover_main_constraints_sharded), in Python over the oracle's primitives and gloo: a rank's Prefix selector is eq at the point without the rank
    coordinates on the rank's part of the range, its eq factor rides on the coefficients; q local rounds with the partial evaluations summed over
    the ranks; every table gathered with the rank bits lowest; the tail replicated (a table that has run out of variables is multiplied by
    every further challenge: the front-load rule).  Test infrastructure: every piece of arithmetic is the oracle's."""
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    k, q, D = world.bit_length() - 1, int(os.environ.get("CENO_TEST_ROW_BLOCK_LOG", "3")), 4
    chips = main_case(world)
    one = (1, 0)

    def gather(x):
        got = [None] * world
        dist.all_gather_object(got, x)
        return got

    def local_rows(t):
        lo, g, hi = t & ((1 << q) - 1), (t >> q) & (world - 1), t >> (q + k)
        return (hi << q) + ((1 << q) if rank < g else (lo if rank == g else 0))

    tabs, nv_of, cf_loc, cf_glob, terms, sel_scale = [], [], [], [], [], []
    s_tabs, s_cf, s_terms = [], [], []  # the chips that are too small to be sharded: whole tables, the same on every rank
    for ch in chips:
        if ch["nv"] - k < q + 1:
            start = len(s_tabs)
            s_tabs += list(ch["cols"]) + [po.selector_compute(po.SEL_PREFIX, ch["point"], ch["off"], ch["n"])]
            s_terms += [[start + j for j in t_] for t_ in ch["terms"]]
            s_cf += list(ch["coeffs"])
            continue
        pt = [(int(x[0]), int(x[1])) for x in ch["point"]]
        eq_g = one
        for j in range(k):
            eq_g = po.e2_mul(eq_g, pt[q + j] if (rank >> j) & 1 else po.e2_sub(one, pt[q + j]))
        pt_loc = po.ext([pt[j] for j in range(ch["nv"]) if j < q or j >= q + k])
        lo, hi = local_rows(ch["off"]), local_rows(ch["off"] + ch["n"])
        start = len(tabs)
        tabs += [prover.shard_rows(c_, world, rank, q) for c_ in ch["cols"]] + [po.selector_compute(po.SEL_PREFIX, pt_loc, lo, hi - lo)]
        sel_scale += [one] * len(ch["cols"]) + [eq_g]
        nv_of += [ch["nv"]] * (len(ch["cols"]) + 1)
        for t_, c_ in zip(ch["terms"], ch["coeffs"]):
            terms.append([start + j for j in t_])
            cf_glob.append(c_)
            cf_loc.append(po.e2_mul(c_, eq_g))
    max_nv = max(c_["nv"] for c_ in chips)
    tr = po.StubTranscript(5)
    _usize(tr, max_nv)
    _usize(tr, D)
    msgs, rt = [], []

    def first(tabs_, cf_, remaining, terms_=None):
        m, _, _ = po.sumcheck_prove(tabs_, po.ext(cf_), terms if terms_ is None else terms_, remaining, D, po.StubTranscript(1))
        return [(int(m[0][e][0]), int(m[0][e][1])) for e in range(D)]

    s_evals = [None] * len(s_tabs)

    def small_part(i):  # the replicated chips' part of round i (added once) ...
        return first(s_tabs, s_cf, max_nv - i, s_terms) if s_tabs else [(0, 0)] * D

    def small_fold(ch_):  # ... and their tables after it: a fold, or — out of variables — the front-load rule
        for j, t_ in enumerate(s_tabs):
            if t_.shape[0] > 1:
                t_ = po.mle_fix_variable(t_, ch_)
                if t_.shape[0] == 1:
                    s_evals[j] = (int(t_[0][0]), int(t_[0][1]))
            else:
                if t_.ndim == 1:
                    t_ = po.ext([(int(t_[0]), 0)])
                t_ = po.ext([po.e2_mul((int(t_[0][0]), int(t_[0][1])), ch_)])
            s_tabs[j] = t_

    def publish(m):
        for e in range(D):
            tr.append_ext(m[e])
        tr.append_label(b"Internal round")
        ch_ = tr.sample_ext()
        msgs.append(m)
        rt.append(ch_)
        return ch_

    for i in range(q):  # every chip still has local variables: plain folds
        tot = small_part(i)
        for part in gather(first(tabs, cf_loc, max_nv - k - i)):
            tot = [po.e2_add(tot[e], part[e]) for e in range(D)]
        ch_ = publish(tot)
        tabs = [po.mle_fix_variable(t_, ch_) for t_ in tabs]
        small_fold(ch_)
    tabs = [po.ext([po.e2_mul((int(v[0]), int(v[1])), sc) for v in t_]) if sc != one else t_ for t_, sc in zip(tabs, sel_scale)]
    tabs = [_interleave(gather(np.ascontiguousarray(t_)), 0, k) for t_ in tabs]
    evals = [None] * len(tabs)
    for i in range(q, max_nv):
        sm = small_part(i)
        ch_ = publish([po.e2_add(a_, b_) for a_, b_ in zip(first(tabs, cf_glob, max_nv - i), sm)])
        small_fold(ch_)
        nxt = []
        for j, t_ in enumerate(tabs):
            if t_.shape[0] > 1:
                t_ = po.mle_fix_variable(t_, ch_)
                if t_.shape[0] == 1:
                    evals[j] = (int(t_[0][0]), int(t_[0][1]))  # the table's evaluation at its prefix of the point
            else:  # out of variables: f * x_i at x_i = r (the front-load rule)
                t_ = po.ext([po.e2_mul((int(t_[0][0]), int(t_[0][1])), ch_)])
            nxt.append(t_)
        tabs = nxt
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), msgs=po.ext([m[e] for m in msgs for e in range(D)]).reshape(max_nv, D, 2), rt=po.ext(rt),
             evals=po.ext(evals + s_evals))  # (the sharded chips come first in main_case)
    dist.barrier()
    dist.destroy_process_group()


def main_chip_gloo(out_dir, log2_n):
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    k, q = world.bit_length() - 1, int(os.environ.get("CENO_TEST_ROW_BLOCK_LOG", "2"))
    cols, coeffs, terms, out_terms, (alpha, beta), shape = chip_case(log2_n, w=6, shape=(2, 3, 0, 4))
    n_loc, rows_loc = log2_n - k, 1 << (log2_n - k)
    local = [prover.shard_rows(c, world, rank, q) for c in cols]
    recs = [po.wit_infer(local, coeffs[ts[0]: ts[-1] + 1], [terms[t] for t in ts], n_loc) for ts in out_terms]
    nr, nw = shape[0], shape[1]
    groups = [recs[:nr], recs[nr: nr + nw], recs[nr + nw:]]

    def gather(x):
        got = [None] * world
        dist.all_gather_object(got, x)
        return got

    towers = []  # dict(layers (local), limbs, s, nv)
    for gi, grp in enumerate(groups):
        c = (max(1, len(grp)) - 1).bit_length()
        if gi < 2:
            limbs = po.interleaving_mles_to_mles(grp, rows_loc, 2, (1, 0))
            layers = po.infer_tower_product_witness(int(limbs[0].shape[0]).bit_length(), limbs)
            towers.append(dict(layers=layers, limbs=2, s=q + c, nv=len(layers) + k))
        else:
            ql = po.interleaving_mles_to_mles(grp, rows_loc, 2, alpha)
            layers = po.infer_tower_logup_witness(None, ql)
            towers.append(dict(layers=layers, limbs=4, s=q + c, nv=len(layers) + k))
    n_prod, n_logup = 2, 1
    s_min, s_max = min(t["s"] for t in towers), max(t["s"] for t in towers)
    r_rep = s_max + k
    for t in towers:  # the replicated top: global layer G, gathered and interleaved, and everything above it
        G = min(r_rep, t["nv"] - 1)
        glob = [_interleave(gather(np.ascontiguousarray(t["layers"][G - k][b])), t["s"], k) for b in range(t["limbs"])]
        t["top"] = po.infer_tower_product_witness(G + 1, glob) if t["limbs"] == 2 else po.infer_tower_logup_witness(glob[:2], glob[2:])
    tr = po.StubTranscript(21)
    for t in towers:
        for b in range(t["limbs"]):
            tr.append_ext((int(t["top"][0][b][0][0]), int(t["top"][0][b][0][1])))
    max_nv = max(t["nv"] for t in towers)
    n_alpha = n_prod + 2 * n_logup
    alphas = _pows(tr, n_alpha)
    tr.append_label(b"product_sum")
    out_rt = [tr.sample_ext()]
    all_msgs, prod_evals, logup_evals = [], np.zeros((n_prod, max_nv - 1, 2, 2), dtype=np.uint64), np.zeros((n_logup, max_nv - 1, 4, 2), dtype=np.uint64)
    for rnd in range(1, max_nv):
        active = [i for i, t in enumerate(towers) if t["nv"] > rnd]
        a_prod = [alphas[i] for i in active if i < n_prod]
        a_num = [alphas[n_prod + 2 * (i - n_prod)] for i in active if i >= n_prod]
        a_den = [alphas[n_prod + 2 * (i - n_prod) + 1] for i in active if i >= n_prod]
        cf, tm = _plan(len(a_prod), len(a_num), a_prod, a_num, a_den)
        msgs, chal = [], []
        _usize(tr, rnd)
        _usize(tr, 3)
        if rnd <= r_rep:
            tabs = [po.build_eq(po.ext(out_rt[:rnd]))]
            for i in active:
                tabs += [np.ascontiguousarray(towers[i]["top"][rnd][b]) for b in range(towers[i]["limbs"])]
            _rounds(tabs, cf, tm, rnd, tr, msgs, chal)
        else:
            # one local sumcheck per group of towers with the same shard position
            by_s = {}
            for i in active:
                by_s.setdefault(towers[i]["s"], []).append(i)
            engines = []
            for s_t, members in sorted(by_s.items()):
                rt_loc = [out_rt[j] for j in range(rnd) if j < s_t or j >= s_t + k]
                eq_g = (1, 0)
                for j in range(k):
                    c_ = out_rt[s_t + j]
                    eq_g = po.e2_mul(eq_g, c_ if (rank >> j) & 1 else po.e2_sub((1, 0), c_))
                tabs = [po.build_eq(po.ext(rt_loc))]
                ap, an, ad = [], [], []
                for i in members:
                    tabs += [np.ascontiguousarray(towers[i]["layers"][rnd - k][b]) for b in range(towers[i]["limbs"])]
                    if i < n_prod:
                        ap.append(po.e2_mul(alphas[i], eq_g))
                    else:
                        an.append(po.e2_mul(alphas[n_prod + 2 * (i - n_prod)], eq_g))
                        ad.append(po.e2_mul(alphas[n_prod + 2 * (i - n_prod) + 1], eq_g))
                c2, t2 = _plan(len(ap), len(an), ap, an, ad)
                engines.append(dict(s=s_t, members=members, tabs=tabs, coeffs=c2, terms=t2, eq_g=eq_g))
            for _ in range(s_min):
                part = np.zeros((3, 2), dtype=np.uint64)
                for E in engines:
                    m = _first_msg(E["tabs"], E["coeffs"], E["terms"])
                    for e in range(3):
                        part[e] = po.ext([po.e2_add((int(part[e][0]), int(part[e][1])), (int(m[e][0]), int(m[e][1])))])[0]
                tot = np.zeros((3, 2), dtype=np.uint64)
                for p_ in gather(part):
                    for e in range(3):
                        tot[e] = po.ext([po.e2_add((int(tot[e][0]), int(tot[e][1])), (int(p_[e][0]), int(p_[e][1])))])[0]
                for e in range(3):
                    tr.append_ext((int(tot[e][0]), int(tot[e][1])))
                tr.append_label(b"Internal round")
                ch = tr.sample_ext()
                msgs.append(tot)
                chal.append(ch)
                for E in engines:
                    E["tabs"] = [po.mle_fix_variable(t_, ch) for t_ in E["tabs"]]
            # gather the folded tables, interleave, finish replicated
            where = {}
            for E in engines:
                cur = 1
                for i in E["members"]:
                    where[i] = (E, cur)
                    cur += towers[i]["limbs"]
            E0 = engines[0]
            eq_scaled = np.array([po.e2_mul((int(v[0]), int(v[1])), E0["eq_g"]) for v in E0["tabs"][0]], dtype=np.uint64).reshape(-1, 2)
            tabs = [_interleave(gather(eq_scaled), E0["s"] - s_min, k)]
            for i in active:
                E, cur = where[i]
                tabs += [_interleave(gather(np.ascontiguousarray(E["tabs"][cur + b])), E["s"] - s_min, k) for b in range(towers[i]["limbs"])]
            _rounds(tabs, cf, tm, rnd - s_min, tr, msgs, chal)
        all_msgs += [np.asarray(m, dtype=np.uint64) for m in msgs]
        fin = [(int(t_[0][0]), int(t_[0][1])) for t_ in tabs]
        cur = 1
        for i in active:
            for b in range(towers[i]["limbs"]):
                tr.append_ext(fin[cur + b])
                if i < n_prod:
                    prod_evals[i, rnd - 1, b] = fin[cur + b]
                else:
                    logup_evals[i - n_prod, rnd - 1, b] = fin[cur + b]
            cur += towers[i]["limbs"]
        tr.append_label(b"merge")
        r_merge = tr.sample_ext()
        out_rt = list(chal) + [r_merge]
        alphas = _pows(tr, n_alpha)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), msgs=np.concatenate([m.reshape(-1) for m in all_msgs]), point=po.ext(out_rt[:max_nv]),
             prod=prod_evals, logup=logup_evals)
    dist.barrier()
    dist.destroy_process_group()


def main():
    if len(sys.argv) > 3 and sys.argv[3] == "main_gloo":
        return main_sharded_gloo(sys.argv[1], int(sys.argv[2]))
    if len(sys.argv) > 3 and sys.argv[3] == "rotation_gloo":
        return main_rotation_gloo(sys.argv[1], int(sys.argv[2]))
    if len(sys.argv) > 3 and sys.argv[3] == "chip_gloo":
        return main_chip_gloo(sys.argv[1], int(sys.argv[2]))
    if len(sys.argv) > 3 and sys.argv[3] == "shm_gpu_open":
        return main_shm_gpu_open(sys.argv[1], int(sys.argv[2]))
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
