"""The GKR half of a chip proof over ROW-SHARDED witness columns (ceno_dist_create_chip_proof, ceno_amd/host/dist_gkr.cpp; SURVEY section 8(e):
a3, a5-a10): record inference, tower witness and tower proof on `world` virtual ranks — threads of one process on their own streams and
transcripts, exchanging through the in-process group — must produce, on EVERY rank, the proof ceno_prover_create_chip_proof produces from the
whole columns, bit for bit; and that proof equals the oracle's (tests/test_gpu_flows.py pins the single-device flow to the oracle: one
representative shape is re-checked here).  Reference: ZKVMProver::create_chip_proof ceno_zkvm/src/scheme/prover.rs:717-833, tower prover
scheme/cpu/mod.rs:346-554."""
import os
import threading

import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
P = po.P


@pytest.fixture(scope="module")
def dev():
    from ceno_amd import Device

    d = Device(0)
    yield d
    d.close()


@pytest.fixture(scope="module")
def prover():
    from ceno_amd import prover as p

    return p


def record_plan(w, n_records, alpha, beta):
    b2 = po.e2_mul(beta, beta)
    terms, coeffs, out_terms = [], [], []
    for k in range(n_records):
        base = len(terms)
        terms += [[(2 * k) % w], [(2 * k + 1) % w], [(3 * k + 5) % w, (k + 7) % w]]
        coeffs += [beta, b2, alpha]
        out_terms.append([base, base + 1, base + 2])
    return po.ext(coeffs), terms, out_terms


def comm_stats(prover, comm, reset=False):
    """ceno_dist_comm_stats of a communicator handle: what this rank put on the wire"""
    import ctypes as C

    L = prover.plib()
    L.ceno_dist_comm_stats.restype = C.c_int
    L.ceno_dist_comm_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
    out = (C.c_uint64 * 4)()
    assert L.ceno_dist_comm_stats(comm, out, int(reset)) == 0
    return dict(message_exchanges=int(out[0]), message_bytes_sent=int(out[1]), bulk_exchanges=int(out[2]), bulk_bytes_sent=int(out[3]))


def proofs_equal(a, b):
    return (np.array_equal(a.tower_msgs, b.tower_msgs) and np.array_equal(a.tower_point, b.tower_point) and
            np.array_equal(a.tower_prod_evals, b.tower_prod_evals) and np.array_equal(a.tower_logup_evals, b.tower_logup_evals) and
            np.array_equal(a.r_out_evals, b.r_out_evals) and np.array_equal(a.w_out_evals, b.w_out_evals) and
            np.array_equal(a.lk_out_evals, b.lk_out_evals) and np.array_equal(a.rt_main, b.rt_main) and a.tower_num_vars == b.tower_num_vars)


def run_sharded(dev, prover, cols, world, q, log2_n, shape, coeffs, terms, out_terms, challenges, seed):
    num_reads, num_writes, num_lk_tables, num_lk = shape
    w = len(cols)
    group = prover.LocalGroup(world)
    results, errors = [None] * world, []

    def rank_main(g):
        try:
            st = dev.stream_create()
            local = [dev.upload(prover.shard_rows(c, world, g, q)) for c in cols]
            task = dict(mles=local, n_witin=w, n_fixed=0, n_structural=0, num_instances=(1 << log2_n) - 5, log2_num_instances=log2_n - (world.bit_length() - 1),
                        num_reads=num_reads, num_writes=num_writes, num_lk_tables=num_lk_tables, num_lk=num_lk, record_coeffs=coeffs, record_terms=terms,
                        record_out_terms=out_terms)
            results[g] = prover.dist_create_chip_proof(dev, group.comms[g], task, log2_n, q, challenges, prover.Transcript.stub(seed), st)
            dev.sync(st)
            for m in local:
                m.free()
            dev.stream_destroy(st)
        except Exception as e:  # noqa: BLE001
            errors.append((g, e))

    ths = [threading.Thread(target=rank_main, args=(g,)) for g in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(300)
    alive = any(t.is_alive() for t in ths)
    if not alive:
        group.close()
    assert not alive, "a virtual rank hangs"
    assert not errors, errors
    return results


@pytest.mark.parametrize("world,log2_n,q,shape", [
    (2, 12, 4, (4, 4, 0, 8)),      # the ADD shape: two product towers of 2 record bits, one LogUp tower of 3 -> two engine groups per layer
    (4, 12, 3, (4, 4, 0, 8)),
    (8, 13, 3, (4, 4, 0, 8)),
    (4, 11, 4, (3, 0, 2, 0)),      # table circuit: numerators, no write set
    (2, 10, 5, (0, 1, 0, 1)),      # single records (no record bits)
    (4, 12, 2, (5, 2, 0, 3)),      # three different record counts -> three engine groups
    (8, 14, 10 - 3, (4, 4, 0, 8)),
])
def test_row_sharded_chip_proof_equals_the_single_device_proof(dev, prover, world, log2_n, q, shape):
    num_reads, num_writes, num_lk_tables, num_lk = shape
    w = 9
    rows = 1 << log2_n
    n_lk_den = num_lk_tables if num_lk_tables else num_lk
    n_rec = num_reads + num_writes + num_lk_tables + n_lk_den
    alpha, beta = (0x1234567, 0x89ABCDE), (0x13579B, 0x2468AC)
    cols = [po.rand_base(rows, 700 + j) for j in range(w)]
    coeffs, terms, out_terms = record_plan(w, n_rec, alpha, beta)
    full = [dev.upload(c) for c in cols]
    task = dict(mles=full, n_witin=w, n_fixed=0, n_structural=0, num_instances=rows - 5, log2_num_instances=log2_n, num_reads=num_reads,
                num_writes=num_writes, num_lk_tables=num_lk_tables, num_lk=num_lk, record_coeffs=coeffs, record_terms=terms, record_out_terms=out_terms)
    want = prover.create_chip_proof(dev, task, [alpha, beta], prover.Transcript.stub(21))
    got = run_sharded(dev, prover, cols, world, q, log2_n, shape, coeffs, terms, out_terms, [alpha, beta], 21)
    for g in range(world):
        assert proofs_equal(got[g], want), f"rank {g} of {world}: the sharded proof differs from the single-device proof"
    for m in full:
        m.free()


def test_row_sharded_chip_proof_matches_the_oracle(dev, prover):
    """the same flow against the oracle's restatement directly (one shape; the single-device flow is pinned to the oracle on five)"""
    world, log2_n, q = 4, 10, 3
    num_reads, num_writes, num_lk_tables, num_lk = 4, 4, 0, 8
    w, rows = 9, 1 << log2_n
    alpha, beta = (0x1234567, 0x89ABCDE), (0x13579B, 0x2468AC)
    cols = [po.rand_base(rows, 900 + j) for j in range(w)]
    coeffs, terms, out_terms = record_plan(w, 16, alpha, beta)
    got = run_sharded(dev, prover, cols, world, q, log2_n, (num_reads, num_writes, num_lk_tables, num_lk), coeffs, terms, out_terms, [alpha, beta], 33)[0]
    recs = [po.wit_infer(cols, coeffs[ts[0]: ts[-1] + 1], [terms[t] for t in ts], log2_n) for ts in out_terms]
    prod_specs, out_evals = [], []
    for group in (recs[:4], recs[4:8]):
        limbs = po.interleaving_mles_to_mles(group, rows, 2, (1, 0))
        layers = po.infer_tower_product_witness(int(limbs[0].shape[0]).bit_length(), limbs)
        prod_specs.append(layers)
        out_evals += [layers[0][0][0], layers[0][1][0]]
    ql = po.interleaving_mles_to_mles(recs[8:], rows, 2, alpha)
    layers = po.infer_tower_logup_witness(None, ql)
    out_evals += [layers[0][k][0] for k in range(4)]
    tr = po.StubTranscript(33)
    for e in out_evals:
        tr.append_ext((int(e[0]), int(e[1])))
    oproof = po.tower_prove(prod_specs, [layers], tr)
    assert np.array_equal(got.tower_msgs, oproof.msgs) and np.array_equal(got.tower_point, oproof.point[: got.tower_num_vars])
    assert np.array_equal(got.tower_prod_evals, oproof.prod_evals) and np.array_equal(got.tower_logup_evals, oproof.logup_evals)


@pytest.mark.parametrize("world,log2_inst,rot_log,q", [(2, 3, 5, 5), (4, 4, 5, 6), (2, 2, 6, 6), (8, 5, 5, 5), (4, 4, 6, 7)])
def test_row_sharded_chip_proof_with_rotation_equals_the_single_device_proof(dev, prover, world, log2_inst, rot_log, q):
    """a keccak-style chip (2^log2_inst instances x 2^rot_log rotation rows; rotation pairs inside cyclic groups of 32 / 64 rows) with its ROWS
    sharded: tower proof as before, then the rotation argument — local rotations, q local sumcheck rounds, the gathered tail replicated, left
    evaluations summed over the ranks — must give the single-device proof's messages, points and evaluations word for word on every rank
    (prove_rotation, gkr_iop/src/gkr/layer/cpu/mod.rs:249-389)"""
    w = 4
    nv = log2_inst + rot_log
    k = world.bit_length() - 1
    alpha, beta = (5, 6), (7, 8)
    src = po.rand_base(1 << nv, 31 + world)
    cols = [src, po.rotation_next_base_mle(src, rot_log), po.rand_base(1 << nv, 32), po.rand_base(1 << nv, 33)]
    coeffs, terms, out_terms = record_plan(w, 3, alpha, beta)
    rotation = dict(pairs=[(0, 1), (2, 3)], cyclic_subgroup_size=23 if rot_log == 5 else 45, cyclic_group_log2=rot_log)
    base = dict(n_witin=w, n_fixed=0, n_structural=0, num_instances=(1 << log2_inst) - 1, rotation_vars=rot_log, num_reads=1, num_writes=1,
                num_lk_tables=0, num_lk=1, record_coeffs=coeffs, record_terms=terms, record_out_terms=out_terms, rotation=rotation)
    mles = [dev.upload(c) for c in cols]
    want = prover.create_chip_proof(dev, dict(base, mles=mles, log2_num_instances=log2_inst), [alpha, beta], prover.Transcript.stub(8))
    for m in mles:
        m.free()
    group = prover.LocalGroup(world)
    results, errors = [None] * world, []

    def rank_main(g):
        try:
            st = dev.stream_create()
            local = [dev.upload(prover.shard_rows(c, world, g, q)) for c in cols]
            task = dict(base, mles=local, log2_num_instances=log2_inst - k)
            results[g] = prover.dist_create_chip_proof(dev, group.comms[g], task, log2_inst, q, [alpha, beta], prover.Transcript.stub(8), st)
            dev.sync(st)
            for m in local:
                m.free()
            dev.stream_destroy(st)
        except Exception as e:  # noqa: BLE001
            import traceback

            errors.append((g, repr(e), traceback.format_exc(limit=3)))

    ths = [threading.Thread(target=rank_main, args=(g,)) for g in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(300)
    alive = any(t.is_alive() for t in ths)
    if not alive:
        group.close()
    assert not alive, "a virtual rank hangs"
    assert not errors, errors
    for g in range(world):
        got = results[g]
        assert proofs_equal(got, want), f"rank {g}: tower part"
        assert np.array_equal(got.rotation_msgs, want.rotation_msgs), f"rank {g}: rotation messages"
        assert np.array_equal(got.rotation_points, want.rotation_points) and np.array_equal(got.rotation_evals, want.rotation_evals), f"rank {g}"


def test_row_sharded_main_constraints_refuse_what_they_cannot_shard(dev, prover):
    """a SHARDED chip with a selector that is no eq table on a row range (OrderedSparse): refused with a message instead of a wrong proof (a chip
    too small to be sharded is proved replicated, whatever its selectors)"""
    from ceno_amd.api import CenoHipError

    world, q, w = 4, 4, 3
    group = prover.LocalGroup(world)

    def job(nv, sel):
        cols = [dev.upload(prover.shard_rows(po.rand_base(1 << nv, 60 + j), world, 0, q)) for j in range(w)]
        return dict(num_vars=nv, mles=cols + [None], n_witin=w, n_fixed=0, n_structural=1, selectors=[sel], n_exprs=2, max_degree=3,
                    terms=[[w, 0, 1], [w, 2]], scalars=[[((3, 1), [2])], [((5, 0), [3])]])

    with pytest.raises(CenoHipError) as ei:
        prover.dist_prove_batched_main_constraints(dev, group.comms[0], [job(8, (po.SEL_ORDERED_SPARSE, 0, 10, 0, (0, 2), 2, po.rand_ext(8, 2)))],
                                                   [(1, 2), (3, 4)], prover.Transcript.stub(1), q)
    assert "Whole and Prefix" in str(ei.value)
    group.close()


def test_row_sharded_chip_proof_refuses_what_it_cannot_shard(dev, prover):
    from ceno_amd.api import CenoHipError

    world, log2_n, q = 4, 6, 5   # needs log2 rows >= q + log2 world + 1
    cols = [po.rand_base(1 << log2_n, 5 + j) for j in range(4)]
    coeffs, terms, out_terms = record_plan(4, 2, (3, 4), (5, 6))
    group = prover.LocalGroup(world)
    local = [dev.upload(prover.shard_rows(c, world, 0, 1)) for c in cols]
    task = dict(mles=local, n_witin=4, n_fixed=0, n_structural=0, num_instances=60, log2_num_instances=log2_n - 2, num_reads=1, num_writes=1,
                num_lk_tables=0, num_lk=0, record_coeffs=coeffs, record_terms=terms, record_out_terms=out_terms)
    with pytest.raises(CenoHipError) as ei:
        prover.dist_create_chip_proof(dev, group.comms[0], task, log2_n, q, [(3, 4), (5, 6)], prover.Transcript.stub(1))
    assert "too small" in str(ei.value)
    for m in local:
        m.free()
    group.close()


def test_multi_rank_opening_refuses_a_commitment_shorter_than_the_ranks(dev, prover):
    """a commitment whose TALLEST codeword has fewer rows than there are ranks has no row shards at all: the opening says so before it touches
    any table"""
    from ceno_amd.api import CenoHipError

    group = prover.LocalGroup(8)
    stream = dev.stream_create()
    pts = [np.zeros((1, 2), dtype=np.uint64), np.zeros((1, 2), dtype=np.uint64)]
    evs = [np.zeros((8, 2), dtype=np.uint64)] * 2
    with pytest.raises(CenoHipError) as ei:
        prover.dist_basefold_open(dev, group.comms[0], [1, 1], [[1] * 8, [1] * 8], 1, [8, 8], [8, 8], 8, 8, pts, evs, 4, 0, prover.Transcript.stub(1), stream)
    assert "fewer rows than there are ranks" in str(ei.value)
    group.close()


# ------------------------------------------------------------------------------------------------------------------
# the opening of a commitment made across ranks (ceno_dist_basefold_open, ceno_amd/host/dist_open.cpp)
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("world,log_rows,col_split,transcript", [
    (2, 8, [[2, 1], [1, 3]], "stub"),
    (4, 9, [[2, 1, 0, 3], [1, 1, 2, 1]], "stub"),          # a rank without columns of the first matrix
    (8, 10, [[1] * 8], "poseidon2"),
    (4, 7, [[3, 2, 2, 1]], "poseidon2"),
    # matrices of different heights in one commitment (a shard's traces): one batched codeword per height class, mixed-height sub-trees
    (2, [8, 6], [[2, 1], [1, 3]], "stub"),
    (4, [7, 9, 5, 9], [[1, 1, 1, 1], [2, 1, 0, 3], [1, 2, 1, 1], [1, 1, 2, 1]], "poseidon2"),
    (8, [10, 6, 8], [[1] * 8, [2, 1, 1, 1, 1, 1, 1, 1], [1] * 8], "stub"),
    # chips of one or two rows in a shard (traces are padded to two rows, scheme/hal.rs:127-128): a codeword with exactly as many rows as ranks
    # joins at the sub-trees' roots; one with FEWER rows than ranks is held whole by every rank and lives in the replicated top tree
    (4, [6, 1, 5], [[1, 1, 1, 1], [1, 2, 1, 1], [2, 1, 1, 1]], "stub"),
    (8, [1, 9, 1], [[1] * 8, [1] * 8, [1, 1, 2, 1, 1, 1, 1, 1]], "poseidon2"),
    (8, [7, 1], [[1] * 8, [2, 1, 1, 1, 1, 1, 1, 3]], "stub"),
])
def test_multi_rank_opening_equals_the_single_device_opening(dev, prover, world, log_rows, col_split, transcript):
    """commit across `world` virtual ranks (column-sharded RS encoding, re-shard by rows, sub-trees + replicated top), then open across them:
    batched codeword from the row shards, batched polynomial from the column shards (all-gather + modular sum), owner ranks answer the queries
    of the commitment, the commit phase replicated — every rank must end with the proof the single-device opening of the same matrices gives,
    word for word, under the stub and under the Poseidon2 duplex transcript (OpeningProver::open, ceno_zkvm/src/scheme/cpu/mod.rs:1418-1457;
    protocol ceno_recursion_v2/src/pcs/mod.rs:1111-1316)"""
    import torch

    from ceno_amd import dist as cdist

    blow, n_queries, pow_bits = 1, 12, 4
    n_mats = len(col_split)
    heights = list(log_rows) if isinstance(log_rows, list) else [log_rows] * n_mats
    fulls = [po.rand_base((1 << heights[i]) * sum(ws), 800 + i).reshape(1 << heights[i], sum(ws)) for i, ws in enumerate(col_split)]
    point = np.array([[(i * 7919 + 13) % P, (i * 104729 + 17) % P] for i in range(max(heights))], dtype=np.uint64)
    points = [point[:h] for h in heights]
    evals = [np.array([po.mle_evaluate(np.ascontiguousarray(full[:, c]), points[m]) for c in range(full.shape[1])], dtype=np.uint64)
             for m, full in enumerate(fulls)]
    new_tr = (lambda: prover.Transcript.stub(77)) if transcript == "stub" else (lambda: prover.Transcript.poseidon2(b"open"))
    stream = dev.stream_create()
    pcs = prover.PcsData(dev, fulls, blow, stream)
    want_root = pcs.root()
    want = pcs.basefold_open(points, evals, n_queries, pow_bits, new_tr())
    pcs.free()
    group = prover.LocalGroup(world)
    res, errors = [None] * world, []

    def run(rank):
        try:
            keep, ptrs = [], []
            for ws, full in zip(col_split, fulls):
                c0 = sum(ws[:rank])
                cols = np.ascontiguousarray(full[:, c0:c0 + ws[rank]].T)
                t = torch.from_numpy(cols.view(np.int64).copy()).to("cuda:0") if cols.size else torch.empty(1, dtype=torch.int64, device="cuda:0")
                keep.append(t)
                ptrs.append(t.data_ptr())
            torch.cuda.synchronize()
            s_ = dev.stream_create()
            com = cdist.sharded_commit_mmcs_native(dev, group.comms[rank], ptrs, col_split, heights, blow, rank, s_)
            dev.sync(s_)
            proof = prover.dist_basefold_open(dev, group.comms[rank], log_rows, col_split, blow, ptrs, [t.data_ptr() for t in com["codeword_rows"]], com["subtree"],
                                              com["top"], points, evals, n_queries, pow_bits, new_tr(), s_)
            res[rank] = (np.asarray(com["root"]), proof)
            for key in ("subtree", "top"):
                if com.get(key):
                    dev.L.ceno_hip_merkle_free(dev.h, com[key])
        except Exception as e:  # noqa: BLE001
            import traceback

            errors.append((rank, repr(e), traceback.format_exc(limit=4)))

    ths = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(300)
    alive = any(t.is_alive() for t in ths)
    if not alive:
        group.close()
    assert not alive, "a virtual rank hangs"
    assert not errors, errors
    for r in range(world):
        assert np.array_equal(res[r][0].reshape(-1), np.asarray(want_root, dtype=np.uint64).reshape(-1)), f"rank {r}: commitment root"
        assert res[r][1].shape == want.shape and np.array_equal(res[r][1], want), f"rank {r}: the opening differs from the single-device opening"


@pytest.mark.parametrize("world,heights_w,heights_f,transcript", [
    (2, [8, 6], [8], "stub"),              # witness of two heights + a fixed commitment that shares the taller one (one batched codeword for both)
    (4, [9], [7, 9], "poseidon2"),
    (8, [7, 10], [6], "stub"),             # the fixed commitment is shorter than every height of the witness
])
def test_multi_rank_opening_of_witness_and_fixed_commitments(dev, prover, world, heights_w, heights_f, transcript):
    """OpeningProver::open takes the witness AND the fixed commitment (scheme/hal.rs:284-294, cpu/mod.rs:1418-1457): two commitments made across the
    ranks, opened in one proof — a height both share is ONE batched codeword (the second is added to the gathered first), each commitment answers
    the queries from its own sub-trees — equal to the single-device opening of the same two commitments word for word"""
    import torch

    from ceno_amd import dist as cdist

    blow, n_queries, pow_bits = 1, 10, 3
    sets = []
    for tag, heights in (("w", heights_w), ("f", heights_f)):
        col_split = [[1 + ((m + g + len(tag)) % 2) for g in range(world)] for m in range(len(heights))]
        fulls = [po.rand_base((1 << h) * sum(ws), 1300 + 7 * i + len(heights)).reshape(1 << h, sum(ws)) for i, (h, ws) in enumerate(zip(heights, col_split))]
        sets.append((heights, col_split, fulls))
    max_h = max(heights_w + heights_f)
    point = np.array([[(i * 7919 + 29) % P, (i * 104729 + 31) % P] for i in range(max_h)], dtype=np.uint64)
    points, evals = [], []
    for heights, _cs, fulls in sets:
        for h, full in zip(heights, fulls):
            points.append(point[:h])
            evals.append(np.array([po.mle_evaluate(np.ascontiguousarray(full[:, c]), point[:h]) for c in range(full.shape[1])], dtype=np.uint64))
    new_tr = (lambda: prover.Transcript.stub(78)) if transcript == "stub" else (lambda: prover.Transcript.poseidon2(b"open2"))
    stream = dev.stream_create()
    pcs_w = prover.PcsData(dev, sets[0][2], blow, stream)
    pcs_f = prover.PcsData(dev, sets[1][2], blow, stream)
    want = pcs_w.basefold_open(points, evals, n_queries, pow_bits, new_tr(), more_commits=[pcs_f])
    pcs_w.free()
    pcs_f.free()
    group = prover.LocalGroup(world)
    res, errors = [None] * world, []

    def run(rank):
        try:
            s_ = dev.stream_create()
            keep, commits = [], []
            for heights, col_split, fulls in sets:
                ptrs = []
                for ws, full in zip(col_split, fulls):
                    c0 = sum(ws[:rank])
                    cols = np.ascontiguousarray(full[:, c0:c0 + ws[rank]].T)
                    t = torch.from_numpy(cols.view(np.int64).copy()).to("cuda:0")
                    keep.append(t)
                    ptrs.append(t.data_ptr())
                torch.cuda.synchronize()
                com = cdist.sharded_commit_mmcs_native(dev, group.comms[rank], ptrs, col_split, heights, blow, rank, s_)
                dev.sync(s_)
                keep.append(com)
                commits.append(dict(log_rows=heights, widths=col_split, trace_ptrs=ptrs, cw_row_ptrs=[t.data_ptr() for t in com["codeword_rows"]],
                                    subtree=com["subtree"], top=com["top"]))
            res[rank] = prover.dist_basefold_open_commits(dev, group.comms[rank], commits, blow, points, evals, n_queries, pow_bits, new_tr(), s_)
            for cm in commits:
                for key in ("subtree", "top"):
                    if cm.get(key):
                        dev.L.ceno_hip_merkle_free(dev.h, cm[key])
        except Exception as e:  # noqa: BLE001
            import traceback

            errors.append((rank, repr(e), traceback.format_exc(limit=4)))

    ths = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(300)
    alive = any(t.is_alive() for t in ths)
    if not alive:
        group.close()
    assert not alive, "a virtual rank hangs"
    assert not errors, errors
    for r in range(world):
        assert res[r].shape == want.shape and np.array_equal(res[r], want), f"rank {r}: the opening differs from the single-device opening"


@pytest.mark.parametrize("world,q,nvs,lincomb", [
    (2, 4, (9, 7, 8), None),
    (4, 3, (10, 8, 6), None),
    (8, 3, (11, 9, 7), None),
    (4, 4, (9, 9, 7), "0"),        # every column a table of the sumcheck (no combination)
    (2, 5, (10, 7), "1"),
    # chips too small to be sharded ride along replicated (whole tables on every rank, their part of a message added once)
    (4, 3, (10, 8, 5, 2), None),
    (8, 4, (12, 6, 3), "0"),
    (2, 6, (9, 7, 6, 1), None),
])
def test_row_sharded_main_constraints_equal_the_single_device_proof(dev, prover, monkeypatch, world, q, nvs, lincomb):
    """prove_batched_main_constraints over ROW-SHARDED tables in the block layout of the sharded chip proof: chips of different sizes, Prefix
    selectors that start and end anywhere (a rank's part of the range is a prefix of its own rows) and a Whole selector, products of two and
    three columns and columns that are read only linearly (combined per rank, their evaluations summed over the ranks) — q local rounds, the
    gathered tail replicated — must give, on every rank, the messages, point, evaluations and claimed sum of the single-device proof
    (BatchedMainConstraintProver::prove_batched_main_constraints, ceno_zkvm/src/scheme/cpu/mod.rs:1052-1390)"""
    if lincomb is not None:
        monkeypatch.setenv("CENO_PROVER_MAIN_LINCOMB", lincomb)
    gch, w = [(11, 22), (33, 44)], 7
    k = world.bit_length() - 1
    cases = []
    for c, nv in enumerate(nvs):
        cols = [po.rand_base(1 << nv, 3100 + 13 * c + j) for j in range(w)]
        point = po.rand_ext(nv, 3200 + c)
        sels = [(po.SEL_PREFIX, min(5 * c, (1 << nv) // 4), max(1, (1 << nv) - 9 - 7 * c) if nv >= 5 else max(1, (1 << nv) - 1 - min(5 * c, (1 << nv) // 4)), 0, (), 0, point)]
        terms = [[w, 0, 1], [w, 1, 2, 3], [w, 2, 0], [w, 3], [w, 4], [w, 5], [w, 6], [w, 1], [w]]
        if c == 1:  # a second selector (Whole) over some of the columns
            sels.append((po.SEL_WHOLE, 0, 0, 1, (), 0, point))
            terms += [[w + 1, 4], [w + 1, 5, 6], [w + 1, 2]]
        scalars = [[((9 + 3 * t + c, 2 + t), [2 + (t % 2)])] + ([((5 + t, 0), [t % 2, 3])] if t % 3 == 0 else []) for t in range(len(terms))]
        cases.append((nv, cols, sels, terms, scalars))

    def jobs_for(tables_of):
        return [dict(num_vars=nv, mles=tables_of(cols, nv) + [None] * len(sels), n_witin=w, n_fixed=0, n_structural=len(sels), selectors=sels, n_exprs=2,
                     max_degree=4, terms=terms, scalars=scalars) for (nv, cols, sels, terms, scalars) in cases]

    full = jobs_for(lambda cols, nv: [dev.upload(c_) for c_ in cols])
    want = prover.prove_batched_main_constraints(dev, full, gch, prover.Transcript.stub(5))
    group = prover.LocalGroup(world)
    results, errors = [None] * world, []

    def rank_main(g):
        try:
            st = dev.stream_create()
            # (a chip of fewer than 2^(q + k + 1) rows is not sharded: every rank passes its whole tables)
            local = jobs_for(lambda cols, nv: [dev.upload(prover.shard_rows(c_, world, g, q) if nv - k >= q + 1 else c_) for c_ in cols])
            results[g] = prover.dist_prove_batched_main_constraints(dev, group.comms[g], local, gch, prover.Transcript.stub(5), q, st)
            dev.sync(st)
            dev.stream_destroy(st)
        except Exception as e:  # noqa: BLE001
            import traceback

            errors.append((g, repr(e), traceback.format_exc(limit=3)))

    ths = [threading.Thread(target=rank_main, args=(g,)) for g in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(300)
    alive = any(t.is_alive() for t in ths)
    if not alive:
        group.close()
    assert not alive, "a virtual rank hangs"
    assert not errors, errors
    for g in range(world):
        got = results[g]
        assert np.array_equal(got[1], want[1]), f"rank {g}: messages"
        assert np.array_equal(got[2], want[2]) and np.array_equal(got[3], want[3]), f"rank {g}: point / evaluations"
        assert got[0] == want[0], f"rank {g}: claimed sum"


@pytest.mark.parametrize("world,log2_n,q", [(2, 10, 4), (4, 11, 3)])
def test_gkr_half_and_main_constraints_of_one_chip_on_one_row_layout(dev, prover, world, log2_n, q):
    """the phases of a chip between commitment and opening on ONE layout of its rows (blocks of 2^q rows dealt round-robin): the row-sharded chip
    proof, then — selectors at the rt_main it returns, the same transcript — the row-sharded main-constraint sumcheck; every rank ends with the
    single-device flow's chip proof, messages, point and evaluations"""
    shape, w = (4, 4, 0, 8), 9
    rows = 1 << log2_n
    n_rec = shape[0] + shape[1] + shape[2] + (shape[2] if shape[2] else shape[3])
    alpha, beta = (0x1234567, 0x89ABCDE), (0x13579B, 0x2468AC)
    cols = [po.rand_base(rows, 700 + j) for j in range(w)]
    coeffs, terms, out_terms = record_plan(w, n_rec, alpha, beta)
    mterms = [[w, j, (j + 1) % w] for j in range(w)] + [[w, j, (j + 3) % w, (j + 5) % w] for j in range(0, w, 3)] + [[w, j] for j in range(w)]
    mscal = [[((3 + 5 * t, 11 * t + 1), [2 + (t % 2)])] for t in range(len(mterms))]

    def flow(dev_cols, log2_local, chip_fn, main_fn):
        task = dict(mles=dev_cols, n_witin=w, n_fixed=0, n_structural=0, num_instances=rows - 5, log2_num_instances=log2_local, num_reads=shape[0],
                    num_writes=shape[1], num_lk_tables=shape[2], num_lk=shape[3], record_coeffs=coeffs, record_terms=terms, record_out_terms=out_terms)
        tr = prover.Transcript.stub(21)
        proof = chip_fn(task, tr)
        sel = (po.SEL_PREFIX, 0, rows - 5, 0, (), 0, np.ascontiguousarray(proof.rt_main))
        job = dict(num_vars=log2_n, mles=dev_cols + [None], n_witin=w, n_fixed=0, n_structural=1, selectors=[sel], n_exprs=2, max_degree=4, terms=mterms,
                   scalars=mscal)
        return proof, main_fn([job], tr)

    full = [dev.upload(c) for c in cols]
    want_proof, want_main = flow(full, log2_n, lambda task, tr: prover.create_chip_proof(dev, task, [alpha, beta], tr),
                                 lambda jobs, tr: prover.prove_batched_main_constraints(dev, jobs, [alpha, beta], tr))
    group = prover.LocalGroup(world)
    results, errors = [None] * world, []

    def rank_main(g):
        try:
            st = dev.stream_create()
            local = [dev.upload(prover.shard_rows(c, world, g, q)) for c in cols]
            results[g] = flow(local, log2_n - (world.bit_length() - 1),
                              lambda task, tr: prover.dist_create_chip_proof(dev, group.comms[g], task, log2_n, q, [alpha, beta], tr, st),
                              lambda jobs, tr: prover.dist_prove_batched_main_constraints(dev, group.comms[g], jobs, [alpha, beta], tr, q, st))
            dev.sync(st)
            dev.stream_destroy(st)
        except Exception as e:  # noqa: BLE001
            import traceback

            errors.append((g, repr(e), traceback.format_exc(limit=3)))

    ths = [threading.Thread(target=rank_main, args=(g,)) for g in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(300)
    alive = any(t.is_alive() for t in ths)
    if not alive:
        group.close()
    assert not alive, "a virtual rank hangs"
    assert not errors, errors
    for g in range(world):
        proof, main = results[g]
        assert proofs_equal(proof, want_proof), f"rank {g}: chip proof"
        assert main[0] == want_main[0] and all(np.array_equal(a, b) for a, b in zip(main[1:], want_main[1:])), f"rank {g}: main constraints"


@pytest.mark.parametrize("world,log2_n,q,transcript", [(2, 10, 4, "stub"), (4, 11, 3, "poseidon2")])
def test_whole_chip_flow_across_ranks_equals_the_single_device_flow(dev, prover, world, log2_n, q, transcript):
    whole_chip_flow(dev, prover, world, log2_n, q, transcript)


def whole_chip_flow(dev, prover, world, log2_n, q, transcript, w=9, shape=(4, 4, 0, 8), stats=None):
    """config #3's flow with every phase across the ranks on ONE transcript: the commitment (column shards; ceno_dist_commit_traces_mmcs), its root
    into the transcript, two challenges out, the chip proof (row shards, block layout; ceno_dist_create_chip_proof), the main-constraint sumcheck
    at its rt_main (same row shards; ceno_dist_prove_batched_main_constraints), the opening of the commitment at the sumcheck's point with the
    sumcheck's evaluations (ceno_dist_basefold_open) — every rank must end with the root, the chip proof, the sumcheck's messages / point /
    evaluations and the opening proof of the single-device flow, word for word (ZKVMProver::create_proof's per-chip phases,
    ceno_zkvm/src/scheme/prover.rs:324-586)"""
    import torch

    from ceno_amd import dist as cdist

    blow, nq, pow_bits = 1, 8, 3
    rows, k = 1 << log2_n, world.bit_length() - 1
    n_rec = shape[0] + shape[1] + shape[3]
    cols = [po.rand_base(rows, 900 + j) for j in range(w)]
    col_split = [[w // world + (1 if g < w % world else 0) for g in range(world)]]
    mterms = [[w, j, (j + 1) % w] for j in range(w)] + [[w, j, (j + 3) % w, (j + 5) % w] for j in range(0, w, 3)] + [[w, j] for j in range(w)]
    mscal = [[((3 + 5 * t, 11 * t + 1), [2 + (t % 2)])] for t in range(len(mterms))]
    new_tr = (lambda: prover.Transcript.stub(31)) if transcript == "stub" else (lambda: prover.Transcript.poseidon2(b"flow"))

    def flow(tables, log2_local, commit_fn, chip_fn, main_fn, open_fn):
        import time as _t
        _t0 = _t.time()
        def lap(what):
            if os.environ.get("CENO_TEST_TIMING"):
                print(f"[flow {log2_local}] {what}: {_t.time() - _t0:.2f} s", flush=True)
        laps = {}
        _lap = lap

        def lap(what):  # noqa: F811
            laps[what] = _t.time() - _t0
            _lap(what)
        tr = new_tr()
        root, commit_state = commit_fn()
        lap("commit")
        root = np.asarray(root, dtype=np.uint64).reshape(-1)
        for v in root:
            tr.append_base(int(v))
        alpha, beta = tr.sample_ext(), tr.sample_ext()
        coeffs, terms, out_terms = record_plan(w, n_rec, alpha, beta)
        task = dict(mles=tables, n_witin=w, n_fixed=0, n_structural=0, num_instances=rows - 5, log2_num_instances=log2_local, num_reads=shape[0],
                    num_writes=shape[1], num_lk_tables=shape[2], num_lk=shape[3], record_coeffs=coeffs, record_terms=terms, record_out_terms=out_terms)
        proof = chip_fn(task, [alpha, beta], tr)
        lap("chip proof")
        sel = (po.SEL_PREFIX, 0, rows - 5, 0, (), 0, np.ascontiguousarray(proof.rt_main))
        job = dict(num_vars=log2_n, mles=tables + [None], n_witin=w, n_fixed=0, n_structural=1, selectors=[sel], n_exprs=2, max_degree=4, terms=mterms,
                   scalars=mscal)
        main = main_fn([job], [alpha, beta], tr)
        lap("main")
        opening = open_fn(commit_state, [np.ascontiguousarray(main[2][:log2_n])], [np.ascontiguousarray(main[3][:w])], tr)
        lap("open")
        flow.laps = laps
        return root, proof, main, opening

    stream = dev.stream_create()
    full = [dev.upload(c) for c in cols]
    matrix = np.ascontiguousarray(np.stack(cols, axis=1))

    def commit_single():
        pcs = prover.PcsData(dev, [matrix], blow, stream)
        return pcs.root(), pcs

    want = flow(full, log2_n, commit_single, lambda task, ch, tr: prover.create_chip_proof(dev, task, ch, tr),
                lambda jobs, ch, tr: prover.prove_batched_main_constraints(dev, jobs, ch, tr),
                lambda pcs, pts, evs, tr: pcs.basefold_open(pts, evs, nq, pow_bits, tr))
    single_laps = dict(flow.laps)
    group = prover.LocalGroup(world)
    rank_stats = [None] * world
    results, errors = [None] * world, []

    def rank_main(g):
        try:
            st = dev.stream_create()
            c0 = sum(col_split[0][:g])
            mine = np.ascontiguousarray(matrix[:, c0:c0 + col_split[0][g]].T)
            t = torch.from_numpy(mine.view(np.int64).copy()).to("cuda:0")
            torch.cuda.synchronize()
            local = [dev.upload(prover.shard_rows(c, world, g, q)) for c in cols]

            def commit_dist():
                com = cdist.sharded_commit_mmcs_native(dev, group.comms[g], [t.data_ptr()], col_split, [log2_n], blow, g, st)
                dev.sync(st)
                return com["root"], com

            results[g] = flow(local, log2_n - k, commit_dist,
                              lambda task, ch, tr: prover.dist_create_chip_proof(dev, group.comms[g], task, log2_n, q, ch, tr, st),
                              lambda jobs, ch, tr: prover.dist_prove_batched_main_constraints(dev, group.comms[g], jobs, ch, tr, q, st),
                              lambda com, pts, evs, tr: prover.dist_basefold_open(dev, group.comms[g], log2_n, col_split, blow, [t.data_ptr()],
                                                                                  [x.data_ptr() for x in com["codeword_rows"]], com["subtree"], com["top"], pts,
                                                                                  evs, nq, pow_bits, tr, st))
            dev.sync(st)
            rank_stats[g] = comm_stats(prover, group.comms[g])
        except Exception as e:  # noqa: BLE001
            import traceback

            errors.append((g, repr(e), traceback.format_exc(limit=3)))

    ths = [threading.Thread(target=rank_main, args=(g,)) for g in range(world)]
    for t_ in ths:
        t_.start()
    for t_ in ths:
        t_.join(600)
    alive = any(t_.is_alive() for t_ in ths)
    if not alive:
        group.close()
    assert not alive, "a virtual rank hangs"
    assert not errors, errors
    if stats is not None:
        stats.update(single_device_s=single_laps, rank0_wire=rank_stats[0], world=world, log2_n=log2_n, q=q, w=w)
    for g in range(world):
        root, proof, main, opening = results[g]
        assert np.array_equal(root, want[0]), f"rank {g}: commitment root"
        assert proofs_equal(proof, want[1]), f"rank {g}: chip proof"
        assert main[0] == want[2][0] and all(np.array_equal(a, b) for a, b in zip(main[1:], want[2][1:])), f"rank {g}: main constraints"
        assert opening.shape == want[3].shape and np.array_equal(opening, want[3]), f"rank {g}: opening"


@pytest.mark.parametrize("world,q,transcript", [(2, 3, "stub"), (4, 3, "poseidon2")])
def test_whole_shard_flow_across_ranks_equals_the_single_device_flow(dev, prover, world, q, transcript):
    """a SHARD — chips of 2^11, 2^9, 2^6 and 2^2 rows — through every phase across the ranks on one transcript: ONE commitment of the four traces
    (column shards, mixed heights), its root in, two challenges out, a chip proof per chip (row-sharded where the chip is large enough, the
    single-device proof on every rank where it is not), ONE batched main-constraint sumcheck over all chips (sharded and replicated chips side by
    side), ONE opening of the commitment at every chip's prefix of the sumcheck's point.  Every rank ends with what the single-device flow
    produces, word for word (ZKVMProver::create_proof, ceno_zkvm/src/scheme/prover.rs:324-586)"""
    import torch

    from ceno_amd import dist as cdist

    shape, w, blow, nq, pow_bits = (2, 2, 0, 4), 6, 1, 8, 3
    k = world.bit_length() - 1
    log_rows = [11, 9, 6, 2]
    n_rec = shape[0] + shape[1] + shape[3]
    cols = [[po.rand_base(1 << lr, 1500 + 20 * c + j) for j in range(w)] for c, lr in enumerate(log_rows)]
    col_split = [[w // world + (1 if g < w % world else 0) for g in range(world)] for _ in log_rows]
    mterms = [[w, j, (j + 1) % w] for j in range(w)] + [[w, 0, 2, 4]] + [[w, j] for j in range(w)]
    mscal = [[((3 + 5 * t, 11 * t + 1), [2 + (t % 2)])] for t in range(len(mterms))]
    new_tr = (lambda: prover.Transcript.stub(41)) if transcript == "stub" else (lambda: prover.Transcript.poseidon2(b"shard"))
    sharded = [lr - k >= q + 1 for lr in log_rows]

    def flow(tables, commit_fn, chip_fn, main_fn, open_fn):
        tr = new_tr()
        root, commit_state = commit_fn()
        for v in np.asarray(root, dtype=np.uint64).reshape(-1):
            tr.append_base(int(v))
        alpha, beta = tr.sample_ext(), tr.sample_ext()
        coeffs, terms, out_terms = record_plan(w, n_rec, alpha, beta)
        proofs, jobs = [], []
        for c, lr in enumerate(log_rows):
            task = dict(mles=tables[c], n_witin=w, n_fixed=0, n_structural=0, num_instances=max(1, (1 << lr) - 1 - c), num_reads=shape[0], num_writes=shape[1],
                        num_lk_tables=shape[2], num_lk=shape[3], record_coeffs=coeffs, record_terms=terms, record_out_terms=out_terms)
            proofs.append(chip_fn(c, task, [alpha, beta], tr))
            sel = (po.SEL_PREFIX, 0, max(1, (1 << lr) - 1 - c), 0, (), 0, np.ascontiguousarray(proofs[-1].rt_main))
            jobs.append(dict(num_vars=lr, mles=tables[c] + [None], n_witin=w, n_fixed=0, n_structural=1, selectors=[sel], n_exprs=2, max_degree=4,
                             terms=mterms, scalars=mscal))
        main = main_fn(jobs, [alpha, beta], tr)
        pts, evs, off = [], [], 0
        for c, lr in enumerate(log_rows):
            pts.append(np.ascontiguousarray(main[2][:lr]))
            evs.append(np.ascontiguousarray(main[3][off:off + w]))
            off += w + 1
        opening = open_fn(commit_state, pts, evs, tr)
        return np.asarray(root, dtype=np.uint64).reshape(-1), proofs, main, opening

    stream = dev.stream_create()
    full = [[dev.upload(c_) for c_ in cc] for cc in cols]
    mats = [np.ascontiguousarray(np.stack(cc, axis=1)) for cc in cols]

    def commit_single():
        pcs = prover.PcsData(dev, mats, blow, stream)
        return pcs.root(), pcs

    want = flow(full, commit_single, lambda c, task, ch, tr: prover.create_chip_proof(dev, dict(task, log2_num_instances=log_rows[c]), ch, tr),
                lambda jobs, ch, tr: prover.prove_batched_main_constraints(dev, jobs, ch, tr),
                lambda pcs, pts, evs, tr: pcs.basefold_open(pts, evs, nq, pow_bits, tr))
    group = prover.LocalGroup(world)
    results, errors = [None] * world, []

    def rank_main(g):
        try:
            st = dev.stream_create()
            keep, ptrs = [], []
            for m_, ws in zip(mats, col_split):
                c0 = sum(ws[:g])
                t = torch.from_numpy(np.ascontiguousarray(m_[:, c0:c0 + ws[g]].T).view(np.int64).copy()).to("cuda:0")
                keep.append(t)
                ptrs.append(t.data_ptr())
            torch.cuda.synchronize()
            local = [[dev.upload(prover.shard_rows(c_, world, g, q) if sharded[c] else c_) for c_ in cc] for c, cc in enumerate(cols)]

            def commit_dist():
                com = cdist.sharded_commit_mmcs_native(dev, group.comms[g], ptrs, col_split, log_rows, blow, g, st)
                dev.sync(st)
                return com["root"], com

            def chip(c, task, ch, tr):
                if sharded[c]:
                    return prover.dist_create_chip_proof(dev, group.comms[g], dict(task, log2_num_instances=log_rows[c] - k), log_rows[c], q, ch, tr, st)
                return prover.create_chip_proof(dev, dict(task, log2_num_instances=log_rows[c]), ch, tr, st)  # replicated: the same proof on every rank

            results[g] = flow(local, commit_dist, chip, lambda jobs, ch, tr: prover.dist_prove_batched_main_constraints(dev, group.comms[g], jobs, ch, tr, q, st),
                              lambda com, pts, evs, tr: prover.dist_basefold_open(dev, group.comms[g], log_rows, col_split, blow, ptrs,
                                                                                  [x.data_ptr() for x in com["codeword_rows"]], com["subtree"], com["top"], pts,
                                                                                  evs, nq, pow_bits, tr, st))
            dev.sync(st)
        except Exception as e:  # noqa: BLE001
            import traceback

            errors.append((g, repr(e), traceback.format_exc(limit=3)))

    ths = [threading.Thread(target=rank_main, args=(g,)) for g in range(world)]
    for t_ in ths:
        t_.start()
    for t_ in ths:
        t_.join(300)
    alive = any(t_.is_alive() for t_ in ths)
    if not alive:
        group.close()
    assert not alive, "a virtual rank hangs"
    assert not errors, errors
    for g in range(world):
        root, proofs, main, opening = results[g]
        assert np.array_equal(root, want[0]), f"rank {g}: commitment root"
        for c in range(len(log_rows)):
            assert proofs_equal(proofs[c], want[1][c]), f"rank {g}: chip proof {c}"
        assert main[0] == want[2][0] and all(np.array_equal(a, b) for a, b in zip(main[1:], want[2][1:])), f"rank {g}: main constraints"
        assert opening.shape == want[3].shape and np.array_equal(opening, want[3]), f"rank {g}: opening"
