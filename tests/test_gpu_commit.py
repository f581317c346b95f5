"""GPU parity for the rows either side of the sumcheck: witness inference (a5), trace transpose, and the
Basefold commit path (a14: RS encode = NTT, Poseidon2 row hash, Merkle tree).  Bit-exact vs the oracle.
The commit path is PARITY UNPINNED against the reference (EXT constants); it is pinned against the
oracle's independent O(N^2) DFT and its own Poseidon2 restatement."""
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


def _dev_tensor(a: np.ndarray):
    from tests import hipbuf as torch

    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to("cuda:0")


def _to_np(t):
    return t.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("nv", [1, 10, 14])
def test_wit_infer_matches_oracle(dev, nv):
    cols = [po.rand_base(1 << nv, 10 + j) for j in range(6)] + [po.rand_ext(1 << nv, 99)]
    mles = [dev.upload(c) for c in cols]
    # records: r = alpha + beta*w0 + w1*w2 ; w = const + w3 ; lk = w4*w5*ext6 + w0
    terms = [[0], [1, 2], [3], [3, 3], [4, 5, 6], [0]]
    out_terms = [[0, 1], [2, 3], [4, 5]]
    coeffs = po.rand_ext(len(terms), 5)
    outs = dev.wit_infer(mles, coeffs, terms, out_terms, nv)
    for o, ts in enumerate(out_terms):
        exp = po.wit_infer(cols, coeffs[ts[0]: ts[-1] + 1], [terms[t] for t in ts], nv)
        assert np.array_equal(outs[o].download(), exp)


@pytest.mark.parametrize("rows,width", [(8, 3), (64, 22), (1000, 33), (4096, 1)])
def test_transpose(dev, rows, width):
    from ceno_amd import api

    m = po.rand_base(rows * width, rows + width).reshape(rows, width)
    src = _dev_tensor(m)
    from tests import hipbuf as torch

    dst = torch.empty(rows * width, dtype=torch.int64, device="cuda:0")
    api.transpose(dev, src.data_ptr(), rows, width, dst.data_ptr())
    dev.sync()
    assert np.array_equal(_to_np(dst).reshape(width, rows), m.T)


@pytest.mark.parametrize("rows,width", [(1 << 21, 22), ((1 << 21) + 37, 3), (1 << 22, 40)])
def test_transpose_large_row_counts(dev, rows, width):
    """>= 2^21 rows (the reference's add_op_21 bench, benches/riscv_add.rs:74-150): more than 65535 row tiles, which the
    grid's y dimension cannot hold.  Checked against numpy's transpose of the same words."""
    from tests import hipbuf as torch

    from ceno_amd import api

    h_src = np.random.default_rng(rows + width).integers(0, 1 << 62, rows * width, dtype=np.int64)
    src = torch.from_numpy(h_src).to("cuda:0")
    dst = torch.full((rows * width,), -1, dtype=torch.int64, device="cuda:0")
    api.transpose(dev, src.data_ptr(), rows, width, dst.data_ptr())
    dev.sync()
    assert np.array_equal(dst.cpu().numpy().reshape(width, rows), h_src.reshape(rows, width).T)


@pytest.mark.parametrize("log_n", [1, 2, 5, 9, 11, 12, 13, 16, 17])
def test_ntt_forward_inverse_vs_dft(dev, log_n):
    from ceno_amd import api

    n_cols = 3
    cols = np.stack([po.rand_base(1 << log_n, 7 * log_n + c) for c in range(n_cols)])
    d = _dev_tensor(cols)
    api.ntt_batch(dev, d.data_ptr(), log_n, n_cols, inverse=False)
    dev.sync()
    got = _to_np(d).reshape(n_cols, -1)
    for c in range(n_cols):  # O(N^2) transform by definition for small sizes, the oracle's radix-2 restatement beyond
        assert np.array_equal(got[c], po.dft_bitrev(cols[c]) if log_n <= 11 else po.fft_bitrev(cols[c]))
    api.ntt_batch(dev, d.data_ptr(), log_n, n_cols, inverse=True)
    dev.sync()
    assert np.array_equal(_to_np(d).reshape(n_cols, -1), cols)


def test_ntt_large_roundtrip_and_linearity(dev):
    from ceno_amd import api

    log_n, n_cols = 20, 4
    a = np.stack([po.rand_base(1 << log_n, 100 + c) for c in range(n_cols)])
    da = _dev_tensor(a)
    api.ntt_batch(dev, da.data_ptr(), log_n, n_cols)
    dev.sync()
    fa = _to_np(da).reshape(n_cols, -1).copy()
    # linearity: NTT(col0 + col1) = NTT(col0) + NTT(col1)
    s = ((a[0].astype(object) + a[1].astype(object)) % P).astype(np.uint64)
    ds = _dev_tensor(s[None])
    api.ntt_batch(dev, ds.data_ptr(), log_n, 1)
    dev.sync()
    exp = ((fa[0].astype(object) + fa[1].astype(object)) % P).astype(np.uint64)
    assert np.array_equal(_to_np(ds).reshape(-1), exp)
    # value at bit-reversed index 0 is the plain sum; index bitrev(1) = N/2 is f(w)
    assert int(fa[2][0]) == int(sum(int(x) for x in a[2]) % P)
    api.ntt_batch(dev, da.data_ptr(), log_n, n_cols, inverse=True)
    dev.sync()
    assert np.array_equal(_to_np(da).reshape(n_cols, -1), a)


def test_rs_encode_is_low_degree_extension(dev):
    from ceno_amd import api
    from tests import hipbuf as torch

    log_n, blow, n_cols = 8, 1, 2
    cols = np.stack([po.rand_base(1 << log_n, 40 + c) for c in range(n_cols)])
    src = _dev_tensor(cols)
    dst = torch.empty(n_cols << (log_n + blow), dtype=torch.int64, device="cuda:0")
    api.rs_encode(dev, src.data_ptr(), log_n, n_cols, blow, dst.data_ptr())
    dev.sync()
    got = _to_np(dst).reshape(n_cols, -1)
    for c in range(n_cols):
        padded = np.concatenate([cols[c], np.zeros(cols[c].shape[0] * ((1 << blow) - 1), dtype=np.uint64)])
        assert np.array_equal(got[c], po.dft_bitrev(padded))


def test_poseidon2_permutation_and_merkle(dev):
    from ceno_amd import api

    states = po.rand_base(8 * 100, 3).reshape(100, 8)
    d = _dev_tensor(states)
    api.poseidon2_permute(dev, d.data_ptr(), 100)
    dev.sync()
    got = _to_np(d).reshape(100, 8)
    for i in (0, 1, 57, 99):
        assert np.array_equal(got[i], po.poseidon2_permute(states[i]))
    # zero state must not stay zero and different inputs must differ
    assert not np.array_equal(po.poseidon2_permute(np.zeros(8, dtype=np.uint64)), np.zeros(8, dtype=np.uint64))
    for log_rows, width in [(0, 1), (3, 4), (6, 22), (5, 9)]:
        rows = 1 << log_rows
        m = po.rand_base(rows * width, 11 * width + log_rows).reshape(width, rows)  # column-major
        dm = _dev_tensor(m)
        t = api.Merkle(dev, dm.data_ptr(), log_rows, width)
        levels = po.merkle_commit(m, log_rows, width)
        assert np.array_equal(t.root(), levels[-1][0])
        for idx in {0, rows - 1, rows // 3}:
            path = t.open(idx)
            i = idx
            for l in range(log_rows):
                assert np.array_equal(path[l], levels[l][i ^ 1])
                i >>= 1
        t.free()


def test_poseidon2_set_constants_changes_and_restores(dev):
    from ceno_amd import CenoHipError, api

    s = po.rand_base(8, 5).reshape(1, 8)
    params = po.poseidon2_default_params()
    params2 = params.copy()
    params2[0] = (int(params2[0]) + 1) % P
    api.poseidon2_set_constants(dev, params2[:64], params2[64:86], params2[86:])
    d = _dev_tensor(s)
    api.poseidon2_permute(dev, d.data_ptr(), 1)
    dev.sync()
    assert np.array_equal(_to_np(d).reshape(-1), po.poseidon2_permute(s[0], params2))
    with pytest.raises(CenoHipError):
        bad = params[:64].copy()
        bad[3] = np.uint64(P)
        api.poseidon2_set_constants(dev, bad, None, None)
    api.poseidon2_set_constants(dev, None, None, None)
    d = _dev_tensor(s)
    api.poseidon2_permute(dev, d.data_ptr(), 1)
    dev.sync()
    assert np.array_equal(_to_np(d).reshape(-1), po.poseidon2_permute(s[0]))


def _codewords(mats, blow):
    """oracle-side codewords of padded traces: (width, rows << blow) column-major, bit-reversed DFT of the zero-extended column"""
    out, padded_all = [], []
    for m in mats:
        rows = 2
        while rows < m.shape[0]:
            rows *= 2
        padded = np.zeros((rows, m.shape[1]), dtype=np.uint64)
        padded[: m.shape[0]] = m
        padded_all.append(padded)
        out.append(np.stack([po.fft_bitrev(np.concatenate([padded[:, c], np.zeros(rows * ((1 << blow) - 1), dtype=np.uint64)]))
                             for c in range(m.shape[1])]))
    return out, padded_all


@pytest.mark.parametrize("blow", [1, 2])
def test_commit_traces_matches_oracle_composition(dev, blow):
    """ONE commitment for all matrices (scheme/cpu/mod.rs:559-584): root, every tree level's opened siblings and the opened rows
    equal the oracle's mixed-height MMCS over the oracle's codewords; equal heights, a 2-row matrix, ragged instance counts"""
    from ceno_amd import prover

    stream = dev.stream_create()
    mats = [po.rand_base(5 * 3, 1).reshape(5, 3), po.rand_base(64 * 22, 2).reshape(64, 22), po.rand_base(1 * 4, 3).reshape(1, 4),
            po.rand_base(8 * 2, 4).reshape(8, 2), po.rand_base(33 * 5, 5).reshape(33, 5), po.rand_base(16 * 1, 6).reshape(16, 1)]
    pcs = prover.PcsData(dev, mats, blow, stream)
    cws, padded = _codewords(mats, blow)
    for i, m in enumerate(mats):
        assert pcs.num_vars(i) == padded[i].shape[0].bit_length() - 1
        for c in (0, m.shape[1] - 1):  # witness MLE views are the (padded) columns
            assert np.array_equal(pcs.witness_mle(i, c).download(), padded[i][:, c])
    levels = po.mmcs_commit(cws)
    H = len(levels) - 1
    assert np.array_equal(pcs.root(), levels[-1][0])
    shapes = [(cw.shape[1].bit_length() - 1, cw.shape[0]) for cw in cws]
    for idx in sorted({0, (1 << H) - 1, (1 << H) // 3, 5, 64}):
        rows, path = pcs.open(idx)
        want_rows, want_path = po.mmcs_open(cws, levels, idx)
        assert np.array_equal(np.concatenate(rows), want_rows) and np.array_equal(path, want_path)
        assert po.mmcs_verify(shapes, levels[-1][0], idx, np.concatenate(rows), path) == 0
    pcs.free()
    dev.stream_destroy(stream)


def test_mmcs_commit_scattered_matrices_and_tall_injection(dev):
    """ceno_hip_mmcs_commit on matrices that are NOT stored back to back (one segment each), with matrices joining at levels
    above AND below the 2^14-node switch between the lane-per-node and the 8-lanes-per-node kernels; all levels vs the oracle"""
    import ctypes as C

    from tests import hipbuf as torch

    shapes = [(16, 2), (10, 3), (16, 1), (15, 2), (3, 5), (15, 1), (0, 2), (12, 7)]
    mats = [po.rand_base((1 << lr) * w, 40 + i).reshape(w, 1 << lr) for i, (lr, w) in enumerate(shapes)]
    d_mats = [torch.from_numpy(m.view(np.int64)).to("cuda:0") for m in mats]
    n = len(mats)
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in d_mats])
    lr = (C.c_int * n)(*[s_[0] for s_ in shapes])
    ws = (C.c_int * n)(*[s_[1] for s_ in shapes])
    h = C.c_void_p()
    dev.check(dev.L.ceno_hip_mmcs_commit(dev.h, ptrs, lr, ws, n, None, C.byref(h)))
    levels = po.mmcs_commit(mats)
    root = np.zeros(4, dtype=np.uint64)
    dev.check(dev.L.ceno_hip_merkle_root(dev.h, h, root.ctypes.data_as(C.POINTER(C.c_uint64)), None))
    assert np.array_equal(root, levels[-1][0])
    words = int(dev.L.ceno_hip_mmcs_opening_words(h))
    assert words == sum(w for _, w in shapes) + 4 * 16
    idx = np.array([0, 1, 65535, 40000, 12345, 32768], dtype=np.uint64)
    d_idx = torch.from_numpy(idx.view(np.int64)).to("cuda:0")
    d_out = torch.zeros(len(idx) * (words + 3), dtype=torch.int64, device="cuda:0")
    dev.check(dev.L.ceno_hip_mmcs_open_batch(dev.h, h, d_idx.data_ptr(), len(idx), 0, d_out.data_ptr(), words + 3, None))
    dev.sync()
    got = d_out.cpu().numpy().view(np.uint64).reshape(len(idx), words + 3)
    for q, i in enumerate(idx):
        rows, path = po.mmcs_open(mats, levels, int(i))
        assert np.array_equal(got[q, : len(rows)], rows), q
        assert np.array_equal(got[q, len(rows): words].reshape(-1, 4), path), q   # every level's sibling = the whole tree is right
        assert po.mmcs_verify(shapes, root, int(i), rows, path) == 0
    dev.check(dev.L.ceno_hip_merkle_free(dev.h, h))


def test_sharded_commit_virtual_ranks_equals_single_device(dev):
    """SURVEY §8(e) commit path: column-parallel encode, ONE all-to-all into row shards, local sub-trees, top levels
    from the gathered roots.  4 virtual ranks (threads, in-process collectives) on one GPU; ragged column split."""
    import threading

    import torch

    from ceno_amd import dist as cdist
    from ceno_amd import prover

    world, log_rows, blow = 4, 7, 1
    width_split = [3, 1, 4, 2]
    rows = 1 << log_rows
    full = po.rand_base(rows * sum(width_split), 77).reshape(rows, sum(width_split))
    stream = dev.stream_create()
    pcs = prover.PcsData(dev, [full], blow, stream)
    want = pcs.root()
    slots = {"g": [None] * world, "a2a": [None] * world}
    bar = threading.Barrier(world)

    def dist_for(rank):
        class D:
            def get_backend(self):
                return "threads"

            def all_gather(self, outs, t):
                slots["g"][rank] = t.clone()
                bar.wait()
                for g in range(world):
                    outs[g].copy_(slots["g"][g])
                bar.wait()

            def all_to_all(self, outs, ins):
                slots["a2a"][rank] = [x.clone() for x in ins]
                bar.wait()
                for g in range(world):
                    outs[g].copy_(slots["a2a"][g][rank])
                bar.wait()

        return D()

    res, errors = [None] * world, []

    def run(rank):
        try:
            c0 = sum(width_split[:rank])
            cols = np.ascontiguousarray(full[:, c0:c0 + width_split[rank]].T)
            res[rank] = cdist.sharded_commit(dev, cols, log_rows, blow, dist=dist_for(rank), world=world, rank=rank, stream=dev.stream_create())
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))
            bar.abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not errors, errors
    for r in range(world):
        assert np.array_equal(res[r]["root"], want)
    # rank g's rows are rows [g R/world, ...) of the single-device codeword: check one opened row per rank
    R = rows << blow
    for r in range(world):
        idx = r * (R // world) + 5
        row, _ = pcs.open_row(0, idx)
        local = res[r]["codeword_rows"].cpu().numpy().view(np.uint64).reshape(sum(width_split), R // world)
        assert np.array_equal(local[:, 5], row)
    pcs.free()


@pytest.mark.parametrize("world,width_split", [(4, [3, 1, 4, 2]), (2, [5, 0]), (8, [1, 2, 1, 3, 1, 1, 2, 1])])
def test_native_sharded_commit_local_group_equals_single_device(dev, world, width_split):
    """ceno_dist_commit_traces (C++ driver of the SURVEY section 8e commit path) with `world` virtual ranks = threads of this
    process, each on its own stream, exchanging through the in-process group: root, sub-tree roots and the local codeword
    rows equal the single-device commitment; ragged and EMPTY column shares"""
    import ctypes as C
    import threading

    import torch

    from ceno_amd import dist as cdist
    from ceno_amd import prover

    L = prover.plib()
    L.ceno_dist_local_group_create.restype = C.c_void_p
    L.ceno_dist_local_group_create.argtypes = [C.c_int]
    L.ceno_dist_local_group_destroy.argtypes = [C.c_void_p]
    L.ceno_dist_comm_init_local.restype = C.c_int
    L.ceno_dist_comm_init_local.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    log_rows, blow = 9, 1
    rows = 1 << log_rows
    wt = sum(width_split)
    full = po.rand_base(rows * wt, 91).reshape(rows, wt)
    stream = dev.stream_create()
    pcs = prover.PcsData(dev, [full], blow, stream)
    want = pcs.root()
    group = L.ceno_dist_local_group_create(world)
    assert group
    res, errors = [None] * world, []

    def run(rank):
        try:
            comm = C.c_void_p()
            assert L.ceno_dist_comm_init_local(group, rank, C.byref(comm)) == 0
            c0 = sum(width_split[:rank])
            cols = np.ascontiguousarray(full[:, c0:c0 + width_split[rank]].T)
            d_cols = torch.from_numpy(cols.view(np.int64).copy()).to("cuda:0") if cols.size else torch.empty(1, dtype=torch.int64, device="cuda:0")
            torch.cuda.synchronize()
            s = dev.stream_create()
            res[rank] = cdist.sharded_commit_native(dev, comm, d_cols.data_ptr(), width_split, log_rows, blow, rank, s)
            dev.sync(s)
            L.ceno_dist_comm_destroy(comm)
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=200)
    assert not errors, errors
    R = rows << blow
    for r in range(world):
        assert np.array_equal(res[r]["root"], want), r
        assert np.array_equal(res[r]["subtree_roots"], res[0]["subtree_roots"])
        idx = r * (R // world) + 3
        row, _ = pcs.open_row(0, idx)
        local = res[r]["codeword_rows"].cpu().numpy().view(np.uint64).reshape(wt, R // world)
        assert np.array_equal(local[:, 3], row)
        dev.check(dev.L.ceno_hip_merkle_free(dev.h, res[r]["subtree"]))
    L.ceno_dist_local_group_destroy(group)
    pcs.free()
    dev.stream_destroy(stream)


@pytest.mark.parametrize("world,shapes", [
    (4, [(7, [2, 1, 0, 3]), (5, [1, 1, 1, 1]), (7, [0, 2, 2, 0]), (1, [1, 0, 2, 0]), (0, [1, 1, 0, 0])]),
    (8, [(6, [1, 2, 0, 1, 1, 0, 2, 1]), (2, [1, 0, 0, 1, 0, 0, 0, 1]), (1, [0, 0, 3, 0, 0, 0, 0, 0]), (4, [1, 1, 1, 1, 1, 1, 1, 1])]),
    (2, [(3, [2, 3]), (8, [4, 1])]),
])
def test_native_sharded_mixed_height_commit_equals_single_device(dev, world, shapes):
    """ceno_dist_commit_traces_mmcs: several trace matrices of several heights, every matrix column-sharded over `world` virtual
    ranks (ragged, some ranks without columns), under ONE root — equal to the single-device commit_traces root (mixed-height MMCS)
    bit for bit.  Includes matrices whose codeword has exactly `world` rows (joins at the sub-tree roots) and fewer rows than
    ranks (gathered whole on every rank, joins in the replicated top levels); shapes = [(log2 trace rows, columns per rank)]"""
    import ctypes as C
    import threading

    import torch

    from ceno_amd import dist as cdist
    from ceno_amd import prover

    L = prover.plib()
    L.ceno_dist_local_group_create.restype = C.c_void_p
    L.ceno_dist_local_group_create.argtypes = [C.c_int]
    L.ceno_dist_local_group_destroy.argtypes = [C.c_void_p]
    L.ceno_dist_comm_init_local.restype = C.c_int
    L.ceno_dist_comm_init_local.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    blow = 1
    log_w = world.bit_length() - 1
    # log2 rows >= 1 for real traces (next_pow2_instance_padding >= 2); a 1-row "trace" (log 0) only reaches the C entry point directly
    fulls = [po.rand_base((1 << lr) * sum(ws), 500 + i).reshape(1 << lr, sum(ws)) for i, (lr, ws) in enumerate(shapes)]
    stream = dev.stream_create()
    # single-device reference: the same matrices through the MMCS primitive (commit_traces pads 1-row traces to 2, so the oracle /
    # primitive level is used here: RS-encode on the device, then ceno_hip_mmcs_commit)
    cws = []
    for (lr, ws), full in zip(shapes, fulls):
        w = sum(ws)
        d_cols = torch.from_numpy(np.ascontiguousarray(full.T).view(np.int64).copy()).to("cuda:0")
        d_cw = torch.empty(w << (lr + blow), dtype=torch.int64, device="cuda:0")
        dev.check(dev.L.ceno_hip_rs_encode(dev.h, d_cols.data_ptr(), lr, w, blow, d_cw.data_ptr(), stream))
        cws.append(d_cw)
    dev.sync(stream)
    n = len(shapes)
    h = C.c_void_p()
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in cws])
    lra = (C.c_int * n)(*[lr + blow for lr, _ in shapes])
    wsa = (C.c_int * n)(*[sum(ws) for _, ws in shapes])
    dev.check(dev.L.ceno_hip_mmcs_commit(dev.h, ptrs, lra, wsa, n, stream, C.byref(h)))
    want = np.zeros(4, dtype=np.uint64)
    dev.check(dev.L.ceno_hip_merkle_root(dev.h, h, want.ctypes.data_as(C.POINTER(C.c_uint64)), stream))
    host_cws = [t.cpu().numpy().view(np.uint64).reshape(sum(ws), 1 << (lr + blow)) for t, (lr, ws) in zip(cws, shapes)]
    assert np.array_equal(want, po.mmcs_commit(host_cws)[-1][0])   # and the oracle agrees with the single-device tree
    group = L.ceno_dist_local_group_create(world)
    assert group
    res, errors = [None] * world, []

    def run(rank):
        try:
            comm = C.c_void_p()
            assert L.ceno_dist_comm_init_local(group, rank, C.byref(comm)) == 0
            keep, ptrs_r = [], []
            for (lr, ws), full in zip(shapes, fulls):
                c0 = sum(ws[:rank])
                cols = np.ascontiguousarray(full[:, c0:c0 + ws[rank]].T)
                t = torch.from_numpy(cols.view(np.int64).copy()).to("cuda:0") if cols.size else torch.empty(1, dtype=torch.int64, device="cuda:0")
                keep.append(t)
                ptrs_r.append(t.data_ptr())
            torch.cuda.synchronize()
            s_ = dev.stream_create()
            res[rank] = cdist.sharded_commit_mmcs_native(dev, comm, ptrs_r, [ws for _, ws in shapes], [lr for lr, _ in shapes], blow, rank, s_)
            dev.sync(s_)
            L.ceno_dist_comm_destroy(comm)
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=200)
    assert not errors, errors
    for r in range(world):
        assert np.array_equal(res[r]["root"], want), r
        assert np.array_equal(res[r]["subtree_roots"], res[0]["subtree_roots"])
        for m, ((lr, ws), cw) in enumerate(zip(shapes, host_cws)):   # rank r's rows of every matrix = its slice of the codeword
            R = 1 << (lr + blow)
            local = res[r]["codeword_rows"][m].cpu().numpy().view(np.uint64)
            if lr + blow < log_w:
                assert np.array_equal(local.reshape(sum(ws), R), cw)
            else:
                rl = R // world
                assert np.array_equal(local.reshape(sum(ws), rl), cw[:, r * rl:(r + 1) * rl]), (r, m)
        dev.check(dev.L.ceno_hip_merkle_free(dev.h, res[r]["subtree"]))
        if res[r]["top"]:
            dev.check(dev.L.ceno_hip_merkle_free(dev.h, res[r]["top"]))
    dev.check(dev.L.ceno_hip_merkle_free(dev.h, h))
    L.ceno_dist_local_group_destroy(group)
    dev.stream_destroy(stream)


def test_native_sharded_commit_world1_through_rccl_self_exchange(dev, monkeypatch):
    """the RCCL arm of the same driver on a one-rank communicator: symbols resolve, and with CENO_DIST_SELF_P2P=1 the own block
    really goes through ncclSend / ncclRecv inside a group (the call sequence the multi-GPU run uses)"""
    import ctypes as C

    import torch

    from ceno_amd import dist as cdist
    from ceno_amd import prover

    log_rows, blow, w = 10, 1, 6
    full = po.rand_base((1 << log_rows) * w, 92).reshape(1 << log_rows, w)
    stream = dev.stream_create()
    pcs = prover.PcsData(dev, [full], blow, stream)
    want = pcs.root()
    comm = prover.RcclComm(1, 0, None)
    d_cols = torch.from_numpy(np.ascontiguousarray(full.T).view(np.int64).copy()).to("cuda:0")
    torch.cuda.synchronize()
    for self_p2p in ("0", "1"):
        monkeypatch.setenv("CENO_DIST_SELF_P2P", self_p2p)
        out = cdist.sharded_commit_native(dev, comm.h, d_cols.data_ptr(), [w], log_rows, blow, 0, stream)
        assert np.array_equal(out["root"], want), self_p2p
        row, _ = pcs.open_row(0, 77)
        local = out["codeword_rows"].cpu().numpy().view(np.uint64).reshape(w, -1)
        assert np.array_equal(local[:, 77], row)
        dev.check(dev.L.ceno_hip_merkle_free(dev.h, out["subtree"]))
    comm.close()
    pcs.free()
    dev.stream_destroy(stream)


def test_commit_of_several_matrices_on_a_fresh_context_equals_a_warm_one():
    """On a FRESH context the first transform of a size also builds that size's twiddle table (a per-context cache): the
    commitment of a shard's traces, two of them of one size, must not depend on whether the caches are warm (regression: a table
    used to be visible before it was complete) and must equal the oracle's mixed-height commitment"""
    from ceno_amd import Device, prover

    shapes = [(1 << 12, 5), (1 << 11, 3), (1 << 9, 4), (1 << 9, 4), (1 << 7, 2), (1 << 7, 6)]
    mats = [po.rand_base(r * w, 300 + i).reshape(r, w) for i, (r, w) in enumerate(shapes)]
    roots = []
    for warm in (False, True):
        d1 = Device(0)
        s1 = d1.stream_create()
        for _ in range(2 if warm else 1):
            together = prover.PcsData(d1, mats, 1, s1)
            root = together.root().copy()
            together.free()
        roots.append(root)
        d1.stream_destroy(s1)
        d1.close()
    assert np.array_equal(roots[0], roots[1])
    cws, _ = _codewords(mats, 1)
    assert np.array_equal(roots[0], po.mmcs_commit(cws)[-1][0])
