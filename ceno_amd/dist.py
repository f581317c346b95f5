"""Hypercube-sharded sumcheck over torch.distributed (RCCL on MI355X, gloo on CPU tests).

SURVEY.md §8(e): the reference has no intra-sumcheck distribution ("Distributed Sumcheck — TODO",
docs/src/optimizations.md:3-5); this is new design.  With LSB-first binding (pairs are adjacent
indices) splitting every table by its TOP log2(world) index bits keeps all folds local for the first
n_local = n - log2(world) rounds:

  rank g holds indices [g * 2^n_local, (g+1) * 2^n_local) of every table.

Per local round each rank produces d partial evaluations; one all-gather of world*d extension
elements (RCCL has no mod-p reduction, so partials are gathered and summed on the host mod p), the
transcript (replicated, deterministic) absorbs the message and yields the challenge.  After the local
rounds each rank owns one value per table; one more all-gather builds world-sized tables on every rank
and the last log2(world) rounds run replicated.  Collectives are latency-bound (world*d*16 bytes).

The per-shard work goes through a `ShardEngine`; the product engine is `HipShardEngine` (C ABI);
tests inject the CPU oracle as engine to exercise the collective logic under gloo without a GPU.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

P = 0xFFFFFFFF00000001
W = 7


def e2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def sum_partials(parts: np.ndarray) -> np.ndarray:
    """parts: (world, d, 2) uint64 -> (d, 2) modular sum"""
    world, d, _ = parts.shape
    out = np.zeros((d, 2), dtype=np.uint64)
    for t in range(d):
        c0 = sum(int(parts[g, t, 0]) for g in range(world)) % P
        c1 = sum(int(parts[g, t, 1]) for g in range(world)) % P
        out[t, 0], out[t, 1] = c0, c1
    return out


class ShardEngine:
    """what the sharded driver needs from a device back end"""

    def begin(self, n_local: int, degree: int):  # -> state
        raise NotImplementedError

    def round_partial(self, state, challenge) -> "object":
        """returns this shard's partial message as something `gather` understands"""
        raise NotImplementedError

    def finish(self, state, last_challenge) -> np.ndarray:  # (k, 2)
        raise NotImplementedError

    def tail(self, tables: List[np.ndarray], degree: int, transcript, msgs_out, chal_out, first_round: int):
        """run the replicated last rounds on world-sized tables; returns final evals (k,2)"""
        raise NotImplementedError


class HipShardEngine(ShardEngine):
    """product engine: one dense product term over k ext tables resident on this rank's GPU"""

    def __init__(self, dev, mles, coeff=(1, 0)):
        import torch

        from .api import Sumcheck

        self.dev, self.mles, self.coeff = dev, list(mles), coeff
        self.torch = torch
        self.Sumcheck = Sumcheck
        self.k = len(mles)

    def begin(self, n_local, degree):
        coeffs = np.array([[self.coeff[0], self.coeff[1]]], dtype=np.uint64)
        sc = self.Sumcheck(self.dev, self.mles, coeffs, [list(range(self.k))], n_local, degree)
        # device buffer for the partial message: int64 view of d x 2 uint64 words
        buf = self.torch.empty(degree * 2, dtype=self.torch.int64, device=f"cuda:{self.dev.device}")
        return {"sc": sc, "buf": buf}

    def round_partial(self, state, challenge):
        state["sc"].round_dev(challenge, state["buf"].data_ptr())
        # the library launches on its own stream: order the collective after it
        self.dev.sync()
        return state["buf"]

    def finish(self, state, last_challenge):
        fin = state["sc"].finish(last_challenge)
        state["sc"].free()
        return fin

    def tail(self, tables, degree, transcript, msgs_out, chal_out, first_round):
        from . import prover

        mles = [self.dev.upload(t) for t in tables]
        nv = int(tables[0].shape[0]).bit_length() - 1
        sc = self.Sumcheck(self.dev, mles, np.array([[self.coeff[0], self.coeff[1]]], dtype=np.uint64),
                           [list(range(self.k))], nv, degree)
        ch = None
        for r in range(nv):
            msg = sc.round(ch)
            ch = _absorb_round(transcript, msg)
            msgs_out[first_round + r] = msg
            chal_out[first_round + r] = ch
        fin = sc.finish(ch)
        sc.free()
        return fin


def _absorb_round(transcript, msg: np.ndarray) -> Tuple[int, int]:
    for t in range(msg.shape[0]):
        transcript.append_ext((int(msg[t, 0]), int(msg[t, 1])))
    transcript.append_label(b"Internal round")
    return transcript.sample_ext()


def _gather(dist, local, world: int, degree: int, device_tensor: bool) -> np.ndarray:
    import torch

    if device_tensor:
        out = torch.empty(world * degree * 2, dtype=torch.int64, device=local.device)
        dist.all_gather_into_tensor(out, local)
        return out.cpu().numpy().view(np.uint64).reshape(world, degree, 2)
    t = torch.from_numpy(np.ascontiguousarray(local).view(np.int64).reshape(-1))
    if dist.get_backend() == "nccl":  # RCCL moves device memory only
        t = t.cuda()
        out = torch.empty(world * t.numel(), dtype=torch.int64, device=t.device)
        dist.all_gather_into_tensor(out, t)
        return out.cpu().numpy().view(np.uint64).reshape(world, degree, 2)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return np.stack([o.numpy().view(np.uint64).reshape(degree, 2) for o in outs])


def sharded_sumcheck_prove(engine: ShardEngine, n_total: int, degree: int, transcript, dist=None, world: int = 1,
                           rank: int = 0):
    """Sumcheck of prod_k f_k over a 2^n_total hypercube sharded by top bits over `world` ranks.

    Transcript script is identical to IOPProverState::prove (n, d prologue; per round d evals,
    label, sample), so the proof equals the single-device proof of the unsharded tables.
    Returns (msgs (n,d,2), challenges (n,2), final_evals (k,2)) — identical on every rank."""
    assert world & (world - 1) == 0
    log_w = world.bit_length() - 1
    n_local = n_total - log_w
    assert n_local >= 0
    transcript.append_label(int(n_total).to_bytes(8, "little"))
    transcript.append_label(int(degree).to_bytes(8, "little"))
    msgs = np.zeros((n_total, degree, 2), dtype=np.uint64)
    chal = np.zeros((n_total, 2), dtype=np.uint64)
    state = engine.begin(n_local, degree)
    ch = None
    for r in range(n_local):
        part = engine.round_partial(state, ch)
        if world > 1:
            is_dev = hasattr(part, "data_ptr")
            parts = _gather(dist, part, world, degree, is_dev)
            msg = sum_partials(parts)
        else:
            msg = part.cpu().numpy().view(np.uint64).reshape(degree, 2) if hasattr(part, "data_ptr") else np.asarray(part)
        ch = _absorb_round(transcript, msg)
        msgs[r] = msg
        chal[r] = ch
    fin_local = engine.finish(state, ch)  # (k, 2): f_k restricted to this shard at (r_0..r_{n_local-1})
    k = fin_local.shape[0]
    if world == 1:
        return msgs, chal, fin_local
    # one value per table per rank -> world-sized tables, index = rank (the top bits)
    allv = _gather(dist, fin_local, world, k, False)  # (world, k, 2)
    tables = [np.ascontiguousarray(allv[:, j, :]) for j in range(k)]
    fin = engine.tail(tables, degree, transcript, msgs, chal, n_local)
    return msgs, chal, fin


# ==================================================================================================
# Mixed-size (front-loaded) batched sumcheck across ranks — SURVEY.md §8(e), BASELINE config #4
# ==================================================================================================
"""
`prove_batched_main_constraints` batches chips of different sizes in ONE sumcheck (front-load rule,
ceno_zkvm/src/scheme/verifier.rs:180-238): an MLE with n' < n variables is f(x_0..x_{n'-1}) * prod_{i>=n'} x_i.
The round polynomial is a plain sum over the indices of each size class, so every class can be split over
the ranks along the TOP bits of ITS OWN hypercube, independently of the other classes:

  * sharded class (n' - log2(world) local variables per rank): rounds 0..n'-log2(world)-1 are local folds and
    partial messages; then ONE all-gather of the k per-rank values turns the class into `world`-element
    tables, replicated on every rank, which continue as a replicated class;
  * replicated class (small chips: every rank holds the whole table and runs the same rounds): its message
    is added ONCE after the cross-rank sum.  Its engine is begun with `max_num_vars` = the rounds that are
    left, so the engine itself applies the front-load scalars once the class runs out of variables.

Per round: every live sharded class leaves d partial evaluations, ONE all-gather of (#live classes * d) ext
per rank, modular sum on the host, replicated transcript.  Nothing else crosses ranks.
"""


class BatchedEngine:
    """one size class on one rank (product: HipBatchedEngine over the C ABI; tests: the CPU oracle)"""

    def round(self, challenge) -> np.ndarray:  # (d, 2) host
        raise NotImplementedError

    def finish(self, last_challenge) -> np.ndarray:  # (k, 2) pure evaluations
        raise NotImplementedError

    def free(self):
        pass


class HipBatchedEngine(BatchedEngine):
    def __init__(self, dev, mles, coeffs, terms, max_num_vars, degree, stream=None):
        from .api import Sumcheck

        self.sc = Sumcheck(dev, mles, coeffs, terms, max_num_vars, degree, stream=stream)

    def round(self, challenge):
        return self.sc.round(challenge)

    def finish(self, last_challenge):
        return self.sc.finish(last_challenge)

    def free(self):
        self.sc.free()


def hip_engine_factory(dev, stream=None):
    """factory(tables_or_mles, coeffs, terms, max_num_vars, degree): numpy tables are uploaded, Mle handles borrowed"""

    def make(tables, coeffs, terms, max_num_vars, degree):
        mles = [t if hasattr(t, "h") else dev.upload(np.ascontiguousarray(t)) for t in tables]
        return HipBatchedEngine(dev, mles, coeffs, terms, max_num_vars, degree, stream)

    return make


def _gather_rows(dist, local: np.ndarray, world: int) -> np.ndarray:
    """all-gather a small (m, 2) uint64 array -> (world, m, 2)"""
    import torch

    flat = np.ascontiguousarray(local, dtype=np.uint64).reshape(-1)
    if world == 1 or dist is None:
        return flat.reshape(1, -1, 2)
    t = torch.from_numpy(flat.view(np.int64).copy())
    if dist.get_backend() == "nccl":
        t = t.cuda()
        out = torch.empty(world * t.numel(), dtype=torch.int64, device=t.device)
        dist.all_gather_into_tensor(out, t)
        return out.cpu().numpy().view(np.uint64).reshape(world, -1, 2)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return np.stack([o.numpy().view(np.uint64).reshape(-1, 2) for o in outs])


def _sum_mod(rows: np.ndarray) -> np.ndarray:
    """(a, m, 2) -> (m, 2) modular sum over axis 0"""
    a, m, _ = rows.shape
    out = np.zeros((m, 2), dtype=np.uint64)
    for j in range(m):
        out[j, 0] = sum(int(rows[g, j, 0]) for g in range(a)) % P
        out[j, 1] = sum(int(rows[g, j, 1]) for g in range(a)) % P
    return out


def sharded_batched_sumcheck_prove(factory, classes, n_total: int, degree: int, transcript, dist=None, world: int = 1,
                                   rank: int = 0):
    """classes: list of dicts
         {"tables": [k local tables or Mle handles], "num_vars": GLOBAL number of variables of the class,
          "sharded": bool, "coeffs": (T, 2) uint64, "terms": [[class-local MLE ids]]}
       A sharded class holds, on rank g, indices [g * 2^(nv - log2 world), ...) of each of its tables; a replicated
       class holds the whole tables on every rank.  All factors of a term belong to one class (reference rule,
       gkr_iop/src/gkr/layer/gpu/utils.rs:54-63).
       Returns (msgs (n_total, d, 2), challenges (n_total, 2), [final evals (k_c, 2) per class]) — identical on
       every rank and equal to the single-prover proof of the unsharded plan."""
    assert world & (world - 1) == 0
    log_w = world.bit_length() - 1
    transcript.append_label(int(n_total).to_bytes(8, "little"))
    transcript.append_label(int(degree).to_bytes(8, "little"))
    msgs = np.zeros((n_total, degree, 2), dtype=np.uint64)
    chal = np.zeros((n_total, 2), dtype=np.uint64)
    finals = [None] * len(classes)
    # segment = [class id, engine, kind, first round, last sharded round (exclusive)]
    segs = []
    pending_tail = {}  # class id -> local values (k, 2) of a sharded class that has no local variable at all
    for ci, c in enumerate(classes):
        nv = int(c["num_vars"])
        assert 0 < nv <= n_total or (nv == 0 and not c["sharded"])
        if c["sharded"] and world > 1:
            nv_local = nv - log_w
            assert nv_local >= 0, "a sharded class needs at least log2(world) variables"
            if nv_local == 0:
                vals = np.stack([np.asarray(t.download() if hasattr(t, "download") else t, dtype=np.uint64).reshape(-1, 2)[0]
                                 for t in c["tables"]])
                pending_tail[ci] = vals
            else:
                segs.append({"ci": ci, "eng": factory(c["tables"], c["coeffs"], c["terms"], nv_local, degree), "sharded": True,
                             "start": 0, "end": nv_local})
        else:
            segs.append({"ci": ci, "eng": factory(c["tables"], c["coeffs"], c["terms"], n_total, degree), "sharded": False,
                         "start": 0, "end": n_total})

    def start_tail(ci, local_vals, first_round):
        """per-rank values of a sharded class -> world-sized replicated tables from round `first_round` on"""
        allv = _gather_rows(dist, local_vals, world)  # (world, k, 2)
        k = allv.shape[1]
        tables = [np.ascontiguousarray(allv[:, j, :]) for j in range(k)]
        c = classes[ci]
        if first_round == n_total:  # nothing left to bind (cannot happen for world > 1)
            finals[ci] = allv[0]
            return
        segs.append({"ci": ci, "eng": factory(tables, c["coeffs"], c["terms"], n_total - first_round, degree), "sharded": False,
                     "start": first_round, "end": n_total})

    for ci, vals in sorted(pending_tail.items()):
        start_tail(ci, vals, 0)
    ch = None
    for i in range(n_total):
        live_sh = [s for s in segs if s["sharded"] and s["start"] <= i < s["end"]]
        live_rp = [s for s in segs if not s["sharded"] and s["start"] <= i < s["end"]]
        msg = np.zeros((degree, 2), dtype=np.uint64)
        if live_sh:
            parts = np.concatenate([s["eng"].round(None if i == s["start"] else ch) for s in live_sh])  # (#live * d, 2)
            msg = _sum_mod(_gather_rows(dist, parts, world).reshape(world * len(live_sh), degree, 2))
        for s in live_rp:
            m = s["eng"].round(None if i == s["start"] else ch)
            msg = _sum_mod(np.stack([msg, m]))
        ch = _absorb_round(transcript, msg)
        msgs[i] = msg
        chal[i] = ch
        for s in live_sh:
            if s["end"] == i + 1:  # last local variable bound: one value per table per rank
                fin_local = s["eng"].finish(ch)
                s["eng"].free()
                start_tail(s["ci"], fin_local, i + 1)
    for s in segs:
        if not s["sharded"]:
            finals[s["ci"]] = s["eng"].finish(ch)
            s["eng"].free()
    return msgs, chal, finals


# ==================================================================================================
# Trace commitment across ranks — SURVEY.md §8(e) "Commit path"
# ==================================================================================================
def sharded_commit(dev, local_columns: np.ndarray, log_rows: int, log_blowup: int, dist=None, world: int = 1, rank: int = 0,
                   stream=None):
    """Column-parallel RS encoding + row-sharded Merkle leaves.

    `local_columns`: this rank's columns of the (padded) trace, shape (w_local, 2^log_rows) — columns are
    independent, so every rank encodes its own (NTT, no exchange).  A Merkle leaf hashes one codeword ROW across ALL
    columns and the sponge is sequential along the row, so the codeword is re-sharded by rows with ONE all-to-all
    (the only step of the whole path whose cost is xGMI bandwidth: (world-1)/world of the codeword bytes leave each
    rank); rank g then owns rows [g R/world, (g+1) R/world) = the sub-tree under node g of level log2(world) from the
    top.  The `world` sub-tree roots are all-gathered and the top levels finished on every rank.
    Returns {"root", "subtree" (api.Merkle over the local rows), "subtree_roots" (world, 4), "codeword_rows" (tensor:
    (w_total, R/world) column-major int64 view of the local rows), "widths"}.  Equal to the single-device commitment."""
    import torch

    from . import api

    assert world & (world - 1) == 0
    log_w = world.bit_length() - 1
    cols = np.ascontiguousarray(local_columns, dtype=np.uint64)
    w_local, rows = cols.shape
    assert rows == 1 << log_rows and log_rows + log_blowup >= log_w
    R = rows << log_blowup
    device = f"cuda:{dev.device}"
    d_cols = torch.from_numpy(cols.view(np.int64)).to(device)
    d_cw = torch.empty(w_local * R, dtype=torch.int64, device=device)
    api.rs_encode(dev, d_cols.data_ptr(), log_rows, w_local, log_blowup, d_cw.data_ptr(), stream)
    dev.sync(stream)
    if world == 1:
        widths = [w_local]
        recv = d_cw
    else:
        wt = torch.tensor([w_local], dtype=torch.int64, device=device)
        wl = [torch.empty_like(wt) for _ in range(world)]
        dist.all_gather(wl, wt)
        widths = [int(x.item()) for x in wl]
        rl = R // world
        # destination h gets rows [h*rl, (h+1)*rl) of each of my columns, column-major
        send = d_cw.view(w_local, world, rl).permute(1, 0, 2).contiguous()
        outs = [torch.empty(widths[g] * rl, dtype=torch.int64, device=device) for g in range(world)]
        dist.all_to_all(outs, [send[h].reshape(-1) for h in range(world)])
        recv = torch.cat(outs)  # source ranks in order = global column order
        torch.cuda.current_stream().synchronize()  # torch's stream produced `recv`; the library reads it on ITS stream
    w_total = sum(widths)
    rl = R // world
    sub = api.Merkle(dev, recv.data_ptr(), log_rows + log_blowup - log_w, w_total, stream)
    sub_root = sub.root(stream)
    if world == 1:
        return {"root": sub_root, "subtree": sub, "subtree_roots": sub_root.reshape(1, 4), "codeword_rows": recv, "widths": widths}
    roots = _gather_rows(dist, sub_root.reshape(2, 2), world).reshape(world, 4)
    level = torch.from_numpy(roots.view(np.int64).copy()).to(device)
    while level.shape[0] > 1:  # top log2(world) levels: node = permute(left || right)[0..4)
        st = level.reshape(-1, 8).contiguous()
        torch.cuda.current_stream().synchronize()
        api.poseidon2_permute(dev, st.data_ptr(), st.shape[0], stream)
        dev.sync(stream)
        level = st[:, :4].contiguous()
    root = level.cpu().numpy().view(np.uint64).reshape(4)
    return {"root": root, "subtree": sub, "subtree_roots": roots, "codeword_rows": recv, "widths": widths}


def sharded_commit_native(dev, comm_handle, d_cols_ptr: int, widths, log_rows: int, log_blowup: int, rank: int, stream):
    """The same commitment through the C++ driver `ceno_dist_commit_traces` (ceno_amd/host/dist.cpp): encode, ONE
    all-to-all of unequal blocks (RCCL grouped send/recv, or device copies inside a local group), local sub-tree, top levels.
    `comm_handle`: a ceno_dist_comm* (prover.RcclComm(...).h, or a local-group communicator).  Returns
    {"root", "subtree" (ceno_hip_merkle*), "subtree_roots", "codeword_rows" (int64 tensor, (w_total, R/world) column-major)}."""
    import ctypes as C

    import torch

    from . import prover
    from .api import CenoHipError

    L = prover.plib()
    world = len(widths)
    L.ceno_dist_commit_traces.restype = C.c_int
    L.ceno_dist_commit_traces.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    R = 1 << (log_rows + log_blowup)
    w_total = int(sum(widths))
    rows = torch.empty(w_total * (R // world), dtype=torch.int64, device=f"cuda:{dev.device}")
    roots = np.zeros((world, 4), dtype=np.uint64)
    root = np.zeros(4, dtype=np.uint64)
    sub = C.c_void_p()
    wa = (C.c_int * world)(*[int(w) for w in widths])
    u64p = C.POINTER(C.c_uint64)
    rc = L.ceno_dist_commit_traces(dev.h, comm_handle, C.c_void_p(d_cols_ptr), wa, log_rows, log_blowup, stream, C.c_void_p(rows.data_ptr()),
                                   C.byref(sub), roots.ctypes.data_as(u64p), root.ctypes.data_as(u64p))
    if rc != 0:
        raise CenoHipError(rc, (L.ceno_dist_last_error() or b"").decode())
    return {"root": root, "subtree": sub, "subtree_roots": roots, "codeword_rows": rows, "rank": rank}


def sharded_commit_mmcs_native(dev, comm_handle, d_cols_ptrs, widths, log_rows, log_blowup: int, rank: int, stream):
    """Several matrices of several heights under ONE root across ranks (`ceno_dist_commit_traces_mmcs`, ceno_amd/host/dist.cpp): the
    multi-rank form of commit_traces, equal to the single-device mixed-height commitment bit for bit.  widths[m][g] = columns of matrix
    m on rank g; d_cols_ptrs[m] = this rank's columns of matrix m (device pointer, column-major).  Returns {"root", "subtree", "top"
    (ceno_hip_merkle* or None), "subtree_roots", "codeword_rows": per matrix an int64 tensor ((total width) x rows of this rank)}."""
    import ctypes as C

    import torch

    from . import prover
    from .api import CenoHipError

    L = prover.plib()
    n = len(log_rows)
    world = len(widths[0])
    log_w = world.bit_length() - 1
    L.ceno_dist_commit_traces_mmcs.restype = C.c_int
    L.ceno_dist_commit_traces_mmcs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_int,
                                               C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                               C.POINTER(C.c_uint64)]
    outs = []
    for m in range(n):
        R = 1 << (log_rows[m] + log_blowup)
        rl = R if log_rows[m] + log_blowup < log_w else R // world
        outs.append(torch.empty(max(1, int(sum(widths[m])) * rl), dtype=torch.int64, device=f"cuda:{dev.device}"))
    roots = np.zeros((world, 4), dtype=np.uint64)
    root = np.zeros(4, dtype=np.uint64)
    sub, top = C.c_void_p(), C.c_void_p()
    lr = (C.c_int * n)(*[int(x) for x in log_rows])
    wa = (C.c_int * (n * world))(*[int(w) for row in widths for w in row])
    cp = (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in d_cols_ptrs])
    op = (C.c_void_p * n)(*[C.c_void_p(t.data_ptr()) for t in outs])
    u64p = C.POINTER(C.c_uint64)
    rc = L.ceno_dist_commit_traces_mmcs(dev.h, comm_handle, n, lr, wa, cp, log_blowup, stream, op, C.byref(sub), C.byref(top),
                                        roots.ctypes.data_as(u64p), root.ctypes.data_as(u64p))
    if rc != 0:
        raise CenoHipError(rc, (L.ceno_dist_last_error() or b"").decode())
    return {"root": root, "subtree": sub, "top": top if top else None, "subtree_roots": roots, "codeword_rows": outs, "rank": rank}
