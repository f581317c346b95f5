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
