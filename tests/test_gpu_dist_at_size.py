"""The multi-rank paths AT SIZE on the one GPU (round-5 verdict item 2): the first run on eight GPUs must not be the first run at size.

* BASELINE config #4's headline — the nv = 26 hypercube split over 8 ranks (`ceno_dist_sumcheck_prove`, 2^23 elements per table and rank) — as 8
  PROCESSES sharing the device over the shared segment; every message, challenge and final evaluation against the ORACLE at full size (its
  AVX-512 dense sumcheck fed the challenges the ranks drew), not against another run of this library.
* BASELINE config #3's whole flow at 2^20 rows x 22 columns with the production block size q = 10 across 8 virtual ranks (in-process group):
  commitment -> chip proof -> main constraints -> opening, every rank's words equal to the single-device flow's.
* config #4's batched main-constraint sumcheck on the WIDE plan (48 chips, 22..96 columns) at max_nv = 20 across 8 virtual ranks.
Each writes what one rank put on the wire (exchanges, bytes) and the phase times to $CENO_DIST_STATS_OUT (one JSON line per case): the numbers of
DESIGN.md section 6 "per-rank critical path — PROJECTED, not a scaling curve".  8 virtual ranks share ONE device: the times say what a rank
computes and exchanges, not how eight devices scale."""
import json
import os
import tempfile
import threading
import time

import numpy as np
import pytest

from oracle import pyoracle as po
from tests.ranks import run_ranks

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def _emit(case, **kw):
    path = os.environ.get("CENO_DIST_STATS_OUT")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps(dict(case=case, **kw)) + "\n")


@pytest.mark.parametrize("world", [8, pytest.param(2, marks=pytest.mark.slow), pytest.param(4, marks=pytest.mark.slow)])
def test_nv26_hypercube_split_over_the_ranks_matches_the_oracle(world):
    """config #4's headline shape.  The ranks run the real engine on 2^(26 - log2 world)-element shards and exchange d partial evaluations per round
    through the shared segment; the oracle recomputes every round message of the WHOLE 2^26 hypercube from the challenges they drew."""
    n_total, k = 26, 3
    n_local = n_total - (world.bit_length() - 1)
    with tempfile.TemporaryDirectory() as tmp:
        run_ranks(world, [tmp, str(n_local), "shm_gpu"], extra_env={"CENO_TEST_DIST_REPS": "3"}, deadline_s=900)
        res = [dict(np.load(os.path.join(tmp, f"rank{r}.npz"))) for r in range(world)]
    for r in range(1, world):
        for key in ("msgs", "chal", "fin"):
            assert np.array_equal(res[r][key], res[0][key]), f"rank {r}: {key}"
    # the oracle at full size: 3 x 2^26 extension elements of the same SplitMix streams, the fused dense sumcheck on the ranks' challenges
    full = [po.fill_splitmix(2 << n_total, 0xCE10 + j, 0).reshape(-1, 2) for j in range(k)]
    t0 = time.time()
    omsgs, ofin = po.sumcheck_dense_mt(full, np.ascontiguousarray(res[0]["chal"]), threads=0, avx512=po.have_avx512())
    oracle_s = time.time() - t0
    assert np.array_equal(res[0]["msgs"], omsgs), "round messages"
    assert np.array_equal(res[0]["fin"], ofin), "final evaluations"
    wire = [int(x) for x in res[0]["wire"]]
    _emit("dist_sumcheck_nv26", world=world, n_local=n_local, wall_ms_per_sumcheck=[round(float(x) * 1e3, 3) for x in res[0]["wall_s"]],
          message_exchanges=wire[0], message_bytes_sent=wire[1], bulk_exchanges=wire[2], bulk_bytes_sent=wire[3], oracle_seconds=round(oracle_s, 2),
          transport="shared segment, 8 processes on ONE device" if world == 8 else f"shared segment, {world} processes on ONE device")


def test_config3_whole_flow_at_size_across_eight_virtual_ranks(dev, prover):
    """2^20 rows x 22 columns, 4 + 4 + 8 records, block size q = 10 (ceno_dist_chip_block_log's default), Poseidon2 transcript: commitment (column
    shards, re-shard by rows), chip proof (row shards), main constraints (same layout), opening — equal to the single-device flow word for word"""
    from tests.test_gpu_dist_gkr import whole_chip_flow

    stats = {}
    t0 = time.time()
    whole_chip_flow(dev, prover, 8, 20, 10, "poseidon2", w=22, stats=stats)
    _emit("config3_whole_flow", seconds_test=round(time.time() - t0, 1), **stats)
    assert stats["rank0_wire"]["message_exchanges"] > 0 and stats["rank0_wire"]["bulk_exchanges"] > 0


@pytest.mark.parametrize("max_nv", [20, pytest.param(22, marks=pytest.mark.slow)])
def test_wide_batched_main_constraints_at_size_across_eight_virtual_ranks(dev, prover, max_nv):
    """the wide plan of config #4 (48 chips of 2^(max_nv - 12) .. 2^max_nv rows, 22..96 columns, degree <= 5) over ROW-SHARDED tables, q = 10: chips of
    fewer than 2^(q + 4) rows ride along replicated; every rank's messages, point, evaluations and claimed sum equal the single-device proof's"""
    from ceno_amd import synthetic
    from tests.test_gpu_dist_gkr import comm_stats

    world, q, k = 8, 10, 3
    gch = [(11, 22), (33, 44)]
    jobs, chips, _ = synthetic.wide_batched_jobs(dev, max_nv)
    t0 = time.time()
    want = prover.prove_batched_main_constraints(dev, prover.MainJobs(jobs), gch, prover.Transcript.stub(5))
    single_s = time.time() - t0
    # row shards of every chip's columns, made on the device's host copy once (block-cyclic: index bits [q, q + k) name the rank)
    host_cols = [[m.download() for m in ch["cols"]] for ch in chips]
    group = prover.LocalGroup(world)
    results, errors, walls, wires = [None] * world, [], [0.0] * world, [None] * world

    def rank_main(g):
        try:
            st = dev.stream_create()
            local = []
            for j, ch, hc in zip(jobs, chips, host_cols):
                nv = ch["nv"]
                tabs = [dev.upload(prover.shard_rows(c_, world, g, q) if nv - k >= q + 1 else c_) for c_ in hc]
                local.append(dict(j, mles=tabs + [None] * ch["n_sel"]))
            mj = prover.MainJobs(local)
            dev.sync()
            t1 = time.time()
            results[g] = prover.dist_prove_batched_main_constraints(dev, group.comms[g], mj, gch, prover.Transcript.stub(5), q, st)
            dev.sync(st)
            walls[g] = time.time() - t1
            wires[g] = comm_stats(prover, group.comms[g])
            dev.stream_destroy(st)
        except Exception as e:  # noqa: BLE001
            import traceback

            errors.append((g, repr(e), traceback.format_exc(limit=3)))

    ths = [threading.Thread(target=rank_main, args=(g,)) for g in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(900)
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
    _emit("batched_main_wide", world=world, max_nv=max_nv, q=q, single_device_ms=round(single_s * 1e3, 2),
          wall_ms_eight_ranks_on_one_device=round(max(walls) * 1e3, 2), rank0_wire=wires[0])
    for ch in chips:
        for m in ch["cols"]:
            m.free()
