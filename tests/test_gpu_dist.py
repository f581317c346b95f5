"""Multi-rank runs of the C++ sharded driver on ONE GPU: with the host shared-memory exchange there is no device
collective on the round path, so `world` processes can share device 0 — the real HIP engine, the real cross-process
exchange, the real replicated tail.  The result must equal the single-prover proof of the unsharded tables."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from tests.ranks import run_ranks

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,n_local", [(2, 9), (4, 7), (8, 5)])
def test_cpp_sharded_driver_shared_memory_exchange(world, n_local):
    with tempfile.TemporaryDirectory() as tmp:
        run_ranks(world, [tmp, str(n_local), "shm_gpu"], deadline_s=600)
        res = [np.load(os.path.join(tmp, f"rank{r}.npz")) for r in range(world)]
    k = 3
    n_total = n_local + world.bit_length() - 1
    full = [po.fill_splitmix(2 << n_total, 0xCE10 + j, 0).reshape(-1, 2) for j in range(k)]
    omsgs, ochal, ofin = po.sumcheck_prove(full, po.ext([1]), [list(range(k))], n_total, k, po.StubTranscript(0xF5))
    for r in range(world):
        assert np.array_equal(res[r]["msgs"], omsgs)
        assert np.array_equal(res[r]["chal"], ochal)
        assert np.array_equal(res[r]["fin"], ofin)


@pytest.mark.parametrize("world,n_total", [pytest.param(1, 7, marks=pytest.mark.slow), (2, 8), (4, 9)])
def test_cpp_batched_mixed_size_sharded_driver(world, n_total):
    """SURVEY section 8(e) mixed-size batches through the C++ driver (ceno_dist_batched_sumcheck_prove): classes sharded
    along their own top bits or replicated, `world` processes on one GPU, against the single-prover oracle proof"""
    from tests.dist_worker import batched_case

    with tempfile.TemporaryDirectory() as tmp:
        run_ranks(world, [tmp, str(n_total), "shm_gpu_batched"], deadline_s=600)
        res = [np.load(os.path.join(tmp, f"rank{r}.npz")) for r in range(world)]
    tables, coeffs, terms, off = [], [], [], 0
    for c in batched_case(n_total):
        tables += c["tables"]
        coeffs.append(c["coeffs"])
        terms += [[off + j for j in t] for t in c["terms"]]
        off += len(c["tables"])
    omsgs, ochal, ofin = po.sumcheck_prove(tables, np.concatenate(coeffs), terms, n_total, 3, po.StubTranscript(0xF5))
    for r in range(world):
        assert np.array_equal(res[r]["msgs"], omsgs)
        assert np.array_equal(res[r]["chal"], ochal)
        assert np.array_equal(res[r]["fin"], ofin)


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_commitment_and_opening_as_processes_on_one_gpu(world):
    """ceno_dist_commit_traces_mmcs + ceno_dist_basefold_open_mmcs as `world` PROCESSES sharing GPU 0: the re-shard of the codewords, the gathered
    batched codeword / polynomial and the query answers all travel through the shared segment's bulk area (RCCL does not run between ranks of one
    device); every rank's root and opening must equal the single-device commitment's, computed here, word for word"""
    from ceno_amd import Device, prover
    from oracle import pyoracle as po
    from tests.dist_worker import open_case

    with tempfile.TemporaryDirectory() as tmp:
        run_ranks(world, [tmp, "0", "shm_gpu_open"], deadline_s=600)
        res = [dict(np.load(os.path.join(tmp, f"rank{r}.npz"))) for r in range(world)]
    dev = Device(0)
    heights, col_split, fulls = open_case(world)
    stream = dev.stream_create()
    pcs = prover.PcsData(dev, fulls, 1, stream)
    point = np.array([[(i * 7919 + 13) % po.P, (i * 104729 + 17) % po.P] for i in range(max(heights))], dtype=np.uint64)
    points = [point[:h] for h in heights]
    evals = [np.array([po.mle_evaluate(np.ascontiguousarray(full[:, c]), points[m]) for c in range(full.shape[1])], dtype=np.uint64)
             for m, full in enumerate(fulls)]
    want_root = np.asarray(pcs.root(), dtype=np.uint64).reshape(-1)
    want = pcs.basefold_open(points, evals, 10, 3, prover.Transcript.poseidon2(b"open"))
    pcs.free()
    for r in range(world):
        assert np.array_equal(res[r]["root"].reshape(-1), want_root), f"rank {r}: root"
        assert res[r]["proof"].shape == want.shape and np.array_equal(res[r]["proof"], want), f"rank {r}: opening"
    dev.close()


@pytest.mark.parametrize("world,log2_n", [(2, 10), (4, 11)])
def test_row_sharded_chip_proof_as_processes_on_one_gpu(world, log2_n):
    """ceno_dist_create_chip_proof (record inference, tower witness, tower proof over row-sharded columns) and, on the same row shards and
    transcript, ceno_dist_prove_batched_main_constraints, as `world` PROCESSES sharing GPU 0 with the shared-memory exchange: every rank's proof
    and sumcheck must equal the single-device ones of the whole columns, computed here"""
    from ceno_amd import Device, prover
    from tests.dist_worker import chip_case

    with tempfile.TemporaryDirectory() as tmp:
        run_ranks(world, [tmp, str(log2_n), "shm_gpu_chip"], extra_env={"CENO_TEST_ROW_BLOCK_LOG": "3"}, deadline_s=600)
        res = [dict(np.load(os.path.join(tmp, f"rank{r}.npz"))) for r in range(world)]
    dev = Device(0)
    cols, coeffs, terms, out_terms, challenges, shape = chip_case(log2_n)
    full = [dev.upload(c) for c in cols]
    task = dict(mles=full, n_witin=len(cols), n_fixed=0, n_structural=0, num_instances=(1 << log2_n) - 5, log2_num_instances=log2_n, num_reads=shape[0],
                num_writes=shape[1], num_lk_tables=shape[2], num_lk=shape[3], record_coeffs=coeffs, record_terms=terms, record_out_terms=out_terms)
    from tests.dist_worker import chip_main_job

    tr = prover.Transcript.stub(21)
    want = prover.create_chip_proof(dev, task, challenges, tr)
    wc, wm, wrt, wev = prover.prove_batched_main_constraints(dev, [chip_main_job(full, log2_n, len(cols), want.rt_main)], challenges, tr)
    for r in range(world):
        got = res[r]
        assert np.array_equal(got["msgs"], want.tower_msgs) and np.array_equal(got["point"], want.tower_point)
        assert np.array_equal(got["prod"], want.tower_prod_evals) and np.array_equal(got["logup"], want.tower_logup_evals)
        assert np.array_equal(got["r_out"], want.r_out_evals) and np.array_equal(got["w_out"], want.w_out_evals)
        assert np.array_equal(got["lk_out"], want.lk_out_evals) and np.array_equal(got["rt_main"], want.rt_main)
        # ... and the main-constraint sumcheck on the same row shards and transcript
        assert tuple(int(x) for x in got["main_claim"]) == wc and np.array_equal(got["main_msgs"], wm), f"rank {r}: main constraints"
        assert np.array_equal(got["main_rt"], wrt) and np.array_equal(got["main_evals"], wev), f"rank {r}: main constraints"
    for m in full:
        m.free()
    dev.close()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_py_multi_rank_launch_end_to_end(world):
    """`bench.py --gpus N` exactly as the driver launches it (python -m torch.distributed.run, one rank per GPU), on a 1-GPU box:
    CENO_BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0 with a gloo process group, so the whole N > 1 flow runs — reference
    path through torch.distributed, the C++ driver with the shared-memory exchange validated against it, timing of every
    validated exchange, max over ranks, ONE JSON line from rank 0.  (RCCL cannot place two ranks on one device: it is reported
    as unavailable here and validated on the multi-GPU node.)"""
    import json

    env = dict(os.environ, CENO_BENCH_SINGLE_DEVICE="1", MASTER_ADDR="127.0.0.1", CENO_BENCH_DIST_CHIP_LOG_ROWS="15")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29700 + world), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--nv", "12"]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    # the headline is STRONG scaling (the nv-variable hypercube split over the ranks), the weak-scaling run is reported beside it
    log_w = world.bit_length() - 1
    assert r["n_gpus"] == world and r["steps"] == 3 and r["warmup"] == 1 and r["scaling"] == "strong"
    assert r["config"]["global_num_vars"] == 12 and r["config"]["num_vars_per_gpu"] == 12 - log_w
    assert r["metric"].endswith("nv=12")
    w = r["weak_scaling"]
    assert w["scaling"] == "weak" and w["config"]["global_num_vars"] == 12 + log_w and w["config"]["num_vars_per_gpu"] == 12
    assert w["value"] > 0 and abs(w["value"] - 9 * ((1 << (12 + log_w)) - 1) / (w["ms_per_step"] * 1e-3)) / w["value"] < 1e-6
    assert "shm" in r["exchanges_validated"] and r["collective_ms"]["shm"] > 0 and r["headline_exchange"] == "shm"  # (no RCCL with two ranks on one device)
    # RCCL either counted its ranks or fell back cleanly (here: several ranks on one device, so the fallback); the multi-rank commitment and
    # opening then run through the shared segment's host-staged bulk exchange, validated against the single-device root and opening
    assert r.get("rccl_ranks", world) == world and "rccl" not in r["exchanges_validated"]
    dc = r["extra"]["dist_commit"]
    assert dc["status"] == "ok" and dc["root_matches_single_device"] and dc["bulk_exchange"].startswith("shared segment"), dc
    assert dc["dist_open"]["status"] == "ok" and dc["dist_open"]["proof_equals_single_device"], dc["dist_open"]
    # the GKR half of a chip across the ranks (row-sharded record inference, towers, tower proof), validated against the single-device proof
    dcp = r["extra"]["dist_chip_proof"]
    assert dcp["status"] == "ok" and dcp["proof_equals_single_device"] and dcp["ms"] > 0, dcp
    assert dcp["main_constraints"]["equals_single_device"] and dcp["main_constraints"]["ms"] > 0, dcp  # ... and its main constraints on the same layout
    assert "shared-memory exchange" in r["config"]["collective"] and "checked against the torch.distributed path" in r["config"]["collective"]
    assert r["value"] > 0 and abs(r["value"] - 9 * ((1 << r["config"]["global_num_vars"]) - 1) / (r["ms_per_step"] * 1e-3)) / r["value"] < 1e-6


def test_bench_py_reports_an_rccl_arm_that_never_returns():
    """a collective that never returns (simulated: CENO_BENCH_FAKE_RCCL_HANG=1) must not take the N > 1 bench line along: the
    shared-memory measurement was completed before the RCCL arm was touched, every rank notices the timeout, rank 0 prints that line
    with the reason and the processes leave — a reported fallback, no hang past CENO_BENCH_RCCL_TIMEOUT_S"""
    import json

    env = dict(os.environ, CENO_BENCH_SINGLE_DEVICE="1", MASTER_ADDR="127.0.0.1", CENO_BENCH_FAKE_RCCL_HANG="1", CENO_BENCH_RCCL_TIMEOUT_S="3")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29733", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--nv", "12", "--scaling", "strong"]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["headline_exchange"] == "shm" and r["value"] > 0
    assert r["rccl"].startswith("unavailable: timeout") and r["extra"]["dist_commit"]["status"].startswith("skipped")
    assert r["degraded"] == "rccl_timeout"  # visible at the top level of the line, not only in the exit path


@pytest.mark.parametrize("world", [4])
def test_bench_py_plain_invocation_starts_its_own_ranks(world):
    """`python bench.py --gpus N` invoked exactly as `--gpus 1` is (no launcher, WORLD_SIZE unset): the process must start its N ranks as
    CHILD processes before anything touches the GPU, relay rank 0's single JSON line and its exit status"""
    import json

    env = dict(os.environ, CENO_BENCH_SINGLE_DEVICE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--nv", "12"]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == world and r["steps"] == 3 and r["scaling"] == "strong" and r["value"] > 0
    assert "shm" in r["exchanges_validated"]


def test_bench_py_plain_invocation_propagates_a_failing_rank():
    """a rank that dies (simulated: CENO_BENCH_FAIL_RANK) makes the launcher return that rank's exit status and print no JSON line; the
    surviving rank, stuck in the rendezvous, is ended after the grace period"""
    env = dict(os.environ, CENO_BENCH_SINGLE_DEVICE="1", CENO_BENCH_PEER_GRACE_S="5", CENO_BENCH_FAIL_RANK="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--nv", "12"]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 7, (out.returncode, out.stderr[-1000:])
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
