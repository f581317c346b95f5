"""GPU tests of metric M2's flow on a shard with the REFERENCE's population (ceno_amd/synthetic.py ShardFlowWide; round-5 verdict item 1):
45 opcode circuits with on-device witness generation writing inside the commitment's storage, two wide ECALL-class circuits, seven table
circuits with their `mlt` column built from the device lookup counters, a FIXED commitment opened beside the witness commitment
(ceno_zkvm/src/scheme/prover.rs:319-611, scheme/scheduler.rs:231-470, scheme/hal.rs:284-294; population: instructions/riscv/rv32im.rs:124-230,580-587).

* the witness of every opcode circuit and the multiplicity column of every table equal the oracle's CPU assignment of the same step records;
* chip proofs of an opcode circuit, a table with fixed columns and a table with structural columns equal the oracle's PROVER word for word;
* the whole flow at 2^10 and 2^12 cycles is accepted by a verifier assembled from the oracle's restated TowerVerify / sumcheck verifier /
  Basefold verifier on ONE transcript (what the prover binds, and in which order, is pinned by every later challenge);
* the proof does not depend on the number of lanes.
The commit / open path is PARITY UNPINNED against the reference (placeholder Poseidon2 constants)."""
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


def tup(a):
    return int(a[0]), int(a[1])


def e2_pow(a, k):
    r = (1, 0)
    for _ in range(k):
        r = po.e2_mul(r, a)
    return r


def _stub_absorb(t, word):
    """one absorb step of the SplitMix stub transcript (oracle/oracle.c orc_stub_*; host/transcript.cpp Stub::absorb)"""
    M = (1 << 64) - 1
    z = ((t.state.s ^ word) + 0x9E3779B97F4A7C15) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    t.state.s = z ^ (z >> 31)


def _oracle_witgen(ch, recs, idx, base_pc, slots):
    """the oracle's CPU assignment of one opcode circuit: (row-major n x w matrix, {table: counts})"""
    from ceno_amd.synthetic import NO_COLUMN

    w, call, a = ch["w"], ch["call"], ch["args"]
    nat = list(range(w))
    args = (recs, idx, 0, base_pc, slots)
    lk = {}
    if call == "arith":
        m, lk["dyn"], lk["fetch"] = po.witgen_arith(nat + [w], a[0], *args)
    elif call == "addi":
        m, lk["dyn"], lk["fetch"] = po.witgen_addi(nat + [w], *args)
    elif call == "logic_r":
        m, lk["dyn"], lk["fetch"], lk[("and", "or", "xor")[a[0]]] = po.witgen_logic_r(nat + [w], *args)
    elif call == "logic_i":
        m, lk["dyn"], lk["fetch"], lk[("and", "or", "xor")[a[0]]] = po.witgen_logic_i(nat + [w], *args)
    elif call == "lui":
        m, lk["dyn"], lk["fetch"] = po.witgen_lui(nat + [w], *args)
    elif call == "jal":
        m, lk["dyn"], lk["fetch"], lk["du8"], lk["xor"] = po.witgen_jal(nat + [w], *args)
    elif call == "auipc":
        m, lk["dyn"], lk["fetch"], lk["du8"], lk["xor"] = po.witgen_auipc(nat + [w], *args)
    elif call == "jalr":
        m, lk["dyn"], lk["fetch"] = po.witgen_jalr(nat + [w], *args)
    elif call == "slt":
        m, lk["dyn"], lk["fetch"] = po.witgen_slt(nat + [w], a[0], *args)
    elif call == "slti":
        m, lk["dyn"], lk["fetch"] = po.witgen_slti(nat + [w], a[0], *args)
    elif call == "branch":
        m, lk["dyn"], lk["fetch"] = po.witgen_branch(nat + [w], a[0], a[1], *args)
    elif call == "shift":
        m, lk["dyn"], lk["fetch"], lk["du8"], lk["xor"] = po.witgen_shift(nat + [w], a[0], a[1], *args)
    elif call == "mul":
        cols = nat[:22] + (nat[22:26] if a[0] else [NO_COLUMN] * 4) + [w]
        m, lk["dyn"], lk["fetch"] = po.witgen_mul(cols, a[0], *args)
    elif call == "div":
        m, lk["dyn"], lk["fetch"] = po.witgen_div(nat + [w], a[0], *args)
    elif call == "mem":
        m, lk["dyn"], lk["fetch"] = po.witgen_mem(nat + [w], a[0], *args)
    elif call == "load_sub":
        m, lk["dyn"], lk["fetch"] = po.witgen_load_sub(po.load_sub_cols(nat, a[0], a[1], w), a[0], a[1], *args)
    else:
        raise ValueError(call)
    return m, lk


def test_witness_of_every_circuit_and_table_multiplicities_match_the_oracle(dev, prover):
    """on-device witness generation straight into the commitment's storage, all chips counting into ONE session's lookup counters, the
    tables' `mlt` columns from the counters: every word equals the oracle's assignment of the same records (chip by chip, counts summed)"""
    from ceno_amd import synthetic

    flow = synthetic.ShardFlowWide(dev, prover, log_cycles=11, n_queries=10, pow_bits=4)
    recs = synthetic.synthetic_step_records(1 << 11, flow.FETCH_BASE_PC, flow.fetch_slots).view(np.uint8).reshape(1 << 11, 136)
    pcs = flow.generate_witness()
    dev.sync(flow.stream)
    totals = {k: np.zeros(v, dtype=np.uint64) for k, v in flow.counter_slots.items()}
    seen = 0
    for c, ch in enumerate(flow.chips):
        got = np.stack([pcs.witness_mle(c, j).download() for j in range(ch["w"])])  # (w, rows) column-major
        if ch["cls"] == "opcode":
            idx = ch["idx"].download().view(np.uint32)[: ch["n_inst"]]
            seen += len(idx)
            exp, lk = _oracle_witgen(ch, recs, idx, flow.FETCH_BASE_PC, flow.fetch_slots)
            assert np.array_equal(got[:, : ch["n_inst"]], exp.T), ch["name"]
            assert not got[:, ch["n_inst"]:].any(), ch["name"]                      # InstancePaddingStrategy::Default
            for k, v in lk.items():
                totals[k][: len(v)] += v.astype(np.uint64)
    assert seen == 1 << 11
    for c, ch in enumerate(flow.chips):
        if ch["cls"] == "table":
            got = pcs.witness_mle(c, 0).download()
            exp = np.zeros(ch["rows"], dtype=np.uint64)
            exp[: ch["n_inst"]] = totals[ch["src"]][: ch["n_inst"]]
            assert np.array_equal(got, exp), ch["name"]
    assert int(totals["dyn"].sum()) > 6 * (1 << 11) and int(totals["fetch"].sum()) == 1 << 11   # one fetch per step
    # the same witness, committed: the root equals the commitment of the downloaded matrices made the ordinary way (host row-major)
    pcs.finish()
    mats = [np.stack([pcs.witness_mle(c, j).download() for j in range(ch["w"])]).T.copy() for c, ch in enumerate(flow.chips)]
    ref = prover.PcsData(dev, mats, 1, flow.stream)
    assert np.array_equal(pcs.root(), ref.root())
    ref.free()
    pcs.free()
    flow.close()


def _oracle_chip_proof(cols, task, alpha, seed_transcript):
    """the oracle's restatement of create_chip_proof on host tables: wit_infer -> interleave -> towers -> out-evals into the transcript -> tower proof"""
    rows = cols[0].shape[0]
    log2_n = rows.bit_length() - 1
    coeffs, terms, out_terms = task["record_coeffs"], task["record_terms"], task["record_out_terms"]
    nr, nw, nlt, nl = task["num_reads"], task["num_writes"], task["num_lk_tables"], task["num_lk"]
    recs = [po.wit_infer(cols, np.ascontiguousarray(coeffs[ts[0]: ts[-1] + 1]), [terms[t] for t in ts], log2_n) for ts in out_terms]
    r_set, w_set = recs[:nr], recs[nr: nr + nw]
    lk_n, lk_d = recs[nr + nw: nr + nw + nlt], recs[nr + nw + nlt:]
    prod_specs, logup_specs, out_evals = [], [], []
    for group in (r_set, w_set):
        if group:
            limbs = po.interleaving_mles_to_mles(group, rows, 2, (1, 0))
            layers = po.infer_tower_product_witness(int(limbs[0].shape[0]).bit_length(), limbs)
            prod_specs.append(layers)
            out_evals += [layers[0][0][0], layers[0][1][0]]
    ql = po.interleaving_mles_to_mles(lk_d, rows, 2, alpha)
    pl = po.interleaving_mles_to_mles(lk_n, rows, 2, alpha) if lk_n else None
    layers = po.infer_tower_logup_witness(pl, ql)
    logup_specs.append(layers)
    out_evals += [layers[0][k][0] for k in range(4)]
    for e in out_evals:
        seed_transcript.append_ext(tup(e))
    return out_evals, po.tower_prove(prod_specs, logup_specs, seed_transcript)


def _verify_flow(flow, a, log_blowup, nq, pow_bits):
    """replay of the whole transcript with the oracle's verifiers"""
    chips = flow.chips
    vt = po.StubTranscript(0x5A)
    for root in a["roots"]:
        vt.append_ext((int(root[0]), int(root[1])))
        vt.append_ext((int(root[2]), int(root[3])))
    alpha, beta = vt.sample_ext(), vt.sample_ext()
    assert (alpha, beta) == (a["alpha"], a["beta"])
    for c, ch in enumerate(chips):
        proof, task = a["chip_proofs"][c], a["tasks"][c]
        ft = po.StubTranscript(0xF0)
        ft.append_ext(alpha)
        ft.append_ext(beta)
        for v in (c, c, ch["n_inst"], 0):
            _stub_absorb(ft, 0x4241534500000000)
            _stub_absorb(ft, v)
        for e in list(proof.r_out_evals) + list(proof.w_out_evals) + list(proof.lk_out_evals):
            ft.append_ext(tup(e))
        nv = ch["nv"]
        groups = [g for g in (task["num_reads"], task["num_writes"]) if g]
        n_lk = task["num_lk_tables"] or task["num_lk"]
        nvs = [nv + (max(1, g) - 1).bit_length() for g in groups] + [nv + (max(1, n_lk) - 1).bit_length()]
        op = po.TowerProof(proof.tower_num_vars, len(groups), 1)
        op.msgs[:] = proof.tower_msgs
        if groups:
            op.prod_evals[:] = proof.tower_prod_evals
        op.logup_evals[:] = proof.tower_logup_evals
        prod_out = np.concatenate([proof.r_out_evals, proof.w_out_evals]) if groups else np.zeros((0, 2), dtype=np.uint64)
        rc, vpoint, *_ = po.tower_verify(prod_out, proof.lk_out_evals, nvs, op, ft)
        assert rc == 0, ch["name"]
        assert np.array_equal(vpoint, proof.tower_point) and np.array_equal(proof.rt_main, proof.tower_point[-nv:]), ch["name"]
        assert ft.sample_ext() == a["fork_samples"][c]
    for s_ in a["fork_samples"]:
        vt.append_ext(s_)
    # ---- batched main sumcheck ----
    vt.append_label(b"combine subset evals")
    al = vt.sample_ext()
    ne = flow.n_exprs
    pows = [e2_pow(al, k) for k in range(ne * len(chips))]
    coeffs, terms, nvs, off = [], [], [], 0
    for c, ch in enumerate(chips):
        pl = a["plans"][c]
        start = len(nvs)
        nvs += [ch["nv"]] * (pl["n_cols"] + pl["n_sel"])
        chal = [alpha, beta] + pows[ne * c: ne * (c + 1)]
        for monos in pl["scalars"]:
            sc = (0, 0)
            for coeff, ids in monos:
                v = coeff
                for i in ids:
                    v = po.e2_mul(v, chal[i])
                sc = po.e2_add(sc, v)
            coeffs.append(sc)
        terms += [[start + j for j in t] for t in pl["terms"]]
    vpoint, expected = po.sumcheck_verify(a["claimed"], a["msgs"], vt)
    assert np.array_equal(vpoint, a["rt"])
    max_nv = max(ch["nv"] for ch in chips)
    assert expected == po.sumcheck_expected_from_evals(nvs, po.ext(coeffs), terms, max_nv, a["rt"], a["evals"])
    for c, ch in enumerate(chips):  # the selector evaluations the verifier computes itself (selector.rs:247-363)
        pl = a["plans"][c]
        for si, n_inst in enumerate(pl["sel_n_inst"]):
            got = tup(a["evals"][off + pl["n_cols"] + si])
            assert po.selector_evaluate(po.SEL_PREFIX, a["chip_proofs"][c].rt_main, a["rt"][: ch["nv"]], 0, n_inst) == got, ch["name"]
        off += pl["n_cols"] + pl["n_sel"]
    for e in a["evals"]:
        vt.append_ext(tup(e))
    # ---- ONE opening of the witness commitment and the fixed commitment ----
    shapes_w = [(ch["nv"], ch["w"]) for ch in chips]
    shapes_f = [(ch["nv"], ch["n_fixed"]) for ch in chips if ch["cls"] == "table" and ch["n_fixed"]]
    roots = np.stack([a["roots"][1], a["roots"][0]])     # batch_open's rounds: witness, then fixed
    assert po.basefold_verify(shapes_w + shapes_f, roots, a["points"], a["open_evals"], log_blowup, nq, pow_bits, vt, a["open_proof"],
                              commit_sizes=[len(shapes_w), len(shapes_f)]) == 0


@pytest.mark.parametrize("log_cycles,lanes", [(10, 1), (12, 4)])
def test_wide_shard_flow_end_to_end_verifies(dev, prover, log_cycles, lanes):
    from ceno_amd import synthetic

    nq, pow_bits = 12, 6
    flow = synthetic.ShardFlowWide(dev, prover, log_cycles=log_cycles, n_queries=nq, pow_bits=pow_bits)
    res = flow.run(lambda: prover.Transcript.stub(0x5A), lambda: prover.Transcript.stub(0xF0), lanes=lanes)
    assert res["n_chips"] == 45 + 2 + 7 and res["total_ms"] > 0
    a = flow.artifacts
    _verify_flow(flow, a, 1, nq, pow_bits)
    # ---- three chip proofs against the oracle's PROVER: an opcode circuit, a table with fixed columns, a table with structural columns ----
    by_name = {ch["name"]: c for c, ch in enumerate(flow.chips)}
    for name in ("SLLI", "XorTable", "DoubleU8") + (("Program",) if log_cycles == 10 else ()):
        c = by_name[name]
        task, proof = a["tasks"][c], a["chip_proofs"][c]
        cols = [m.download() for m in task["mles"]]
        ft = po.StubTranscript(0xF0)
        ft.append_ext(a["alpha"])
        ft.append_ext(a["beta"])
        for v in (c, c, flow.chips[c]["n_inst"], 0):
            _stub_absorb(ft, 0x4241534500000000)
            _stub_absorb(ft, v)
        out_evals, oproof = _oracle_chip_proof(cols, task, a["alpha"], ft)
        assert np.array_equal(proof.tower_msgs, oproof.msgs), name
        assert np.array_equal(proof.tower_point, oproof.point[: proof.tower_num_vars]), name
        assert np.array_equal(proof.tower_logup_evals, oproof.logup_evals), name
        if proof.n_prod:
            assert np.array_equal(proof.tower_prod_evals, oproof.prod_evals), name
        got_out = [x for x in proof.r_out_evals] + [x for x in proof.w_out_evals] + [x for x in proof.lk_out_evals]
        assert [tup(x) for x in got_out] == [tup(x) for x in out_evals], name
    # pool high-water of the chip-proof phase against what the scheduler booked at its busiest moment (scheduler.rs:342-347,622-652)
    assert res["chip_proofs_booked_high_water_bytes"] > 0
    assert res["chip_proofs_pool_high_water_bytes"] <= res["chip_proofs_booked_high_water_bytes"]
    flow.close()


def test_wide_shard_proof_does_not_depend_on_the_lanes(dev, prover):
    """forked transcripts make a chip proof a function of its task alone: 1, 4 and 16 lanes give the same proof, word for word (the
    scheduler runs at most CENO_HIP_MAX_LANES at once)"""
    from ceno_amd import synthetic

    flow = synthetic.ShardFlowWide(dev, prover, log_cycles=11, n_queries=8, pow_bits=4)
    ref = None
    for lanes in (1, 4, 16):
        flow.run(lambda: prover.Transcript.stub(0x5A), lambda: prover.Transcript.stub(0xF0), lanes=lanes)
        a = flow.artifacts
        cur = (a["roots"][1].tolist(), [p.tower_msgs.tolist() for p in a["chip_proofs"]], a["fork_samples"], a["msgs"].tolist(), a["evals"].tolist(),
               a["open_proof"].tolist())
        if ref is None:
            ref = cur
        assert cur == ref, lanes
    flow.close()


def _proof_words(a):
    per_chip = []
    for p in a["chip_proofs"]:
        per_chip.append((p.tower_num_vars, p.tower_msgs.tolist(), p.tower_point.tolist(), p.tower_logup_evals.tolist(),
                         p.tower_prod_evals.tolist() if p.n_prod else None, [tup(x) for x in p.r_out_evals], [tup(x) for x in p.w_out_evals],
                         [tup(x) for x in p.lk_out_evals], np.asarray(p.rt_main).tolist()))
    return (per_chip, a["fork_samples"], a["msgs"].tolist(), a["evals"].tolist(), a["open_proof"].tolist())


def test_cohort_layers_write_the_same_proofs(dev, prover, monkeypatch):
    """the middle tower layers of all chips proved together (host/cohort.cpp, csrc/tower_cohort.hip) against the per-chip prover
    (CENO_TOWER_COHORT_LAYERS=0), word for word over every chip proof and everything derived from them: cohorts up to 2^13 entries (one workgroup
    per chip), up to 2^16 (the default; a layer is cut into as many sub-cubes as the device holds at once, the host adds their partial messages
    and proves the last rounds) and 2^18, several launches per layer (a device that holds 24 workgroups), with the host layers moved so that
    the cohorts start at layer 5, with fixed sub-cube sizes (2^13: at most eight per layer; 2^6: thirty-two and five host rounds), and with the
    towers built from record tables instead of straight from the record expressions (CENO_TOWER_VIRTUAL_RECORDS=0) — and with the cohort phase
    FAILING while its first / fifth launch is served (CENO_TOWER_COHORT_FAIL_AT): every chip goes back to where it stood before the cohorts (state
    and transcript) and the lanes prove the rest: the same proofs again"""
    from ceno_amd import synthetic

    flow = synthetic.ShardFlowWide(dev, prover, log_cycles=12, n_queries=8, pow_bits=4)

    def run(env):
        for k in ("CENO_TOWER_COHORT_LAYERS", "CENO_TOWER_COHORT_CAPACITY", "CENO_TOWER_HOST_LAYERS", "CENO_TOWER_COHORT_SUB", "CENO_TOWER_VIRTUAL_RECORDS", "CENO_TOWER_COHORT_FAIL_AT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        flow.run(lambda: prover.Transcript.stub(0x5A), lambda: prover.Transcript.stub(0xF0), lanes=8)
        return _proof_words(flow.artifacts)

    ref = run({"CENO_TOWER_COHORT_LAYERS": "0"})
    assert max(p[0] for p in ref[0]) >= 18      # the tables' towers reach past every cohort layer tried here
    for env in ({}, {"CENO_TOWER_COHORT_LAYERS": "13"}, {"CENO_TOWER_COHORT_LAYERS": "18"}, {"CENO_TOWER_COHORT_CAPACITY": "24"},
                {"CENO_TOWER_HOST_LAYERS": "4"}, {"CENO_TOWER_COHORT_SUB": "13"}, {"CENO_TOWER_COHORT_SUB": "6", "CENO_TOWER_COHORT_LAYERS": "11"}, {"CENO_TOWER_VIRTUAL_RECORDS": "0"}, {"CENO_TOWER_COHORT_FAIL_AT": "0"}, {"CENO_TOWER_COHORT_FAIL_AT": "4"}):
        got = run(env)
        for c, (w, g) in enumerate(zip(ref[0], got[0])):
            assert w == g, (env, flow.chips[c]["name"])
        assert got[1:] == ref[1:], env
    flow.close()


def test_witgen_session_rejects_unregistered_tables_and_nests_cleanly(dev, prover):
    """a table named inside a session must have been registered; a second begin on the same context is refused; without a session the per-chip
    calls behave as before (they clear, merge and wait on their own)"""
    from ceno_amd import CenoHipError, api, synthetic

    n = 256
    recs = dev.upload(np.concatenate([synthetic.synthetic_step_records(n, 0x1000, 64).reshape(-1), np.zeros(8192 - 17 * n, dtype=np.uint64)]))
    iw = np.zeros(n // 2, dtype=np.uint64)
    iw.view(np.uint32)[:] = np.arange(n)
    idx = dev.upload(iw)
    st = dev.stream_create()
    w = dev.alloc(13, False)                                  # 22 columns x 256 rows
    dyn, fetch, other = dev.zeros(18, False, st), dev.zeros(5, False, st), dev.zeros(5, False, st)
    cols = list(range(22)) + [22]
    args = (recs.device_ptr, n, idx.device_ptr, n, w.device_ptr, n, 0, 0x1000, 64)
    # no session: counts land in the tables at once
    api.witgen_arith(dev, cols, False, *args, dyn.device_ptr, fetch.device_ptr, stream=st)
    base_d, base_f = dyn.download(st).copy(), fetch.download(st).copy()
    assert int(base_f.view(np.uint32).sum()) == n
    dev.witgen_session_begin([(dyn.device_ptr, 1 << 19), (fetch.device_ptr, 64)], st)
    with pytest.raises(CenoHipError):
        dev.witgen_session_begin([(dyn.device_ptr, 1 << 19)], st)
    with pytest.raises(CenoHipError):
        api.witgen_arith(dev, cols, False, *args, dyn.device_ptr, other.device_ptr, stream=st)   # `other` is not registered
    api.witgen_arith(dev, cols, False, *args, dyn.device_ptr, fetch.device_ptr, stream=st)
    api.witgen_arith(dev, cols, True, *args, dyn.device_ptr, fetch.device_ptr, stream=st)
    dev.witgen_session_end(st)
    got_f = fetch.download(st).view(np.uint32)
    assert np.array_equal(got_f, 3 * base_f.view(np.uint32))                                     # ADDED to, twice more
    assert int(dyn.download(st).view(np.uint32).astype(np.int64).sum()) > 2 * int(base_d.view(np.uint32).astype(np.int64).sum())
    dev.witgen_session_end(st)                                                                   # nothing open: a no-op
    for m in (recs, idx, w, dyn, fetch, other):
        m.free()
    dev.stream_destroy(st)
