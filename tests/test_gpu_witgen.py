"""GPU parity for on-device witness generation of the ADD / SUB chips: the HIP kernel through the C ABI
against the oracle's CPU assignment, bit-exact, on the reference test's step data (chips/add.rs:62-100,119-188:
GPU column-major witness == CPU row-major witness, lookup multiplicities equal)."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests import witgen_cases as wc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from ceno_amd import Device

    d = Device(0)
    yield d
    d.close()


def _to_dev(a):
    from tests import hipbuf as torch

    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def _run(dev, cols, sub, recs, idx, rows, offset, base_pc, slots, lk=True, poison=True):
    from tests import hipbuf as torch

    from ceno_amd import api

    d_recs = _to_dev(recs.reshape(-1))
    d_idx = _to_dev(np.asarray(idx, dtype=np.uint32).view(np.int32))
    num_cols = int(cols[22])
    w = torch.full((num_cols * rows,), -1 if poison else 0, dtype=torch.int64, device="cuda:0")
    lkd = torch.zeros(1 << 19, dtype=torch.int32, device="cuda:0")
    lkf = torch.zeros(max(slots, 1), dtype=torch.int32, device="cuda:0")
    api.witgen_arith(dev, cols, sub, d_recs.data_ptr(), recs.shape[0], d_idx.data_ptr(), len(idx), w.data_ptr(), rows, offset, base_pc, slots,
                     lkd.data_ptr() if lk else 0, lkf.data_ptr() if lk else 0)
    dev.sync()
    return (w.cpu().numpy().view(np.uint64).reshape(num_cols, rows), lkd.cpu().numpy().view(np.uint32), lkf.cpu().numpy().view(np.uint32)[:slots])


@pytest.mark.parametrize("sub", [False, True])
@pytest.mark.parametrize("n,rows", [(1024, 1024), (1000, 1024), (1, 2), (300, 512)])
def test_witness_and_lookups_match_cpu_assignment(dev, sub, n, rows):
    d = wc.reference_test_steps(n, sub)
    recs = po.step_records_r(d["cycles"], d["pcs"], po.INSN_SUB if sub else po.INSN_ADD, 2, 3, 4, d["rs1_vals"], d["rs2_vals"], d["rd_before"],
                             d["rd_after"], d["prev_cycles"])
    idx = np.arange(n)
    got, lkd, lkf = _run(dev, wc.NATURAL_COLS, sub, recs, idx, rows, 0, 0x1000, n)
    exp, elkd, elkf = po.witgen_arith(wc.NATURAL_COLS, sub, recs, idx, 0, 0x1000, n)
    assert np.array_equal(got[:, :n], exp.T)          # assert_witness_colmajor_eq (test_helpers.rs:8-35)
    assert not got[:, n:].any()                       # InstancePaddingStrategy::Default
    assert np.array_equal(lkd, elkd) and np.array_equal(lkf, elkf)


def test_permuted_columns_subset_of_steps_and_shard_offset(dev):
    rng = np.random.default_rng(9)
    n_steps = 5000
    d = wc.reference_test_steps(n_steps)
    d["rs1_vals"] = rng.integers(0, 1 << 32, n_steps, dtype=np.uint64)
    d["rs2_vals"] = rng.integers(0, 1 << 32, n_steps, dtype=np.uint64)
    d["rd_before"] = rng.integers(0, 1 << 32, n_steps, dtype=np.uint64)
    d["rd_after"] = (d["rs1_vals"] + d["rs2_vals"]) & np.uint64(0xFFFFFFFF)
    offset = 1 << 20
    d["cycles"] = d["cycles"] + offset
    d["prev_cycles"] = rng.integers(0, 1 << 21, n_steps, dtype=np.uint64)   # before the shard, inside it, and "later" (invalid but assigned)
    d["prev_cycles"][::5] = 0
    recs = po.step_records_r(d["cycles"], d["pcs"], po.INSN_ADD, 7, 31, 0, d["rs1_vals"], d["rs2_vals"], d["rd_before"], d["rd_after"], d["prev_cycles"])
    cols = list(rng.permutation(40)[:22]) + [40]
    idx = rng.permutation(n_steps)[:3000]
    got, lkd, lkf = _run(dev, cols, False, recs, idx, 4096, offset, 0x1000, n_steps, poison=False)
    exp, elkd, elkf = po.witgen_arith(cols, False, recs, idx, offset, 0x1000, n_steps)
    assert np.array_equal(got[:, :3000], exp.T) and not got[:, 3000:].any()
    assert np.array_equal(lkd, elkd) and np.array_equal(lkf, elkf)
    # counters accumulate across calls (one table per shard serves every chip) and may be omitted
    got2, _, _ = _run(dev, cols, False, recs, idx, 4096, offset, 0x1000, n_steps, lk=False, poison=False)
    assert np.array_equal(got2, got)


def test_chip_flow_size_properties(dev):
    """2^20 instances (BASELINE config #3 shape): the chip's constraints hold on the device output, lookup totals add up"""
    from tests import hipbuf as torch

    from ceno_amd import api

    n = 1 << 20
    rng = np.random.default_rng(3)
    d = dict(cycles=4 + 4 * np.arange(n, dtype=np.uint64), pcs=0x2000 + 4 * (np.arange(n, dtype=np.uint64) % 4096),
             rs1_vals=rng.integers(0, 1 << 32, n, dtype=np.uint64), rs2_vals=rng.integers(0, 1 << 32, n, dtype=np.uint64),
             rd_before=rng.integers(0, 1 << 32, n, dtype=np.uint64))
    d["rd_after"] = (d["rs1_vals"] + d["rs2_vals"]) & np.uint64(0xFFFFFFFF)
    # records built with numpy in the emulator's layout (the per-record C helper is for small cases)
    rec = np.zeros((n, 17), dtype=np.uint64)
    rec[:, 0] = d["cycles"]
    rec[:, 1] = d["pcs"] | ((d["pcs"] + 4) << np.uint64(32))
    rec[:, 4] = np.uint64(1 | (2 << 8) | (3 << 16) | (4 << 24))
    rec[:, 5] = np.uint64(0x00010101) << np.uint64(32)
    rec[:, 6] = np.uint64((2 << 8) // 4) | (d["rs1_vals"] << np.uint64(32))
    rec[:, 8] = np.uint64((3 << 8) // 4) | (d["rs2_vals"] << np.uint64(32))
    rec[:, 10] = np.uint64((4 << 8) // 4) | (d["rd_before"] << np.uint64(32))
    rec[:, 11] = d["rd_after"]
    small = po.step_records_r(d["cycles"][:3], d["pcs"][:3], po.INSN_ADD, 2, 3, 4, d["rs1_vals"][:3], d["rs2_vals"][:3], d["rd_before"][:3],
                              d["rd_after"][:3], np.zeros(3))
    assert np.array_equal(rec[:3].view(np.uint8).reshape(3, 136)[:, :128], small[:, :128])
    recs = rec.view(np.uint8).reshape(n, 136)
    got, lkd, lkf = _run(dev, wc.NATURAL_COLS, False, recs, np.arange(n), n, 0, 0x2000, 4096)
    m = got.astype(np.int64)
    rd = d["rd_after"].astype(np.int64)
    assert np.array_equal(m[16] + m[18], (rd & 0xFFFF) + (m[20] << 16))
    assert np.array_equal(m[17] + m[19] + m[20], (rd >> 16) + (m[21] << 16))
    for base, diff0, sub_cycle in ((3, 4, 0), (7, 8, 1), (11, 14, 2)):
        assert np.array_equal(m[base] - (m[1] + sub_cycle), m[diff0] + (m[diff0 + 1] << 16) - (1 << 29))
    assert int(lkd.astype(np.int64).sum()) == 8 * n and np.all(lkf == n // 4096)
    assert int(lkd[(1 << 16):].astype(np.int64).sum()) == 5 * n and int(lkd[(1 << 13):(1 << 14)].astype(np.int64).sum()) == 3 * n


def test_bad_arguments_fail_loudly(dev):
    from ceno_amd import CenoHipError

    d = wc.reference_test_steps(8)
    recs = po.step_records_r(d["cycles"], d["pcs"], po.INSN_ADD, 2, 3, 4, d["rs1_vals"], d["rs2_vals"], d["rd_before"], d["rd_after"], d["prev_cycles"])
    dup = list(range(22)) + [22]
    dup[5] = dup[4]
    with pytest.raises(CenoHipError):
        _run(dev, dup, False, recs, np.arange(8), 8, 0, 0x1000, 8)
    with pytest.raises(CenoHipError):
        _run(dev, list(range(22)) + [21], False, recs, np.arange(8), 8, 0, 0x1000, 8)
    with pytest.raises(CenoHipError):
        _run(dev, wc.NATURAL_COLS, False, recs, np.arange(8), 4, 0, 0x1000, 8)


def _run_logic(dev, cols, kind, recs, idx, rows, offset, base_pc, slots, lk=True):
    from tests import hipbuf as torch

    from ceno_amd import api

    d_recs = _to_dev(recs.reshape(-1))
    d_idx = _to_dev(np.asarray(idx, dtype=np.uint32).view(np.int32))
    num_cols = int(cols[28])
    w = torch.full((num_cols * rows,), -1, dtype=torch.int64, device="cuda:0")
    lkd = torch.zeros(1 << 19, dtype=torch.int32, device="cuda:0")
    lkf = torch.zeros(max(slots, 1), dtype=torch.int32, device="cuda:0")
    lkl = torch.zeros(1 << 16, dtype=torch.int32, device="cuda:0")
    api.witgen_logic_r(dev, cols, kind, d_recs.data_ptr(), recs.shape[0], d_idx.data_ptr(), len(idx), w.data_ptr(), rows, offset, base_pc, slots,
                       lkd.data_ptr() if lk else 0, lkf.data_ptr() if lk else 0, lkl.data_ptr() if lk else 0)
    dev.sync()
    return (w.cpu().numpy().view(np.uint64).reshape(num_cols, rows), lkd.cpu().numpy().view(np.uint32), lkf.cpu().numpy().view(np.uint32)[:slots],
            lkl.cpu().numpy().view(np.uint32))


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("n,rows", [(1024, 1024), (1000, 1024), (1, 2)])
def test_logic_witness_and_lookups_match_cpu_assignment(dev, kind, n, rows):
    """AND / OR / XOR on the reference test's step data (chips/logic_r.rs:85-118,140-170): GPU column-major witness == CPU row-major
    witness, the dynamic-range, fetch and operation-table multiplicities equal"""
    d = wc.reference_logic_steps(n, kind)
    recs = po.step_records_r(d["cycles"], d["pcs"], (po.INSN_AND, po.INSN_OR, po.INSN_XOR)[kind], 2, 3, 4, d["rs1_vals"], d["rs2_vals"],
                             d["rd_before"], d["rd_after"], d["prev_cycles"])
    idx = np.arange(n)
    got, lkd, lkf, lkl = _run_logic(dev, wc.LOGIC_NATURAL_COLS, kind, recs, idx, rows, 0, 0x1000, n)
    exp, elkd, elkf, elkl = po.witgen_logic_r(wc.LOGIC_NATURAL_COLS, recs, idx, 0, 0x1000, n)
    assert np.array_equal(got[:, :n], exp.T)
    assert not got[:, n:].any()
    assert np.array_equal(lkd, elkd) and np.array_equal(lkf, elkf) and np.array_equal(lkl, elkl)


def test_logic_permuted_columns_subset_of_steps_and_shard_offset(dev):
    rng = np.random.default_rng(11)
    n_steps = 5000
    d = wc.reference_logic_steps(n_steps, 2)
    d["rs1_vals"] = rng.integers(0, 1 << 32, n_steps, dtype=np.uint64)
    d["rs2_vals"] = rng.integers(0, 1 << 32, n_steps, dtype=np.uint64)
    d["rd_before"] = rng.integers(0, 1 << 32, n_steps, dtype=np.uint64)
    d["rd_after"] = d["rs1_vals"] ^ d["rs2_vals"]
    offset = 1 << 20
    d["cycles"] = d["cycles"] + offset
    d["prev_cycles"] = rng.integers(0, 1 << 21, n_steps, dtype=np.uint64)
    d["prev_cycles"][::5] = 0
    recs = po.step_records_r(d["cycles"], d["pcs"], po.INSN_XOR, 7, 31, 0, d["rs1_vals"], d["rs2_vals"], d["rd_before"], d["rd_after"], d["prev_cycles"])
    cols = list(rng.permutation(40)[:28]) + [40]
    idx = rng.permutation(n_steps)[:3000]
    got, lkd, lkf, lkl = _run_logic(dev, cols, 2, recs, idx, 4096, offset, 0x1000, n_steps)
    exp, elkd, elkf, elkl = po.witgen_logic_r(cols, recs, idx, offset, 0x1000, n_steps)
    mapped = sorted(cols[:28])
    assert np.array_equal(got[mapped, :3000], exp.T[mapped]) and not got[mapped, 3000:].any()
    unmapped = [c for c in range(40) if c not in mapped]
    assert np.all(got[unmapped] == np.uint64(0xFFFFFFFFFFFFFFFF))  # columns outside the map are the caller's: left untouched
    assert np.array_equal(lkd, elkd) and np.array_equal(lkf, elkf) and np.array_equal(lkl, elkl)
    got2, _, _, _ = _run_logic(dev, cols, 2, recs, idx, 4096, offset, 0x1000, n_steps, lk=False)
    assert np.array_equal(got2, got)


def test_logic_bad_arguments_fail_loudly(dev):
    from ceno_amd import CenoHipError

    d = wc.reference_logic_steps(8)
    recs = po.step_records_r(d["cycles"], d["pcs"], po.INSN_AND, 2, 3, 4, d["rs1_vals"], d["rs2_vals"], d["rd_before"], d["rd_after"], d["prev_cycles"])
    dup = list(range(28)) + [28]
    dup[20] = dup[3]
    with pytest.raises(CenoHipError):
        _run_logic(dev, dup, 0, recs, np.arange(8), 8, 0, 0x1000, 8)
    with pytest.raises(CenoHipError):
        _run_logic(dev, list(range(28)) + [27], 0, recs, np.arange(8), 8, 0, 0x1000, 8)
    with pytest.raises(CenoHipError):
        _run_logic(dev, wc.LOGIC_NATURAL_COLS, 3, recs, np.arange(8), 8, 0, 0x1000, 8)


def _run_addi(dev, cols, recs, idx, rows, offset, base_pc, slots, lk=True):
    from tests import hipbuf as torch

    from ceno_amd import api

    d_recs = _to_dev(recs.reshape(-1))
    d_idx = _to_dev(np.asarray(idx, dtype=np.uint32).view(np.int32))
    num_cols = int(cols[18])
    w = torch.full((num_cols * rows,), -1, dtype=torch.int64, device="cuda:0")
    lkd = torch.zeros(1 << 19, dtype=torch.int32, device="cuda:0")
    lkf = torch.zeros(max(slots, 1), dtype=torch.int32, device="cuda:0")
    api.witgen_addi(dev, cols, d_recs.data_ptr(), recs.shape[0], d_idx.data_ptr(), len(idx), w.data_ptr(), rows, offset, base_pc, slots,
                    lkd.data_ptr() if lk else 0, lkf.data_ptr() if lk else 0)
    dev.sync()
    return w.cpu().numpy().view(np.uint64).reshape(num_cols, rows), lkd.cpu().numpy().view(np.uint32), lkf.cpu().numpy().view(np.uint32)[:slots]


@pytest.mark.parametrize("n,rows", [(1024, 1024), (1000, 1024), (1, 2), (2100, 4096)])
def test_addi_witness_and_lookups_match_cpu_assignment(dev, n, rows):
    """ADDI on the reference test's step data (chips/addi.rs:80-97,115-146): positive and negative immediates"""
    d = wc.reference_addi_steps(n)
    recs = po.step_records_i(d["cycles"], d["pcs"], po.INSN_ADDI, 2, 4, d["imms"], d["rs1_vals"], d["rd_before"], d["rd_after"], d["prev_cycles"])
    idx = np.arange(n)
    got, lkd, lkf = _run_addi(dev, wc.ADDI_NATURAL_COLS, recs, idx, rows, 0, 0x1000, n)
    exp, elkd, elkf = po.witgen_addi(wc.ADDI_NATURAL_COLS, recs, idx, 0, 0x1000, n)
    assert np.array_equal(got[:, :n], exp.T) and not got[:, n:].any()
    assert np.array_equal(lkd, elkd) and np.array_equal(lkf, elkf)


def test_addi_permuted_columns_random_operands_and_shard_offset(dev):
    rng = np.random.default_rng(12)
    n_steps = 5000
    d = wc.reference_addi_steps(n_steps)
    d["rs1_vals"] = rng.integers(0, 1 << 32, n_steps, dtype=np.uint64)
    d["imms"] = rng.integers(-2048, 2048, n_steps, dtype=np.int64)
    d["rd_before"] = rng.integers(0, 1 << 32, n_steps, dtype=np.uint64)
    d["rd_after"] = ((d["rs1_vals"].astype(np.int64) + d["imms"]) & 0xFFFFFFFF).astype(np.uint64)
    offset = 1 << 20
    d["cycles"] = d["cycles"] + offset
    d["prev_cycles"] = rng.integers(0, 1 << 21, n_steps, dtype=np.uint64)
    d["prev_cycles"][::5] = 0
    recs = po.step_records_i(d["cycles"], d["pcs"], po.INSN_ADDI, 9, 17, d["imms"], d["rs1_vals"], d["rd_before"], d["rd_after"], d["prev_cycles"])
    cols = list(rng.permutation(30)[:18]) + [30]
    idx = rng.permutation(n_steps)[:3000]
    got, lkd, lkf = _run_addi(dev, cols, recs, idx, 4096, offset, 0x1000, n_steps)
    exp, elkd, elkf = po.witgen_addi(cols, recs, idx, offset, 0x1000, n_steps)
    mapped = sorted(cols[:18])
    assert np.array_equal(got[mapped, :3000], exp.T[mapped]) and not got[mapped, 3000:].any()
    assert np.array_equal(lkd, elkd) and np.array_equal(lkf, elkf)
    with pytest.raises(Exception):
        _run_addi(dev, list(range(18)) + [17], recs, idx, 4096, offset, 0x1000, n_steps)


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("n,rows", [(1024, 1024), (1, 2), (600, 1024)])
def test_logic_i_witness_and_lookups_match_cpu_assignment(dev, kind, n, rows):
    """ANDI / ORI / XORI on the reference test's step data (chips/logic_i.rs:79-100) plus sign-extended negative immediates"""
    from tests import hipbuf as torch

    from ceno_amd import api

    d = wc.reference_logic_i_steps(n, kind)
    d["imms"][n // 2:] = -(d["imms"][n // 2:] % 2048) - 1
    d["rd_after"] = wc.LOGIC_OPS[kind](d["rs1_vals"], d["imms"].astype(np.uint64) & np.uint64(0xFFFFFFFF))
    recs = po.step_records_i(d["cycles"], d["pcs"], (po.INSN_ANDI, po.INSN_ORI, po.INSN_XORI)[kind], 2, 4, d["imms"], d["rs1_vals"], d["rd_before"],
                             d["rd_after"], d["prev_cycles"])
    rng = np.random.default_rng(13)
    cols = list(rng.permutation(30)[:24]) + [30]
    idx = np.arange(n)
    d_recs = _to_dev(recs.reshape(-1))
    d_idx = _to_dev(idx.astype(np.uint32).view(np.int32))
    w = torch.full((30 * rows,), -1, dtype=torch.int64, device="cuda:0")
    lkd = torch.zeros(1 << 19, dtype=torch.int32, device="cuda:0")
    lkf = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    lkl = torch.zeros(1 << 16, dtype=torch.int32, device="cuda:0")
    api.witgen_logic_i(dev, cols, kind, d_recs.data_ptr(), n, d_idx.data_ptr(), n, w.data_ptr(), rows, 0, 0x1000, n, lkd.data_ptr(), lkf.data_ptr(),
                       lkl.data_ptr())
    dev.sync()
    got = w.cpu().numpy().view(np.uint64).reshape(30, rows)
    exp, elkd, elkf, elkl = po.witgen_logic_i(cols, recs, idx, 0, 0x1000, n)
    mapped = sorted(cols[:24])
    assert np.array_equal(got[mapped, :n], exp.T[mapped]) and not got[mapped, n:].any()
    assert np.array_equal(lkd.cpu().numpy().view(np.uint32), elkd) and np.array_equal(lkf.cpu().numpy().view(np.uint32), elkf)
    assert np.array_equal(lkl.cpu().numpy().view(np.uint32), elkl)


@pytest.mark.parametrize("n,rows", [(1024, 1024), (1, 2), (500, 512)])
def test_lui_witness_and_lookups_match_cpu_assignment(dev, n, rows):
    """LUI on step data shaped like the reference's test (chips/lui.rs:79-97), immediates over the whole 20-bit range"""
    from tests import hipbuf as torch

    from ceno_amd import api
    from tests.test_oracle_witgen import _lui_steps

    d = _lui_steps(n)
    recs = po.step_records_i(d["cycles"], d["pcs"], po.INSN_LUI, 0, 4, d["imms"], d["rs1_vals"], d["rd_before"], d["rd_after"], d["prev_cycles"])
    rng = np.random.default_rng(15)
    cols = list(rng.permutation(22)[:16]) + [22]
    idx = np.arange(n)
    d_recs = _to_dev(recs.reshape(-1))
    d_idx = _to_dev(idx.astype(np.uint32).view(np.int32))
    w = torch.full((22 * rows,), -1, dtype=torch.int64, device="cuda:0")
    lkd = torch.zeros(1 << 19, dtype=torch.int32, device="cuda:0")
    lkf = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    api.witgen_lui(dev, cols, d_recs.data_ptr(), n, d_idx.data_ptr(), n, w.data_ptr(), rows, 0, 0x1000, n, lkd.data_ptr(), lkf.data_ptr())
    dev.sync()
    got = w.cpu().numpy().view(np.uint64).reshape(22, rows)
    exp, elkd, elkf = po.witgen_lui(cols, recs, idx, 0, 0x1000, n)
    mapped = sorted(cols[:16])
    assert np.array_equal(got[mapped, :n], exp.T[mapped]) and not got[mapped, n:].any()
    assert np.array_equal(lkd.cpu().numpy().view(np.uint32), elkd) and np.array_equal(lkf.cpu().numpy().view(np.uint32), elkf)


@pytest.mark.parametrize("chip", ["jal", "auipc"])
@pytest.mark.parametrize("n,rows", [(1024, 1024), (1, 2), (400, 512)])
def test_jal_and_auipc_witness_and_lookups_match_cpu_assignment(dev, chip, n, rows):
    """JAL (13 columns) and AUIPC (21 columns): dynamic-range, fetch, double-byte and XOR table multiplicities"""
    from tests import hipbuf as torch

    from ceno_amd import api
    from tests.test_oracle_witgen import _auipc_steps, _jal_steps

    rng = np.random.default_rng(18)
    if chip == "jal":
        d = _jal_steps(n)
        recs = po.step_records_j(d["cycles"], d["pcs"], d["pcs_after"], po.INSN_JAL, 4, d["imms"], d["rd_before"], d["rd_after"], d["prev_cycles"])
        ncol, fn_gpu, fn_cpu = 13, api.witgen_jal, po.witgen_jal
    else:
        d = _auipc_steps(n)
        recs = po.step_records_i(d["cycles"], d["pcs"], po.INSN_AUIPC, 0, 4, d["imms"], d["rs1_vals"], d["rd_before"], d["rd_after"], d["prev_cycles"])
        ncol, fn_gpu, fn_cpu = 21, api.witgen_auipc, po.witgen_auipc
    total = ncol + 5
    cols = list(rng.permutation(total)[:ncol]) + [total]
    idx = np.arange(n)
    slots = 1 << 12
    d_recs = _to_dev(recs.reshape(-1))
    d_idx = _to_dev(idx.astype(np.uint32).view(np.int32))
    w = torch.full((total * rows,), -1, dtype=torch.int64, device="cuda:0")
    tabs = [torch.zeros(sz, dtype=torch.int32, device="cuda:0") for sz in (1 << 19, slots, 1 << 16, 1 << 16)]
    fn_gpu(dev, cols, d_recs.data_ptr(), n, d_idx.data_ptr(), n, w.data_ptr(), rows, 0, 0x1000, slots, *[t.data_ptr() for t in tabs])
    dev.sync()
    got = w.cpu().numpy().view(np.uint64).reshape(total, rows)
    exp, *etabs = fn_cpu(cols, recs, idx, 0, 0x1000, slots)
    mapped = sorted(cols[:ncol])
    assert np.array_equal(got[mapped, :n], exp.T[mapped]) and not got[mapped, n:].any()
    for t, e in zip(tabs, etabs):
        assert np.array_equal(t.cpu().numpy().view(np.uint32), e)


@pytest.mark.parametrize("signed", [True, False])
@pytest.mark.parametrize("n,rows", [(1024, 1024), (1, 2), (700, 1024)])
def test_slt_witness_and_lookups_match_cpu_assignment(dev, signed, n, rows):
    """SLT / SLTU on the reference test's step data (chips/slt.rs:94-118) plus sign / equality / limb-boundary edge cases; the negative top
    limbs are Goldilocks field elements p - (2^16 - limb)"""
    from tests import hipbuf as torch

    from ceno_amd import api
    from tests.test_oracle_witgen import _slt_steps

    d = _slt_steps(n, signed)
    recs = po.step_records_r(d["cycles"], d["pcs"], po.INSN_SLT if signed else po.INSN_SLTU, 2, 3, 4, d["rs1_vals"], d["rs2_vals"], d["rd_before"],
                             d["rd_after"], d["prev_cycles"])
    rng = np.random.default_rng(20)
    cols = list(rng.permutation(30)[:26]) + [30]
    idx = np.arange(n)
    d_recs = _to_dev(recs.reshape(-1))
    d_idx = _to_dev(idx.astype(np.uint32).view(np.int32))
    w = torch.full((30 * rows,), -1, dtype=torch.int64, device="cuda:0")
    lkd = torch.zeros(1 << 19, dtype=torch.int32, device="cuda:0")
    lkf = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    api.witgen_slt(dev, cols, signed, d_recs.data_ptr(), n, d_idx.data_ptr(), n, w.data_ptr(), rows, 0, 0x1000, n, lkd.data_ptr(), lkf.data_ptr())
    dev.sync()
    got = w.cpu().numpy().view(np.uint64).reshape(30, rows)
    exp, elkd, elkf = po.witgen_slt(cols, signed, recs, idx, 0, 0x1000, n)
    mapped = sorted(cols[:26])
    assert np.array_equal(got[mapped, :n], exp.T[mapped]) and not got[mapped, n:].any()
    assert np.array_equal(lkd.cpu().numpy().view(np.uint32), elkd) and np.array_equal(lkf.cpu().numpy().view(np.uint32), elkf)


@pytest.mark.parametrize("signed", [True, False])
@pytest.mark.parametrize("n,rows", [(1024, 1024), (1, 2), (600, 1024)])
def test_slti_witness_and_lookups_match_cpu_assignment(dev, signed, n, rows):
    from tests import hipbuf as torch

    from ceno_amd import api
    from tests.test_oracle_witgen import _slti_steps

    d = _slti_steps(n, signed)
    recs = po.step_records_i(d["cycles"], d["pcs"], po.INSN_SLTI if signed else po.INSN_SLTIU, 2, 4, d["imms"], d["rs1_vals"], d["rd_before"], d["rd_after"],
                             d["prev_cycles"])
    rng = np.random.default_rng(21)
    cols = list(rng.permutation(27)[:22]) + [27]
    idx = np.arange(n)
    d_recs = _to_dev(recs.reshape(-1))
    d_idx = _to_dev(idx.astype(np.uint32).view(np.int32))
    w = torch.full((27 * rows,), -1, dtype=torch.int64, device="cuda:0")
    lkd = torch.zeros(1 << 19, dtype=torch.int32, device="cuda:0")
    lkf = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    api.witgen_slti(dev, cols, signed, d_recs.data_ptr(), n, d_idx.data_ptr(), n, w.data_ptr(), rows, 0, 0x1000, n, lkd.data_ptr(), lkf.data_ptr())
    dev.sync()
    got = w.cpu().numpy().view(np.uint64).reshape(27, rows)
    exp, elkd, elkf = po.witgen_slti(cols, signed, recs, idx, 0, 0x1000, n)
    mapped = sorted(cols[:22])
    assert np.array_equal(got[mapped, :n], exp.T[mapped]) and not got[mapped, n:].any()
    assert np.array_equal(lkd.cpu().numpy().view(np.uint32), elkd) and np.array_equal(lkf.cpu().numpy().view(np.uint32), elkf)


@pytest.mark.parametrize("kind", ["BEQ", "BNE", "BLT", "BGE", "BLTU", "BGEU"])
@pytest.mark.parametrize("n,rows", [(1024, 1024), (1, 2), (500, 512)])
def test_branch_witness_and_lookups_match_cpu_assignment(dev, kind, n, rows):
    """the six branches: comparison gadget (BLT / BGE / BLTU / BGEU) or field-inverse equality marker (BEQ / BNE); the branch offset as a field element"""
    from tests import hipbuf as torch

    from ceno_amd import api
    from tests.test_oracle_witgen import _branch_steps

    is_eq = kind in ("BEQ", "BNE")
    flag = kind == "BEQ" if is_eq else kind in ("BLT", "BGE")
    d = _branch_steps(n, kind)
    recs = po.step_records_b(d["cycles"], d["pcs"], d["pcs_after"], getattr(po, "INSN_" + kind), 2, 3, d["imms"], d["rs1_vals"], d["rs2_vals"], d["prev_cycles"])
    nc = 19 if is_eq else 22
    rng = np.random.default_rng(22)
    cols = list(rng.permutation(nc + 4)[:nc]) + [nc + 4]
    idx = np.arange(n)
    d_recs = _to_dev(recs.reshape(-1))
    d_idx = _to_dev(idx.astype(np.uint32).view(np.int32))
    w = torch.full(((nc + 4) * rows,), -1, dtype=torch.int64, device="cuda:0")
    lkd = torch.zeros(1 << 19, dtype=torch.int32, device="cuda:0")
    lkf = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    api.witgen_branch(dev, cols, is_eq, flag, d_recs.data_ptr(), n, d_idx.data_ptr(), n, w.data_ptr(), rows, 0, 0x2000, n, lkd.data_ptr(), lkf.data_ptr())
    dev.sync()
    got = w.cpu().numpy().view(np.uint64).reshape(nc + 4, rows)
    exp, elkd, elkf = po.witgen_branch(cols, is_eq, flag, recs, idx, 0, 0x2000, n)
    mapped = sorted(cols[:nc])
    assert np.array_equal(got[mapped, :n], exp.T[mapped]) and not got[mapped, n:].any()
    assert np.array_equal(lkd.cpu().numpy().view(np.uint32), elkd) and np.array_equal(lkf.cpu().numpy().view(np.uint32), elkf)


@pytest.mark.parametrize("is_store", [False, True])
@pytest.mark.parametrize("n,rows,offset", [(1024, 1024, 0), (1, 2, 0), (500, 512, 0), (300, 512, 2)])
def test_lw_sw_witness_and_lookups_match_cpu_assignment(dev, is_store, n, rows, offset):
    """LW / SW: register reads / write, the memory access with its own timestamp comparison, the address limbs with their 14-bit range lookups"""
    from tests import hipbuf as torch

    from ceno_amd import api
    from tests.test_oracle_witgen import _mem_records, _mem_steps

    d = _mem_steps(n, is_store)
    if offset:
        d["cycles"] = d["cycles"] + np.uint64(1000)
        d["prev_cycles"][::3] = 500                      # earlier in this shard
        d["prev_cycles"][1::3] = 1                       # before the shard began: aligned to 0
    recs = _mem_records(d, is_store)
    nc = 23
    rng = np.random.default_rng(23 + is_store)
    cols = list(rng.permutation(nc + 5)[:nc]) + [nc + 5]
    idx = rng.permutation(n) if offset else np.arange(n)
    d_recs = _to_dev(recs.reshape(-1))
    d_idx = _to_dev(idx.astype(np.uint32).view(np.int32))
    w = torch.full(((nc + 5) * rows,), -1, dtype=torch.int64, device="cuda:0")
    lkd = torch.zeros(1 << 19, dtype=torch.int32, device="cuda:0")
    lkf = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    off = 996 if offset else 0
    api.witgen_mem(dev, cols, is_store, d_recs.data_ptr(), n, d_idx.data_ptr(), n, w.data_ptr(), rows, off, 0x1000, n, lkd.data_ptr(), lkf.data_ptr())
    dev.sync()
    got = w.cpu().numpy().view(np.uint64).reshape(nc + 5, rows)
    exp, elkd, elkf = po.witgen_mem(cols, is_store, recs, idx, off, 0x1000, n)
    mapped = sorted(cols[:nc])
    assert np.array_equal(got[mapped, :n], exp.T[mapped]) and not got[mapped, n:].any()
    unmapped = sorted(set(range(nc + 5)) - set(mapped))
    assert (got[unmapped] == np.uint64(0xFFFFFFFFFFFFFFFF)).all()            # columns the chip does not own stay untouched
    assert np.array_equal(lkd.cpu().numpy().view(np.uint32), elkd) and np.array_equal(lkf.cpu().numpy().view(np.uint32), elkf)


def test_lw_sw_bad_arguments_fail_loudly(dev):
    from ceno_amd import CenoHipError, api

    with pytest.raises(CenoHipError):
        api.witgen_mem(dev, list(range(23)) + [10], False, 8, 1, 8, 1, 8, 2)      # a column id beyond num_cols
    dup = list(range(23)) + [23]
    dup[21] = dup[0]
    with pytest.raises(CenoHipError):
        api.witgen_mem(dev, dup, True, 8, 1, 8, 1, 8, 2)                          # two fields on one column
    with pytest.raises(CenoHipError):
        api.witgen_mem(dev, list(range(23)) + [23], True, 0, 1, 8, 1, 8, 2)       # no step records


@pytest.mark.parametrize("n,rows,offset", [(1024, 1024, 0), (1, 2, 0), (300, 512, 0), (300, 512, 996)])
def test_jalr_witness_and_lookups_match_cpu_assignment(dev, n, rows, offset):
    """JALR: the jump target as a MemAddr with both low bits witnessed, rd = pc + 4 with its high limb, a branching state (pc, next_pc)"""
    from tests import hipbuf as torch

    from ceno_amd import api
    from tests.test_oracle_witgen import _jalr_steps

    d = _jalr_steps(n)
    if offset:
        d["cycles"] = d["cycles"] + np.uint64(1000)
        d["prev_cycles"][::3] = 500
        d["prev_cycles"][1::3] = 1
    recs = po.step_records_jalr(d["cycles"], d["pcs"], d["pcs_after"], 2, 4, d["imms"], d["rs1_vals"], d["rd_before"], d["rd_after"], d["prev_cycles"])
    nc = 22
    rng = np.random.default_rng(27)
    cols = list(rng.permutation(nc + 3)[:nc]) + [nc + 3]
    idx = rng.permutation(n) if offset else np.arange(n)
    d_recs = _to_dev(recs.reshape(-1))
    d_idx = _to_dev(idx.astype(np.uint32).view(np.int32))
    w = torch.full(((nc + 3) * rows,), -1, dtype=torch.int64, device="cuda:0")
    lkd = torch.zeros(1 << 19, dtype=torch.int32, device="cuda:0")
    lkf = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    api.witgen_jalr(dev, cols, d_recs.data_ptr(), n, d_idx.data_ptr(), n, w.data_ptr(), rows, offset, 0x2000, n, lkd.data_ptr(), lkf.data_ptr())
    dev.sync()
    got = w.cpu().numpy().view(np.uint64).reshape(nc + 3, rows)
    exp, elkd, elkf = po.witgen_jalr(cols, recs, idx, offset, 0x2000, n)
    mapped = sorted(cols[:nc])
    assert np.array_equal(got[mapped, :n], exp.T[mapped]) and not got[mapped, n:].any()
    assert np.array_equal(lkd.cpu().numpy().view(np.uint32), elkd) and np.array_equal(lkf.cpu().numpy().view(np.uint32), elkf)


@pytest.mark.parametrize("is_imm", [False, True])
@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("n,rows,offset", [(1024, 1024, 0), (1, 2, 0), (300, 512, 996)])
def test_shift_witness_and_lookups_match_cpu_assignment(dev, kind, is_imm, n, rows, offset):
    """SLL / SRL / SRA and their immediate forms: byte limbs, the ShiftBase gadget's markers, multiplier, carries and sign, four lookup tables"""
    from tests import hipbuf as torch

    from ceno_amd import api
    from tests.test_oracle_witgen import _shift_records, _shift_steps

    d = _shift_steps(n, kind, is_imm)
    if offset:
        d["cycles"] = d["cycles"] + np.uint64(1000)
        d["prev_cycles"][::3] = 500
        d["prev_cycles"][1::3] = 1
    recs = _shift_records(d, kind, is_imm)
    nc = 40 if is_imm else 47
    rng = np.random.default_rng(31 + kind)
    cols = list(rng.permutation(nc + 3)[:nc]) + [nc + 3]
    idx = rng.permutation(n) if offset else np.arange(n)
    d_recs = _to_dev(recs.reshape(-1))
    d_idx = _to_dev(idx.astype(np.uint32).view(np.int32))
    w = torch.full(((nc + 3) * rows,), -1, dtype=torch.int64, device="cuda:0")
    lkd = torch.zeros(1 << 19, dtype=torch.int32, device="cuda:0")
    lkf = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    lk2 = torch.zeros(1 << 16, dtype=torch.int32, device="cuda:0")
    lkx = torch.zeros(1 << 16, dtype=torch.int32, device="cuda:0")
    api.witgen_shift(dev, cols, is_imm, kind, d_recs.data_ptr(), n, d_idx.data_ptr(), n, w.data_ptr(), rows, offset, 0x1000, n, lkd.data_ptr(),
                     lkf.data_ptr(), lk2.data_ptr(), lkx.data_ptr())
    dev.sync()
    got = w.cpu().numpy().view(np.uint64).reshape(nc + 3, rows)
    exp, elkd, elkf, elk2, elkx = po.witgen_shift(cols, is_imm, kind, recs, idx, offset, 0x1000, n)
    mapped = sorted(cols[:nc])
    assert np.array_equal(got[mapped, :n], exp.T[mapped]) and not got[mapped, n:].any()
    for g_, e_ in ((lkd, elkd), (lkf, elkf), (lk2, elk2), (lkx, elkx)):
        assert np.array_equal(g_.cpu().numpy().view(np.uint32), e_)


def test_shift_and_jalr_bad_arguments_fail_loudly(dev):
    from ceno_amd import CenoHipError, api

    with pytest.raises(CenoHipError):
        api.witgen_shift(dev, list(range(47)) + [47], False, 3, 8, 1, 8, 1, 8, 2)      # kind out of range
    with pytest.raises(CenoHipError):
        api.witgen_shift(dev, list(range(40)) + [39], True, 0, 8, 1, 8, 1, 8, 2)       # a column id beyond num_cols
    with pytest.raises(CenoHipError):
        api.witgen_jalr(dev, list(range(22)) + [22], 0, 1, 8, 1, 8, 2)                 # no step records


@pytest.mark.parametrize("kind", [2, 3])
@pytest.mark.parametrize("n,rows,offset", [(1024, 1024, 0), (1, 2, 0), (300, 512, 996)])
def test_sh_sb_witness_and_lookups_match_cpu_assignment(dev, kind, n, rows, offset):
    """SH / SB: SW's columns, the free address bits, and SB's byte columns of the addressed limb with their byte-range lookups"""
    from tests import hipbuf as torch

    from ceno_amd import api
    from tests.test_oracle_witgen import _sub_store_records, _sub_store_steps

    d = _sub_store_steps(n, kind)
    if offset:
        d["cycles"] = d["cycles"] + np.uint64(1000)
        d["prev_cycles"][::3] = 500
        d["prev_cycles"][1::3] = 1
    recs = _sub_store_records(d, kind)
    nc = 24 if kind == 2 else 29
    rng = np.random.default_rng(40 + kind)
    cols = list(rng.permutation(nc + 4)[:nc]) + [nc + 4]
    idx = rng.permutation(n) if offset else np.arange(n)
    d_recs = _to_dev(recs.reshape(-1))
    d_idx = _to_dev(idx.astype(np.uint32).view(np.int32))
    w = torch.full(((nc + 4) * rows,), -1, dtype=torch.int64, device="cuda:0")
    lkd = torch.zeros(1 << 19, dtype=torch.int32, device="cuda:0")
    lkf = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    api.witgen_mem(dev, cols, kind, d_recs.data_ptr(), n, d_idx.data_ptr(), n, w.data_ptr(), rows, offset, 0x1000, n, lkd.data_ptr(), lkf.data_ptr())
    dev.sync()
    got = w.cpu().numpy().view(np.uint64).reshape(nc + 4, rows)
    exp, elkd, elkf = po.witgen_mem(cols, kind, recs, idx, offset, 0x1000, n)
    mapped = sorted(cols[:nc])
    assert np.array_equal(got[mapped, :n], exp.T[mapped]) and not got[mapped, n:].any()
    assert np.array_equal(lkd.cpu().numpy().view(np.uint32), elkd) and np.array_equal(lkf.cpu().numpy().view(np.uint32), elkf)


@pytest.mark.parametrize("signed", [False, True])
@pytest.mark.parametrize("width", [16, 8])
@pytest.mark.parametrize("n,rows,offset", [(1024, 1024, 0), (1, 2, 0), (300, 512, 996)])
def test_load_sub_witness_and_lookups_match_cpu_assignment(dev, width, signed, n, rows, offset):
    """LH / LHU / LB / LBU: LW's columns, limb and byte selection, the sign bit with its range lookup; absent Option columns are marked, not written"""
    from tests import hipbuf as torch

    from ceno_amd import CenoHipError, api
    from tests.test_oracle_witgen import _sub_load_records, _sub_load_steps

    d = _sub_load_steps(n, width, signed)
    if offset:
        d["cycles"] = d["cycles"] + np.uint64(1000)
        d["prev_cycles"][::3] = 500
        d["prev_cycles"][1::3] = 1
    recs = _sub_load_records(d, width, signed)
    nc = 25 + (3 if width == 8 else 0) + int(signed)
    rng = np.random.default_rng(50 + width + signed)
    ids = [int(x) for x in rng.permutation(nc + 3)[:nc]]
    cols = po.load_sub_cols(ids, width, signed, nc + 3)
    idx = rng.permutation(n) if offset else np.arange(n)
    d_recs = _to_dev(recs.reshape(-1))
    d_idx = _to_dev(idx.astype(np.uint32).view(np.int32))
    w = torch.full(((nc + 3) * rows,), -1, dtype=torch.int64, device="cuda:0")
    lkd = torch.zeros(1 << 19, dtype=torch.int32, device="cuda:0")
    lkf = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    api.witgen_load_sub(dev, cols, width, signed, d_recs.data_ptr(), n, d_idx.data_ptr(), n, w.data_ptr(), rows, offset, 0x1000, n, lkd.data_ptr(),
                        lkf.data_ptr())
    dev.sync()
    got = w.cpu().numpy().view(np.uint64).reshape(nc + 3, rows)
    exp, elkd, elkf = po.witgen_load_sub(cols, width, signed, recs, idx, offset, 0x1000, n)
    mapped = sorted(ids)
    assert np.array_equal(got[mapped, :n], exp.T[mapped]) and not got[mapped, n:].any()
    unmapped = sorted(set(range(nc + 3)) - set(mapped))
    assert (got[unmapped] == np.uint64(0xFFFFFFFFFFFFFFFF)).all()
    assert np.array_equal(lkd.cpu().numpy().view(np.uint32), elkd) and np.array_equal(lkf.cpu().numpy().view(np.uint32), elkf)
    if n == 1 and not signed:
        bad = list(cols)
        bad[28] = 0                                                                   # an unsigned load that names an msb column
        with pytest.raises(CenoHipError):
            api.witgen_load_sub(dev, bad, width, signed, d_recs.data_ptr(), n, d_idx.data_ptr(), n, w.data_ptr(), rows)


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
@pytest.mark.parametrize("n,rows,offset", [(1024, 1024, 0), (1, 2, 0), (300, 512, 996)])
def test_mul_witness_and_lookups_match_cpu_assignment(dev, kind, n, rows, offset):
    """MUL / MULH / MULHU / MULHSU: the schoolbook product over 16-bit limbs, 18-bit carry lookups (the upper part of the 2^19-entry dynamic table)"""
    from tests import hipbuf as torch

    from ceno_amd import CenoHipError, api
    from tests.test_oracle_witgen import _mul_cols, _mul_steps

    d = _mul_steps(n, kind)
    if offset:
        d["cycles"] = d["cycles"] + np.uint64(1000)
        d["prev_cycles"][::3] = 500
        d["prev_cycles"][1::3] = 1
    recs = po.step_records_r(d["cycles"], d["pcs"], [po.INSN_MUL, po.INSN_MULH, po.INSN_MULHU, po.INSN_MULHSU][kind], 2, 3, 4, d["rs1_vals"], d["rs2_vals"],
                             d["rd_before"], d["rd_after"], d["prev_cycles"])
    nc = 26 if kind else 22
    rng = np.random.default_rng(60 + kind)
    ids = [int(x) for x in rng.permutation(nc + 3)[:nc]]
    cols = _mul_cols(ids, kind, nc + 3)
    idx = rng.permutation(n) if offset else np.arange(n)
    d_recs = _to_dev(recs.reshape(-1))
    d_idx = _to_dev(idx.astype(np.uint32).view(np.int32))
    w = torch.full(((nc + 3) * rows,), -1, dtype=torch.int64, device="cuda:0")
    lkd = torch.zeros(1 << 19, dtype=torch.int32, device="cuda:0")
    lkf = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    api.witgen_mul(dev, cols, kind, d_recs.data_ptr(), n, d_idx.data_ptr(), n, w.data_ptr(), rows, offset, 0x1000, n, lkd.data_ptr(), lkf.data_ptr())
    dev.sync()
    got = w.cpu().numpy().view(np.uint64).reshape(nc + 3, rows)
    exp, elkd, elkf = po.witgen_mul(cols, kind, recs, idx, offset, 0x1000, n)
    mapped = sorted(ids)
    assert np.array_equal(got[mapped, :n], exp.T[mapped]) and not got[mapped, n:].any()
    assert np.array_equal(lkd.cpu().numpy().view(np.uint32), elkd) and np.array_equal(lkf.cpu().numpy().view(np.uint32), elkf)
    if n == 1 and kind == 0:
        with pytest.raises(CenoHipError):
            api.witgen_mul(dev, list(range(26)) + [26], 0, d_recs.data_ptr(), n, d_idx.data_ptr(), n, w.data_ptr(), rows)   # MUL naming rd_high columns


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
@pytest.mark.parametrize("n,rows,offset", [(1024, 1024, 0), (1, 2, 0), (300, 512, 996)])
def test_div_witness_and_lookups_match_cpu_assignment(dev, kind, n, rows, offset):
    """DIV / DIVU / REM / REMU: RISC-V's special cases, sign and zero flags with their field inverses, the 18-bit carries of divisor * quotient + remainder,
    the |remainder| < |divisor| comparison"""
    from tests import hipbuf as torch

    from ceno_amd import api
    from tests.test_oracle_witgen import _div_records, _div_steps

    d = _div_steps(n, kind)
    if offset:
        d["cycles"] = d["cycles"] + np.uint64(1000)
        d["prev_cycles"][::3] = 500
        d["prev_cycles"][1::3] = 1
    recs = _div_records(d, kind)
    nc = 39
    rng = np.random.default_rng(70 + kind)
    cols = [int(x) for x in rng.permutation(nc + 3)[:nc]] + [nc + 3]
    idx = rng.permutation(n) if offset else np.arange(n)
    d_recs = _to_dev(recs.reshape(-1))
    d_idx = _to_dev(idx.astype(np.uint32).view(np.int32))
    w = torch.full(((nc + 3) * rows,), -1, dtype=torch.int64, device="cuda:0")
    lkd = torch.zeros(1 << 19, dtype=torch.int32, device="cuda:0")
    lkf = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    api.witgen_div(dev, cols, kind, d_recs.data_ptr(), n, d_idx.data_ptr(), n, w.data_ptr(), rows, offset, 0x1000, n, lkd.data_ptr(), lkf.data_ptr())
    dev.sync()
    got = w.cpu().numpy().view(np.uint64).reshape(nc + 3, rows)
    exp, elkd, elkf = po.witgen_div(cols, kind, recs, idx, offset, 0x1000, n)
    mapped = sorted(cols[:nc])
    assert np.array_equal(got[mapped, :n], exp.T[mapped]) and not got[mapped, n:].any()
    assert np.array_equal(lkd.cpu().numpy().view(np.uint32), elkd) and np.array_equal(lkf.cpu().numpy().view(np.uint32), elkf)
