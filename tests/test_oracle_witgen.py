"""CPU: the oracle's ADD / SUB witness assignment (oracle/witgen.c) against an independent pure-Python model on the
reference test's step data, plus the circuit identities the columns must satisfy.  PARITY UNPINNED beyond the cited
reference code: the reference holds no literal vectors for this path (it compares GPU with CPU at run time)."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests import witgen_cases as wc


def _records(d, kind):
    return po.step_records_r(d["cycles"], d["pcs"], kind, 2, 3, 4, d["rs1_vals"], d["rs2_vals"], d["rd_before"], d["rd_after"], d["prev_cycles"])


def test_step_record_layout_is_the_emulators_repr_c():
    assert po.lib().orc_step_record_bytes() == 136
    d = wc.reference_test_steps(3)
    r = _records(d, po.INSN_ADD)
    dt = np.dtype([("cycle", "<u8"), ("pc", "<u4", 2), ("heap", "<u4", 2), ("hint", "<u4", 2), ("insn", "u1", 4), ("imm", "<i4"), ("raw", "<u4"),
                   ("has", "u1", 4), ("rs1", [("addr", "<u4"), ("value", "<u4"), ("prev", "<u8")]),
                   ("rs2", [("addr", "<u4"), ("value", "<u4"), ("prev", "<u8")]),
                   ("rd", [("addr", "<u4"), ("before", "<u4"), ("after", "<u4"), ("pad", "<u4"), ("prev", "<u8")]),
                   ("mem", [("addr", "<u4"), ("before", "<u4"), ("after", "<u4"), ("pad", "<u4"), ("prev", "<u8")]),
                   ("syscall", "<u4"), ("mask", "u1"), ("pad", "u1", 3)])
    assert dt.itemsize == 136
    v = r.view(dt).reshape(-1)
    assert list(v["cycle"]) == [4, 8, 12] and list(v["pc"][:, 0]) == [0x1000, 0x1004, 0x1008] and list(v["pc"][:, 1]) == [0x1004, 0x1008, 0x100C]
    assert list(v["insn"][0]) == [po.INSN_ADD, 2, 3, 4] and list(v["has"][0]) == [1, 1, 1, 0]
    assert v["rs1"]["addr"][0] == (2 << 8) // 4 and v["rd"]["addr"][0] == (4 << 8) // 4 and v["syscall"][0] == 0xFFFFFFFF
    assert v["rs2"]["value"][1] == 1 and v["rd"]["after"][1] == 1


@pytest.mark.parametrize("sub", [False, True])
@pytest.mark.parametrize("offset,prev", [(0, 0), (0, 5), (400, 100), (400, 402), (400, 900)])
def test_oracle_matches_python_model(sub, offset, prev):
    n = 300
    d = wc.reference_test_steps(n, sub)
    d["cycles"] = d["cycles"] + offset + 1000
    d["prev_cycles"][:] = prev
    d["prev_cycles"][::7] = 0
    recs = _records(d, po.INSN_SUB if sub else po.INSN_ADD)
    rng = np.random.default_rng(5)
    cols = list(rng.permutation(30)[:22]) + [30]
    idx = rng.permutation(n)[:250]
    base_pc, slots = 0x1000, n
    got, lkd, lkf = po.witgen_arith(cols, sub, recs, idx, offset, base_pc, slots)
    exp_dyn, exp_fetch = np.zeros(1 << 19, dtype=np.uint32), np.zeros(slots, dtype=np.uint32)
    for r, i in enumerate(idx):
        row, lk = wc.model_row(cols, sub, int(d["cycles"][i]), int(d["pcs"][i]), 2, 3, 4, int(d["rs1_vals"][i]), int(d["rs2_vals"][i]),
                               int(d["rd_before"][i]), int(d["rd_after"][i]), int(d["prev_cycles"][i]), offset)
        for c, v in row.items():
            assert int(got[r, c]) == v, (r, c)
        unmapped = [c for c in range(30) if c not in row]
        assert not got[r, unmapped].any()
        for t, k in lk:
            if t == "dyn":
                exp_dyn[k] += 1
            else:
                exp_fetch[(k - base_pc) // 4] += 1
    assert np.array_equal(lkd, exp_dyn) and np.array_equal(lkf, exp_fetch)
    assert int(lkd.sum()) == len(idx) * (10 if sub else 8) and int(lkf.sum()) == len(idx)


def test_columns_satisfy_the_chip_constraints():
    """rs1 + rs2 = rd with the witnessed carries (UIntLimbs::add), and prev_ts - ts = diff - 2^29 (InnerLtConfig)"""
    n = 1024
    d = wc.reference_test_steps(n)
    recs = _records(d, po.INSN_ADD)
    m, _, _ = po.witgen_arith(wc.NATURAL_COLS, False, recs, np.arange(n), 0, 0x1000, n)
    m = m.astype(np.int64)
    rd = d["rd_after"].astype(np.int64)
    assert np.array_equal(m[:, 16] + m[:, 18], (rd & 0xFFFF) + (m[:, 20] << 16))
    assert np.array_equal(m[:, 17] + m[:, 19] + m[:, 20], (rd >> 16) + (m[:, 21] << 16))
    for base, diff0, sub_cycle in ((3, 4, 0), (7, 8, 1), (11, 14, 2)):
        assert np.array_equal(m[:, base] - (m[:, 1] + sub_cycle), m[:, diff0] + (m[:, diff0 + 1] << 16) - (1 << 29))
    assert np.all(m[:, 2] == 2) and np.all(m[:, 6] == 3) and np.all(m[:, 10] == 4)


def test_bad_column_map_is_rejected():
    d = wc.reference_test_steps(4)
    recs = _records(d, po.INSN_ADD)
    cols = list(range(22)) + [21]
    with pytest.raises(ValueError):
        po.witgen_arith(cols, False, recs, np.arange(4))


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("offset,prev", [(0, 0), (400, 402), (400, 900)])
def test_logic_oracle_matches_python_model(kind, offset, prev):
    """AND / OR / XOR (logic_circuit.rs:66-160): 28 columns, four byte-pair lookups into the operation's table per instance"""
    n = 300
    d = wc.reference_logic_steps(n, kind)
    d["cycles"] = d["cycles"] + offset + 1000
    d["prev_cycles"][:] = prev
    d["prev_cycles"][::7] = 0
    recs = _records(d, (po.INSN_AND, po.INSN_OR, po.INSN_XOR)[kind])
    rng = np.random.default_rng(6)
    cols = list(rng.permutation(34)[:28]) + [34]
    idx = rng.permutation(n)[:250]
    base_pc, slots = 0x1000, n
    got, lkd, lkf, lkl = po.witgen_logic_r(cols, recs, idx, offset, base_pc, slots)
    exp_dyn, exp_fetch, exp_logic = np.zeros(1 << 19, dtype=np.uint32), np.zeros(slots, dtype=np.uint32), np.zeros(1 << 16, dtype=np.uint32)
    for r, i in enumerate(idx):
        row, lk = wc.model_logic_row(cols, int(d["cycles"][i]), int(d["pcs"][i]), 2, 3, 4, int(d["rs1_vals"][i]), int(d["rs2_vals"][i]),
                                     int(d["rd_before"][i]), int(d["rd_after"][i]), int(d["prev_cycles"][i]), offset)
        assert len(row) == 28
        for c, v in row.items():
            assert int(got[r, c]) == v, (r, c)
        assert not got[r, [c for c in range(34) if c not in row]].any()
        for t, k in lk:
            if t == "dyn":
                exp_dyn[k] += 1
            elif t == "logic":
                exp_logic[k] += 1
            else:
                exp_fetch[(k - base_pc) // 4] += 1
    assert np.array_equal(lkd, exp_dyn) and np.array_equal(lkf, exp_fetch) and np.array_equal(lkl, exp_logic)
    assert int(lkd.sum()) == 6 * len(idx) and int(lkl.sum()) == 4 * len(idx)


def test_logic_columns_satisfy_the_chip_constraints():
    """the byte columns recompose the registers and (rs1 byte, rs2 byte, rd byte) is a row of the operation's table"""
    n = 1024
    for kind in (0, 1, 2):
        d = wc.reference_logic_steps(n, kind)
        recs = _records(d, (po.INSN_AND, po.INSN_OR, po.INSN_XOR)[kind])
        m, _, _, lkl = po.witgen_logic_r(wc.LOGIC_NATURAL_COLS, recs, np.arange(n), 0, 0x1000, n)
        m = m.astype(np.int64)
        for base, reg in ((16, d["rs1_vals"]), (20, d["rs2_vals"]), (24, d["rd_after"])):
            assert np.array_equal(sum(m[:, base + b] << (8 * b) for b in range(4)), reg.astype(np.int64))
        for b in range(4):
            assert np.array_equal(wc.LOGIC_OPS[kind](m[:, 16 + b], m[:, 20 + b]), m[:, 24 + b])
        keys = np.concatenate([m[:, 16 + b] | (m[:, 20 + b] << 8) for b in range(4)])
        assert np.array_equal(np.bincount(keys, minlength=1 << 16).astype(np.uint32), lkl)


@pytest.mark.parametrize("offset,prev", [(0, 0), (400, 402), (400, 900)])
def test_addi_oracle_matches_python_model(offset, prev):
    """ADDI (arith_imm_circuit_v2.rs:85-117): 18 columns; negative immediates, carries out of both limbs"""
    n = 2100
    d = wc.reference_addi_steps(n)
    d["rs1_vals"][:6] = [0, 0xFFFFFFFF, 0xFFFF, 0x10000, 0x7FFFFFFF, 0x80000000]
    d["imms"][:6] = [-1, 1, 1, -1, 2047, -2048]
    d["rd_after"] = ((d["rs1_vals"].astype(np.int64) + d["imms"]) & 0xFFFFFFFF).astype(np.uint64)
    d["cycles"] = d["cycles"] + offset + 1000
    d["prev_cycles"][:] = prev
    d["prev_cycles"][::7] = 0
    recs = po.step_records_i(d["cycles"], d["pcs"], po.INSN_ADDI, 2, 4, d["imms"], d["rs1_vals"], d["rd_before"], d["rd_after"], d["prev_cycles"])
    rng = np.random.default_rng(8)
    cols = list(rng.permutation(25)[:18]) + [25]
    idx = np.concatenate([np.arange(6), 6 + rng.permutation(n - 6)[:400]])
    base_pc, slots = 0x1000, n
    got, lkd, lkf = po.witgen_addi(cols, recs, idx, offset, base_pc, slots)
    exp_dyn, exp_fetch = np.zeros(1 << 19, dtype=np.uint32), np.zeros(slots, dtype=np.uint32)
    for r, i in enumerate(idx):
        row, lk = wc.model_addi_row(cols, int(d["cycles"][i]), int(d["pcs"][i]), 2, 4, int(d["rs1_vals"][i]), int(d["imms"][i]), int(d["rd_before"][i]),
                                    int(d["prev_cycles"][i]), offset)
        assert len(row) == 18
        for c, v in row.items():
            assert int(got[r, c]) == v, (r, c)
        assert not got[r, [c for c in range(25) if c not in row]].any()
        for t, k in lk:
            if t == "dyn":
                exp_dyn[k] += 1
            else:
                exp_fetch[(k - base_pc) // 4] += 1
    assert np.array_equal(lkd, exp_dyn) and np.array_equal(lkf, exp_fetch)
    # the witnessed carries prove rs1 + sign_extend(imm) = rd
    m = got.astype(np.int64)
    c = {k: cols[k] for k in range(18)}
    rd = d["rd_after"][idx].astype(np.int64)
    ext_hi = m[:, c[15]] * 0xFFFF
    assert np.array_equal(m[:, c[12]] + m[:, c[14]], (rd & 0xFFFF) + (m[:, c[16]] << 16))
    assert np.array_equal(m[:, c[13]] + ext_hi + m[:, c[16]], (rd >> 16) + (m[:, c[17]] << 16))


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("offset,prev", [(0, 0), (400, 900)])
def test_logic_i_oracle_matches_python_model(kind, offset, prev):
    """ANDI / ORI / XORI (logic_imm_circuit_v2.rs:105-130,195-224): 24 columns; raw 12-bit immediates as in the reference's test and
    sign-extended negative ones as the decoder produces them"""
    n = 600
    d = wc.reference_logic_i_steps(n, kind)
    d["imms"][300:] = -(d["imms"][300:] % 2048) - 1      # negative immediates: high half 0xffff
    d["rd_after"] = wc.LOGIC_OPS[kind](d["rs1_vals"], d["imms"].astype(np.uint64) & np.uint64(0xFFFFFFFF))
    d["cycles"] = d["cycles"] + offset + 1000
    d["prev_cycles"][:] = prev
    recs = po.step_records_i(d["cycles"], d["pcs"], (po.INSN_ANDI, po.INSN_ORI, po.INSN_XORI)[kind], 2, 4, d["imms"], d["rs1_vals"], d["rd_before"],
                             d["rd_after"], d["prev_cycles"])
    rng = np.random.default_rng(10)
    cols = list(rng.permutation(31)[:24]) + [31]
    idx = rng.permutation(n)[:500]
    base_pc, slots = 0x1000, n
    got, lkd, lkf, lkl = po.witgen_logic_i(cols, recs, idx, offset, base_pc, slots)
    exp_dyn, exp_fetch, exp_logic = np.zeros(1 << 19, dtype=np.uint32), np.zeros(slots, dtype=np.uint32), np.zeros(1 << 16, dtype=np.uint32)
    for r, i in enumerate(idx):
        row, lk = wc.model_logic_i_row(cols, int(d["cycles"][i]), int(d["pcs"][i]), 2, 4, int(d["rs1_vals"][i]), int(d["imms"][i]), int(d["rd_before"][i]),
                                       int(d["rd_after"][i]), int(d["prev_cycles"][i]), offset)
        assert len(row) == 24
        for c, v in row.items():
            assert int(got[r, c]) == v, (r, c)
        for t, k in lk:
            if t == "dyn":
                exp_dyn[k] += 1
            elif t == "logic":
                exp_logic[k] += 1
            else:
                exp_fetch[(k - base_pc) // 4] += 1
    assert np.array_equal(lkd, exp_dyn) and np.array_equal(lkf, exp_fetch) and np.array_equal(lkl, exp_logic)
    # (rs1 byte, imm byte, rd byte) is a row of the operation's table
    m = got.astype(np.int64)
    for b in range(4):
        assert np.array_equal(wc.LOGIC_OPS[kind](m[:, cols[12 + b]], m[:, cols[20 + b]]), m[:, cols[16 + b]])


def _lui_steps(n):
    """chips/lui.rs:79-97: imm = (i % 2^20) << 12, rd = imm, rs1 = x0 reading 0"""
    i = np.arange(n, dtype=np.int64)
    imm = ((i * 4099 + 7) % (1 << 20)) << 12          # spread over the 20-bit range, bit 31 set for half of them
    edge = [0, 1 << 12, 0xFFFFF << 12, 0x80000 << 12]
    imm[:min(n, 4)] = edge[:min(n, 4)]
    imm_i32 = ((imm + (1 << 31)) % (1 << 32)) - (1 << 31)
    return dict(cycles=(4 + 4 * i).astype(np.uint64), pcs=(0x1000 + 4 * i).astype(np.uint64), imms=imm_i32, rs1_vals=np.zeros(n, dtype=np.uint64),
                rd_before=(i % 200).astype(np.uint64), rd_after=imm.astype(np.uint64), prev_cycles=np.zeros(n, dtype=np.uint64))


def test_lui_oracle_rows_and_lookups():
    """LUI (riscv/lui.rs:100-120): rd bytes 1..3 recompose imm << 4 (the circuit's own constraint), each a byte lookup of the dynamic table"""
    n = 500
    d = _lui_steps(n)
    recs = po.step_records_i(d["cycles"], d["pcs"], po.INSN_LUI, 0, 4, d["imms"], d["rs1_vals"], d["rd_before"], d["rd_after"], d["prev_cycles"])
    rng = np.random.default_rng(14)
    cols = list(rng.permutation(20)[:16]) + [20]
    idx = np.arange(n)
    got, lkd, lkf = po.witgen_lui(cols, recs, idx, 0, 0x1000, n)
    m = got.astype(np.int64)
    imm20 = (d["rd_after"] >> np.uint64(12)).astype(np.int64)
    assert np.array_equal(m[:, cols[15]], imm20)
    assert np.array_equal(m[:, cols[12]] + (m[:, cols[13]] << 8) + (m[:, cols[14]] << 16), imm20 << 4)   # require_equal in construct_circuit
    assert np.all(m[:, cols[2]] == 0) and np.all(m[:, cols[6]] == 4) and np.array_equal(m[:, cols[0]], d["pcs"].astype(np.int64))
    for base, diff0, sub_cycle in ((3, 4, 0), (7, 10, 2)):
        assert np.array_equal(m[:, cols[base]] - (m[:, cols[1]] + sub_cycle), m[:, cols[diff0]] + (m[:, cols[diff0 + 1]] << 16) - (1 << 29))
    exp = np.zeros(1 << 19, dtype=np.int64)
    for b in (12, 13, 14):
        np.add.at(exp, (1 << 8) + m[:, cols[b]], 1)
    for d0 in (4, 10):
        np.add.at(exp, (1 << 16) + m[:, cols[d0]], 1)
        np.add.at(exp, (1 << 13) + m[:, cols[d0 + 1]], 1)
    assert np.array_equal(lkd.astype(np.int64), exp) and np.all(lkf == 1)


def _jal_steps(n):
    """chips/jal.rs:66-86: pc = 0x1000 + 4 i, a jump offset, rd = pc + 4"""
    i = np.arange(n, dtype=np.int64)
    pc = 0x1000 + 4 * i
    pc[: min(n, 3)] = [0x3FFFFFF8, 0x00FFFFFC, 0x1000][: min(n, 3)]       # top byte 0x3f (the XOR check's edge), a carry into byte 3
    off = ((i * 52) % 4096) - 2048
    return dict(cycles=(4 + 4 * i).astype(np.uint64), pcs=pc.astype(np.uint64), pcs_after=((pc + off) & 0xFFFFFFFF).astype(np.uint64), imms=off,
                rd_before=(i % 200).astype(np.uint64), rd_after=((pc + 4) & 0xFFFFFFFF).astype(np.uint64), prev_cycles=np.zeros(n, dtype=np.uint64))


def _auipc_steps(n):
    """chips/auipc.rs:81-99: imm = (i % 2^20) << 12, rd = pc + imm"""
    i = np.arange(n, dtype=np.int64)
    pc = 0x1000 + 4 * i
    pc[: min(n, 2)] = [0x3FABCDE0, 0x00010000][: min(n, 2)]
    imm = ((i * 4099 + 7) % (1 << 20)) << 12
    imm_i32 = ((imm + (1 << 31)) % (1 << 32)) - (1 << 31)
    return dict(cycles=(4 + 4 * i).astype(np.uint64), pcs=pc.astype(np.uint64), imms=imm_i32, rs1_vals=np.zeros(n, dtype=np.uint64),
                rd_before=(i % 200).astype(np.uint64), rd_after=((pc + imm) & 0xFFFFFFFF).astype(np.uint64), prev_cycles=np.zeros(n, dtype=np.uint64))


def test_jal_oracle_rows_and_lookups():
    n = 400
    d = _jal_steps(n)
    recs = po.step_records_j(d["cycles"], d["pcs"], d["pcs_after"], po.INSN_JAL, 4, d["imms"], d["rd_before"], d["rd_after"], d["prev_cycles"])
    rng = np.random.default_rng(16)
    cols = list(rng.permutation(17)[:13]) + [17]
    got, lkd, lkf, lk2, lkx = po.witgen_jal(cols, recs, np.arange(n), 0, 0x1000, 1 << 20)
    m = got.astype(np.int64)
    assert np.array_equal(m[:, cols[0]], d["pcs"].astype(np.int64)) and np.array_equal(m[:, cols[1]], d["pcs_after"].astype(np.int64))
    assert np.array_equal(sum(m[:, cols[9 + b]] << (8 * b) for b in range(4)), d["pcs"].astype(np.int64) + 4)       # rd = pc + 4
    assert np.array_equal(m[:, cols[4]] - (m[:, cols[2]] + 2), m[:, cols[7]] + (m[:, cols[8]] << 16) - (1 << 29))
    e2, ex = np.zeros(1 << 16, dtype=np.int64), np.zeros(1 << 16, dtype=np.int64)
    np.add.at(e2, (m[:, cols[9]] << 8) + m[:, cols[10]], 1)
    np.add.at(e2, (m[:, cols[11]] << 8) + m[:, cols[12]], 1)
    np.add.at(ex, m[:, cols[12]] | (0xC0 << 8), 1)
    assert np.array_equal(lk2.astype(np.int64), e2) and np.array_equal(lkx.astype(np.int64), ex)
    assert int(lkd.sum()) == 2 * n and int(lkf.sum()) == n - 2    # two program counters lie outside the fetch table's slots


def test_auipc_oracle_rows_and_lookups():
    n = 400
    d = _auipc_steps(n)
    recs = po.step_records_i(d["cycles"], d["pcs"], po.INSN_AUIPC, 0, 4, d["imms"], d["rs1_vals"], d["rd_before"], d["rd_after"], d["prev_cycles"])
    rng = np.random.default_rng(17)
    cols = list(rng.permutation(26)[:21]) + [26]
    got, lkd, lkf, lk2, lkx = po.witgen_auipc(cols, recs, np.arange(n), 0, 0x1000, 1 << 20)
    m = got.astype(np.int64)
    pc = d["pcs"].astype(np.int64)
    imm24 = (d["imms"].astype(np.int64) & 0xFFFFFFFF) >> 8
    assert np.array_equal(sum(m[:, cols[12 + b]] << (8 * b) for b in range(4)), d["rd_after"].astype(np.int64))
    assert np.array_equal(m[:, cols[16]], (pc >> 8) & 0xFF) and np.array_equal(m[:, cols[17]], (pc >> 16) & 0xFF)
    assert np.array_equal(sum(m[:, cols[18 + b]] << (8 * b) for b in range(3)), imm24)
    # the circuit's identity: rd = pc + (imm24 << 8) mod 2^32
    assert np.array_equal(d["rd_after"].astype(np.int64), (pc + (imm24 << 8)) & 0xFFFFFFFF)
    ed = np.zeros(1 << 19, dtype=np.int64)
    for c in (16, 17, 18, 19, 20):
        np.add.at(ed, (1 << 8) + m[:, cols[c]], 1)
    for d0 in (4, 10):
        np.add.at(ed, (1 << 16) + m[:, cols[d0]], 1)
        np.add.at(ed, (1 << 13) + m[:, cols[d0 + 1]], 1)
    ex = np.zeros(1 << 16, dtype=np.int64)
    np.add.at(ex, (pc >> 24) | (0xC0 << 8), 1)
    assert np.array_equal(lkd.astype(np.int64), ed) and np.array_equal(lkx.astype(np.int64), ex) and int(lk2.sum()) == 2 * n


P_GL = (1 << 64) - (1 << 32) + 1


def _slt_steps(n, signed):
    """chips/slt.rs:94-118: rs1 = 137 i - 500, rs2 = 89 i - 300 (mixing signs), plus edge cases"""
    i = np.arange(n, dtype=np.int64)
    a = (i * 137 - 500) & 0xFFFFFFFF
    b = (i * 89 - 300) & 0xFFFFFFFF
    edge = [(0, 0), (5, 5), (0x80000000, 0x7FFFFFFF), (0x7FFFFFFF, 0x80000000), (0xFFFFFFFF, 0), (0, 0xFFFFFFFF), (0x00010000, 0x0000FFFF),
            (0x8000FFFF, 0x80000000), (0xFFFF0000, 0xFFFF0001), (0x12340000, 0x12340000)]
    for k, (x, y) in enumerate(edge[:n]):
        a[k], b[k] = x, y
    if signed:
        sa, sb = np.where(a >> 31, a - (1 << 32), a), np.where(b >> 31, b - (1 << 32), b)
        lt = (sa < sb).astype(np.uint64)
    else:
        lt = (a < b).astype(np.uint64)
    return dict(cycles=(4 + 4 * i).astype(np.uint64), pcs=(0x1000 + 4 * i).astype(np.uint64), rs1_vals=a.astype(np.uint64), rs2_vals=b.astype(np.uint64),
                rd_before=(i % 200).astype(np.uint64), rd_after=lt, prev_cycles=np.zeros(n, dtype=np.uint64))


@pytest.mark.parametrize("signed", [True, False])
def test_slt_oracle_satisfies_the_comparison_gadget(signed):
    """SLT / SLTU (slt_circuit_v2.rs:86-119): the columns satisfy the UIntLimbsLT constraints as the circuit states them
    (signed_limbs.rs: the marker selects the most significant differing limb, diff_val is the positive difference there, cmp_lt = rd)"""
    n = 700
    d = _slt_steps(n, signed)
    recs = _records(d, po.INSN_SLT if signed else po.INSN_SLTU)
    rng = np.random.default_rng(19)
    cols = list(rng.permutation(32)[:26]) + [32]
    got, lkd, lkf = po.witgen_slt(cols, signed, recs, np.arange(n), 0, 0x1000, n)
    g = [[int(v) for v in row] for row in got]
    exp = np.zeros(1 << 19, dtype=np.int64)
    for r in range(n):
        row = g[r]
        a = [row[cols[0]], row[cols[1]]]
        b = [row[cols[2]], row[cols[3]]]
        assert a[0] | (a[1] << 16) == int(d["rs1_vals"][r]) and b[0] | (b[1] << 16) == int(d["rs2_vals"][r])
        lt, am, bm, mk, dv = row[cols[4]], row[cols[5]], row[cols[6]], [row[cols[7]], row[cols[8]]], row[cols[9]]
        assert lt == int(d["rd_after"][r])                                       # rd_written = is_lt
        # the top limbs as field elements of the signed values
        sa = a[1] - 65536 if (signed and a[1] >> 15) else a[1]
        sb = b[1] - 65536 if (signed and b[1] >> 15) else b[1]
        assert am == sa % P_GL and bm == sb % P_GL
        # marker: one-hot on the most significant differing limb (or all zero when equal), diff positive in the direction cmp_lt says
        vals_a, vals_b = [a[0], sa], [b[0], sb]
        if a == b:
            assert mk == [0, 0] and dv == 0 and lt == 0
            exp[(1 << 16) + 0] += 1
        else:
            k = 1 if a[1] != b[1] else 0
            assert mk == [int(k == 0), int(k == 1)]
            diff = (vals_b[k] - vals_a[k]) if lt else (vals_a[k] - vals_b[k])
            assert 0 < diff < (1 << 16) and dv == diff
            exp[(1 << 16) + diff - 1] += 1
        for limb, neg in ((a[1], signed and a[1] >> 15), (b[1], signed and b[1] >> 15)):
            exp[(1 << 16) + (limb - 0x8000 if neg else limb + (0x8000 if signed else 0))] += 1
        for d0 in (14, 18, 24):
            exp[(1 << 16) + row[cols[d0]]] += 1
            exp[(1 << 13) + row[cols[d0 + 1]]] += 1
    assert np.array_equal(lkd.astype(np.int64), exp) and np.all(lkf == 1)


def _slti_steps(n, signed):
    i = np.arange(n, dtype=np.int64)
    a = (i * 137 - 500) & 0xFFFFFFFF
    imm = (i * 29) % 4096 - 2048
    edge = [(0, 0), (5, 5), (0xFFFFFFFF, -1), (0x7FFFFFFF, -1), (0x80000000, 2047), (0, -2048), (0xFFFFF800, -2048), (0x0000FFFF, -1), (0xFFFF0000, 0)]
    for k, (x, y) in enumerate(edge[:n]):
        a[k], imm[k] = x, y
    b = imm & 0xFFFFFFFF                                  # the sign-extended immediate as a 32-bit word
    if signed:
        lt = (np.where(a >> 31, a - (1 << 32), a) < imm).astype(np.uint64)
    else:
        lt = (a < b).astype(np.uint64)
    return dict(cycles=(4 + 4 * i).astype(np.uint64), pcs=(0x1000 + 4 * i).astype(np.uint64), rs1_vals=a.astype(np.uint64), imms=imm,
                rd_before=(i % 200).astype(np.uint64), rd_after=lt, prev_cycles=np.zeros(n, dtype=np.uint64))


@pytest.mark.parametrize("signed", [True, False])
def test_slti_oracle_agrees_with_slt_on_the_extended_immediate(signed):
    """SLTI / SLTIU: the comparison columns are those SLT / SLTU produces for (rs1, sign_extend(imm)) — the gadget is shared — and cmp_lt = rd"""
    n = 600
    d = _slti_steps(n, signed)
    recs_i = po.step_records_i(d["cycles"], d["pcs"], po.INSN_SLTI if signed else po.INSN_SLTIU, 2, 4, d["imms"], d["rs1_vals"], d["rd_before"], d["rd_after"],
                               d["prev_cycles"])
    cols_i = list(range(22)) + [22]
    got, lkd, lkf = po.witgen_slti(cols_i, signed, recs_i, np.arange(n), 0, 0x1000, n)
    ext = (d["imms"] & 0xFFFFFFFF).astype(np.uint64)
    recs_r = po.step_records_r(d["cycles"], d["pcs"], po.INSN_SLT if signed else po.INSN_SLTU, 2, 3, 4, d["rs1_vals"], ext, d["rd_before"], d["rd_after"],
                               d["prev_cycles"])
    ref, _, _ = po.witgen_slt(list(range(26)) + [26], signed, recs_r, np.arange(n), 0, 0x1000, n)
    assert np.array_equal(got[:, 4:10], ref[:, 4:10])                      # cmp_lt, a_msb_f, b_msb_f, diff_marker, diff_val
    assert np.array_equal(got[:, 0:2], ref[:, 0:2])                        # rs1 limbs
    assert np.array_equal(got[:, 4], d["rd_after"])
    assert np.array_equal(got[:, 2], (d["imms"] & 0xFFFF).astype(np.uint64)) and np.array_equal(got[:, 3], (d["imms"] < 0).astype(np.uint64))
    assert int(lkd.sum()) == 7 * n and np.all(lkf == 1)


def _branch_steps(n, kind):
    """chips/branch_cmp.rs:93-118 / branch_eq.rs: rs1 = 137 i - 500, rs2 = 89 i - 300, offset -8 when taken, +4 otherwise; plus edge cases"""
    d = _slt_steps(n, kind in ("BLT", "BGE"))
    a, b = d["rs1_vals"].astype(np.int64), d["rs2_vals"].astype(np.int64)
    if kind in ("BEQ", "BNE"):
        b[::5] = a[::5]                                  # equal operands
        b[1::7] = a[1::7] ^ 0x10000                      # differ in the high limb only
    sa, sb = np.where(a >> 31, a - (1 << 32), a), np.where(b >> 31, b - (1 << 32), b)
    taken = {"BEQ": a == b, "BNE": a != b, "BLT": sa < sb, "BGE": sa >= sb, "BLTU": a < b, "BGEU": a >= b}[kind]
    pc = d["pcs"].astype(np.int64) + 0x1000
    imm = np.full(n, -8, dtype=np.int64)
    imm[::3] = 2044
    return dict(cycles=d["cycles"], pcs=pc.astype(np.uint64), pcs_after=np.where(taken, pc + imm, pc + 4).astype(np.uint64), imms=imm,
                rs1_vals=a.astype(np.uint64), rs2_vals=b.astype(np.uint64), prev_cycles=np.zeros(n, dtype=np.uint64), taken=taken)


@pytest.mark.parametrize("kind", ["BLT", "BGE", "BLTU", "BGEU"])
def test_branch_cmp_oracle_agrees_with_slt_and_the_branch_decision(kind):
    n = 500
    signed = kind in ("BLT", "BGE")
    d = _branch_steps(n, kind)
    recs = po.step_records_b(d["cycles"], d["pcs"], d["pcs_after"], getattr(po, "INSN_" + kind), 2, 3, d["imms"], d["rs1_vals"], d["rs2_vals"], d["prev_cycles"])
    got, lkd, lkf = po.witgen_branch(list(range(22)) + [22], False, signed, recs, np.arange(n), 0, 0x2000, n)
    recs_r = po.step_records_r(d["cycles"], d["pcs"], po.INSN_SLT if signed else po.INSN_SLTU, 2, 3, 4, d["rs1_vals"], d["rs2_vals"], np.zeros(n), np.zeros(n),
                               d["prev_cycles"])
    ref, _, _ = po.witgen_slt(list(range(26)) + [26], signed, recs_r, np.arange(n), 0, 0x2000, n)
    assert np.array_equal(got[:, :10], ref[:, :10])                       # limbs + the shared comparison gadget
    lt = got[:, 4].astype(bool)
    assert np.array_equal(lt if kind in ("BLT", "BLTU") else ~lt, d["taken"])
    m = got.astype(np.int64)
    assert np.array_equal(m[:, 10], d["pcs"].astype(np.int64)) and np.array_equal(m[:, 11], d["pcs_after"].astype(np.int64))
    imm_f = np.where(d["imms"] < 0, P_GL + d["imms"].astype(object), d["imms"].astype(object))
    assert [int(x) for x in got[:, 21]] == [int(x) for x in imm_f]
    assert int(lkd.sum()) == 7 * n and int(lkf.sum()) == n


@pytest.mark.parametrize("kind", ["BEQ", "BNE"])
def test_branch_eq_oracle_inverse_marker(kind):
    n = 400
    d = _branch_steps(n, kind)
    recs = po.step_records_b(d["cycles"], d["pcs"], d["pcs_after"], getattr(po, "INSN_" + kind), 2, 3, d["imms"], d["rs1_vals"], d["rs2_vals"], d["prev_cycles"])
    got, lkd, lkf = po.witgen_branch(list(range(19)) + [19], True, kind == "BEQ", recs, np.arange(n), 0, 0x2000, n)
    for r in range(n):
        row = [int(v) for v in got[r]]
        a, b = [row[0], row[1]], [row[2], row[3]]
        assert row[4] == int(d["taken"][r])
        # the circuit's equality test: sum_i marker_i (a_i - b_i) = 1 when the operands differ, every marker zero when they are equal
        acc = sum(row[5 + k] * ((a[k] - b[k]) % P_GL) for k in range(2)) % P_GL
        assert acc == (0 if a == b else 1)
        if a != b:
            k = 0 if a[0] != b[0] else 1
            assert row[5 + (1 - k)] == 0
    assert int(lkd.sum()) == 4 * n


def _mem_steps(n, is_store):
    """chips/lw.rs:69-101 / chips/sw.rs:88-117: base 0x1000 + 16 i, offsets 0, 4, -4, -8, memory value 111 i mod 10^6; plus edge cases (a high address,
    the largest and smallest offsets, a wrapped sum, a previous access in an earlier shard)"""
    i = np.arange(n, dtype=np.int64)
    rs1 = 0x1000 + 16 * i
    imm = np.array([0, 4, -4, -8], dtype=np.int64)[i % 4]
    mem_val = (i * 111) % 1000000
    prev_mem = (i * 77) % 500000
    mem_prev_cycle = np.zeros(n, dtype=np.uint64)
    if n >= 8:
        rs1[1], imm[1] = 0x3FFF_F000, 2044        # both address limbs at their 14-bit limit region
        rs1[2], imm[2] = 0x2000_0800, -2048       # the smallest offset
        rs1[3], imm[3] = 0xFFFF_FFFC, 8           # the sum wraps: address 4
        rs1[5], imm[5] = 0x3FFF_FFF8, 4           # the largest address MEM_BITS = 30 admits
        mem_val[6], prev_mem[6] = 0xFFFF_FFFF, 0xFFFF_0001
        mem_prev_cycle[7] = 8                     # touched two steps ago
    addr = (rs1 + imm) & 0xFFFFFFFF
    return dict(cycles=(4 + 4 * i).astype(np.uint64), pcs=(0x1000 + 4 * i).astype(np.uint64), imms=imm, rs1_vals=rs1.astype(np.uint64),
                rs2_vals=mem_val.astype(np.uint64), rd_before=(i % 200).astype(np.uint64), rd_after=mem_val.astype(np.uint64), mem_addrs=addr.astype(np.uint64),
                mem_before=(prev_mem if is_store else mem_val).astype(np.uint64), mem_after=mem_val.astype(np.uint64), prev_cycles=np.zeros(n, dtype=np.uint64),
                mem_prev_cycles=mem_prev_cycle)


def _mem_records(d, is_store):
    return po.step_records_mem(is_store, d["cycles"], d["pcs"], po.INSN_SW if is_store else po.INSN_LW, 2, 3 if is_store else 4, d["imms"], d["rs1_vals"],
                               d["rs2_vals"], d["rd_before"], d["rd_after"], d["mem_addrs"], d["mem_before"], d["mem_after"], d["prev_cycles"],
                               d["mem_prev_cycles"])


@pytest.mark.parametrize("is_store", [False, True])
def test_lw_sw_oracle_rows_satisfy_the_circuit_relations(is_store):
    """load_v2.rs:60-135 / store_v2.rs:50-110: address = rs1 + sign-extended offset (mod 2^32) over u16 limbs; the loaded word is the memory word;
    every timestamp difference re-adds to 2^MAX_TS_BITS + prev - now"""
    n = 600
    d = _mem_steps(n, is_store)
    recs = _mem_records(d, is_store)
    got, lkd, lkf = po.witgen_mem(list(range(23)) + [23], is_store, recs, np.arange(n), 0, 0x1000, n)
    g = got.astype(np.int64)
    assert np.array_equal(g[:, 0], d["pcs"].astype(np.int64)) and np.array_equal(g[:, 1], d["cycles"].astype(np.int64))
    if is_store:
        rs1, rs2, imm, sign, prev, addr = g[:, 13] + (g[:, 14] << 16), g[:, 15] + (g[:, 16] << 16), g[:, 17], g[:, 18], g[:, 19] + (g[:, 20] << 16), g[:, 21] + (g[:, 22] << 16)
        assert np.array_equal(rs2, d["rs2_vals"].astype(np.int64)) and np.array_equal(prev, d["mem_before"].astype(np.int64))
        assert np.array_equal(g[:, 2], np.full(n, 2)) and np.array_equal(g[:, 6], np.full(n, 3))
        mem_prev, mem_diff = g[:, 10], g[:, 11] + (g[:, 12] << 16)
    else:
        rs1, imm, sign, addr, word = g[:, 15] + (g[:, 16] << 16), g[:, 17], g[:, 18], g[:, 19] + (g[:, 20] << 16), g[:, 21] + (g[:, 22] << 16)
        assert np.array_equal(word, d["mem_before"].astype(np.int64))
        assert np.array_equal(g[:, 8] + (g[:, 9] << 16), d["rd_before"].astype(np.int64)) and np.array_equal(g[:, 6], np.full(n, 4))
        mem_prev, mem_diff = g[:, 12], g[:, 13] + (g[:, 14] << 16)
    assert np.array_equal(rs1, d["rs1_vals"].astype(np.int64))
    assert np.array_equal(sign, (d["imms"] < 0).astype(np.int64)) and np.array_equal(imm, d["imms"] & 0xFFFF)
    assert np.array_equal((rs1 + imm + sign * 0xFFFF0000) & 0xFFFFFFFF, addr) and np.array_equal(addr, d["mem_addrs"].astype(np.int64))
    assert not (addr & 3).any() and (addr >> 30 == 0).all()
    assert np.array_equal(mem_prev, d["mem_prev_cycles"].astype(np.int64))
    assert np.array_equal(mem_diff, (1 << 29) + mem_prev - (g[:, 1] + 3))
    # 2 lookups per timestamp comparison (3 of them), 2 for the address, 2 more for a store's previous memory word
    assert int(lkd.sum()) == (10 if is_store else 8) * n and int(lkf.sum()) == n
    assert int(lkd[(1 << 14):(1 << 15)].sum()) >= 2 * n  # the 14-bit table takes the address checks (and shares no key with the others)


def test_lw_sw_oracle_rejects_other_record_shapes():
    d = _mem_steps(16, False)
    with pytest.raises(ValueError):
        po.witgen_mem(list(range(23)) + [23], True, _mem_records(d, False), np.arange(16))     # a load record has no rs2
    with pytest.raises(ValueError):
        po.witgen_mem(list(range(23)) + [23], False, _mem_records(d, True), np.arange(16))     # a store record has no rd
    with pytest.raises(ValueError):
        po.witgen_mem(list(range(23)) + [22], False, _mem_records(d, False), np.arange(16))    # column id out of range


def _jalr_steps(n):
    """chips/jalr.rs tests' shape: rs1 = 0x1000 + 8 i, offsets 0, 4, -4, 100 (signed), rd <- pc + 4; plus edge cases (an odd target: bit 0 witnessed
    and dropped from next_pc, bit 1 set, the smallest and largest offsets, a wrapped sum, a target at the top of the 30-bit range)"""
    i = np.arange(n, dtype=np.int64)
    pc = 0x2000 + 4 * i
    rs1 = 0x1000 + 8 * i
    imm = np.array([0, 4, -4, 100], dtype=np.int64)[i % 4]
    if n >= 8:
        rs1[1], imm[1] = 0x3001, 0                 # odd target
        rs1[2], imm[2] = 0x4000, 2                 # bit 1 set
        rs1[3], imm[3] = 0x0001_0800, -2048        # borrow from the high limb
        rs1[5], imm[5] = 0xFFFF_FFFE, 6            # the sum wraps to 4
        rs1[6], imm[6] = 0x3FFF_F800, 2047         # 0x3FFFFFFF: every checked bit set
    target = (rs1 + imm) & 0xFFFFFFFF
    return dict(cycles=(4 + 4 * i).astype(np.uint64), pcs=pc.astype(np.uint64), pcs_after=(target & ~np.int64(1)).astype(np.uint64), imms=imm,
                rs1_vals=rs1.astype(np.uint64), rd_before=(i % 97).astype(np.uint64), rd_after=(pc + 4).astype(np.uint64),
                prev_cycles=np.zeros(n, dtype=np.uint64), target=target)


def test_jalr_oracle_rows_satisfy_the_circuit_relations():
    """jalr_v2.rs:60-135: rs1 + sign-extended imm = target over u16 limbs with bit carries; next_pc = target with bit 0 cleared; rd = pc + 4"""
    n = 300
    d = _jalr_steps(n)
    recs = po.step_records_jalr(d["cycles"], d["pcs"], d["pcs_after"], 2, 4, d["imms"], d["rs1_vals"], d["rd_before"], d["rd_after"], d["prev_cycles"])
    got, lkd, lkf = po.witgen_jalr(list(range(22)) + [22], recs, np.arange(n), 0, 0x2000, n)
    g = got.astype(np.int64)
    rs1, imm, sign = g[:, 13] + (g[:, 14] << 16), g[:, 15], g[:, 16]
    target, b0, b1, rd_high = g[:, 17] + (g[:, 18] << 16), g[:, 19], g[:, 20], g[:, 21]
    assert np.array_equal(rs1, d["rs1_vals"].astype(np.int64)) and np.array_equal(target, d["target"])
    c0 = g[:, 13] + imm - g[:, 17]
    assert set(np.unique(c0)) <= {0, 1 << 16}                                           # carry_lo_bit
    c1 = g[:, 14] + sign * 0xFFFF + (c0 >> 16) - g[:, 18]
    assert set(np.unique(c1)) <= {0, 1 << 16}                                           # overflow_bit
    assert np.array_equal(g[:, 1], target - b0) and np.array_equal(b0, target & 1) and np.array_equal(b1, (target >> 1) & 1)
    assert np.array_equal((rd_high << 16) + ((g[:, 0] + 4) & 0xFFFF), g[:, 0] + 4)      # rd_low = pc + 4 - rd_high 2^16 is a u16
    assert np.array_equal(g[:, 9] + (g[:, 10] << 16), d["rd_before"].astype(np.int64)) and (g[:, 3] == 2).all() and (g[:, 7] == 4).all()
    assert int(lkd.sum()) == 8 * n and int(lkf.sum()) == n
    assert int(lkd[(1 << 14):(1 << 15)].sum()) == 3 * n                                 # rd_high, the target's middle bits and high limb


def test_jalr_oracle_rejects_bad_maps_and_records():
    d = _jalr_steps(8)
    recs = po.step_records_jalr(d["cycles"], d["pcs"], d["pcs_after"], 2, 4, d["imms"], d["rs1_vals"], d["rd_before"], d["rd_after"], d["prev_cycles"])
    with pytest.raises(ValueError):
        po.witgen_jalr(list(range(22)) + [21], recs, np.arange(8))
    jd = _jal_steps(8)
    jrecs = po.step_records_j(jd["cycles"], jd["pcs"], jd["pcs_after"], po.INSN_JAL, 1, jd["imms"], jd["rd_before"], jd["rd_after"], jd["prev_cycles"])
    with pytest.raises(ValueError):
        po.witgen_jalr(list(range(22)) + [22], jrecs, np.arange(8))                     # a J-type record has no rs1


def _shift_steps(n, kind, is_imm):
    """chips/shift_r.rs / shift_i.rs tests' shape: rs1 = 0x12345678-like words, every shift amount 0..31 cycled; plus edge cases (a negative operand
    under SRA, all ones, zero, and for the register form amounts above 31: only the low five bits of rs2's low byte shift, bits 5..7 are range-checked)"""
    i = np.arange(n, dtype=np.int64)
    a = (0x12345678 + 0x01010101 * i * 7) & 0xFFFFFFFF
    if n >= 8:
        a[1], a[2], a[3] = 0xFFFFFFFF, 0, 0x80000000
        a[5] = 0xF00F0FF0
    amount = i % 32
    c = amount.copy()
    if not is_imm:
        c = (c + 32 * (i % 8)) & 0xFF                  # rs2's low byte above 31
        c = c + ((i * 0x10100) & 0xFFFFFF00)           # and anything in its upper bytes
    sa = np.where(a >> 31, a - (1 << 32), a)
    res = {0: (a << amount) & 0xFFFFFFFF, 1: a >> amount, 2: (sa >> amount) & 0xFFFFFFFF}[kind]
    return dict(cycles=(4 + 4 * i).astype(np.uint64), pcs=(0x1000 + 4 * i).astype(np.uint64), rs1_vals=a.astype(np.uint64), rs2_vals=c.astype(np.uint64),
                imms=c, rd_before=(i % 41).astype(np.uint64), rd_after=res.astype(np.uint64), prev_cycles=np.zeros(n, dtype=np.uint64))


def _shift_records(d, kind, is_imm):
    if is_imm:
        return po.step_records_i(d["cycles"], d["pcs"], [po.INSN_SLLI, po.INSN_SRLI, po.INSN_SRAI][kind], 2, 4, d["imms"], d["rs1_vals"], d["rd_before"],
                                 d["rd_after"], d["prev_cycles"])
    return po.step_records_r(d["cycles"], d["pcs"], [po.INSN_SLL, po.INSN_SRL, po.INSN_SRA][kind], 2, 3, 4, d["rs1_vals"], d["rs2_vals"], d["rd_before"],
                             d["rd_after"], d["prev_cycles"])


@pytest.mark.parametrize("is_imm", [False, True])
@pytest.mark.parametrize("kind", [0, 1, 2])
def test_shift_oracle_rows_satisfy_the_shift_base_constraints(kind, is_imm):
    """ShiftBaseConfig::construct_circuit (shift_circuit_v2.rs:71-200): one-hot markers, the multiplier column, and per result byte
    a[i] = b[i - limb_shift] 2^bit_shift - 256 carry[i - limb_shift] + carry[i - limb_shift - 1]  (left) or
    a[i] 2^bit_shift = b[i + limb_shift] - carry[i + limb_shift] + 256 carry-or-sign-fill above  (right), under the active limb marker"""
    n = 512
    d = _shift_steps(n, kind, is_imm)
    recs = _shift_records(d, kind, is_imm)
    nc = 40 if is_imm else 47
    got, lkd, lkf, lk2, lkx = po.witgen_shift(list(range(nc)) + [nc], is_imm, kind, recs, np.arange(n), 0, 0x1000, n)
    g = got.astype(np.int64)
    o = 12 if is_imm else 16                              # first byte column
    b = g[:, o:o + 4]
    if is_imm:
        a, c_low = g[:, o + 4:o + 8], g[:, o + 8] & 0xFF
        assert np.array_equal(g[:, o + 8], d["imms"] & 0xFFFF)
        m = o + 9
    else:
        cb, a = g[:, o + 4:o + 8], g[:, o + 8:o + 12]
        assert np.array_equal(sum(cb[:, k] << (8 * k) for k in range(4)), d["rs2_vals"].astype(np.int64))
        c_low = cb[:, 0]
        m = o + 12
    bit_m, limb_m, ml, mr, sign, carry = g[:, m:m + 8], g[:, m + 8:m + 12], g[:, m + 12], g[:, m + 13], g[:, m + 14], g[:, m + 15:m + 19]
    assert np.array_equal(sum(b[:, k] << (8 * k) for k in range(4)), d["rs1_vals"].astype(np.int64))
    assert np.array_equal(sum(a[:, k] << (8 * k) for k in range(4)), d["rd_after"].astype(np.int64))
    assert (bit_m.sum(axis=1) == 1).all() and (limb_m.sum(axis=1) == 1).all()
    bit_shift, limb_shift = bit_m.argmax(axis=1), limb_m.argmax(axis=1)
    assert np.array_equal(bit_shift + 8 * limb_shift, c_low % 32)
    assert np.array_equal(ml, np.where(kind == 0, 1 << bit_shift, 0)) and np.array_equal(mr, np.where(kind == 0, 0, 1 << bit_shift))
    assert np.array_equal(sign, (b[:, 3] >> 7) if kind == 2 else np.zeros(n, dtype=np.int64))
    assert (carry < (1 << bit_shift)[:, None]).all()
    for r in range(n):
        ls, bs = int(limb_shift[r]), int(bit_shift[r])
        for i in range(4):
            if kind == 0:
                exp = 0 if i < ls else (int(b[r, i - ls]) << bs) - 256 * int(carry[r, i - ls]) + (int(carry[r, i - ls - 1]) if i > ls else 0)
                assert int(a[r, i]) == exp
            else:
                fill = 255 * int(sign[r])
                if i + ls >= 4:
                    assert int(a[r, i]) == fill
                else:
                    above = int(carry[r, i + ls + 1]) if i + ls + 1 < 4 else fill % (1 << bs) if bs else 0
                    # a[i] 2^bs + carry[i + ls] = b[i + ls] + 2^8 (bits arriving from above)
                    assert (int(a[r, i]) << bs) + int(carry[r, i + ls]) == int(b[r, i + ls]) + 256 * above
    # lookups: 2 per timestamp comparison, 4 carries, the amount's upper bits; the result's two byte pairs; SRA's sign
    n_ts = 2 if is_imm else 3
    assert int(lkd.sum()) == (2 * n_ts + 5) * n and int(lkf.sum()) == n and int(lk2.sum()) == 2 * n and int(lkx.sum()) == (n if kind == 2 else 0)
    assert int(lkd[8:16].sum()) == n + 4 * int((bit_shift == 3).sum())   # the 3-bit range of (c[0] - shift) >> 5, and the carries of 3-bit shifts


def _sub_store_steps(n, kind):
    """SH (kind 2) / SB (kind 3): base 0x1000 + 16 i, offsets that visit every halfword / byte of a word, the stored value's low limb with both bytes set"""
    d = _mem_steps(n, True)
    i = np.arange(n, dtype=np.int64)
    off = (2 * (i % 2)) if kind == 2 else (i % 4)
    rs1 = 0x1000 + 16 * i
    imm = np.array([0, 4, -4, -8], dtype=np.int64)[(i // 4) % 4] + off
    rs2 = (0xA1B2C3D4 + 0x01030507 * i) & 0xFFFFFFFF
    before = (0x11223344 + 0x10101010 * i) & 0xFFFFFFFF
    addr = (rs1 + imm) & 0xFFFFFFFF
    sh = 8 * (addr & 3)
    mask = np.where(kind == 2, 0xFFFF, 0xFF) << sh
    after = (before & ~mask & 0xFFFFFFFF) | ((rs2 << sh) & mask)
    d.update(rs1_vals=rs1.astype(np.uint64), imms=imm, rs2_vals=rs2.astype(np.uint64), mem_addrs=addr.astype(np.uint64), mem_before=before.astype(np.uint64),
             mem_after=after.astype(np.uint64))
    return d


def _sub_store_records(d, kind):
    return po.step_records_mem(True, d["cycles"], d["pcs"], po.INSN_SH if kind == 2 else po.INSN_SB, 2, 3, d["imms"], d["rs1_vals"], d["rs2_vals"],
                               d["rd_before"], d["rd_after"], d["mem_addrs"], d["mem_before"], d["mem_after"], d["prev_cycles"], d["mem_prev_cycles"])


@pytest.mark.parametrize("kind", [2, 3])
def test_sh_sb_oracle_rows_satisfy_the_circuit_relations(kind):
    """store_v2.rs:60-135 + MemWordUtil (memory/gadget.rs:30-130): the address bits select the limb (bit 1) and the byte (bit 0); SH replaces the selected
    limb by rs2's low limb, SB replaces one byte of it (expected_limb); the rest of the row is SW's"""
    n = 400
    d = _sub_store_steps(n, kind)
    recs = _sub_store_records(d, kind)
    nc = 24 if kind == 2 else 29
    got, lkd, lkf = po.witgen_mem(list(range(nc)) + [nc], kind, recs, np.arange(n), 0, 0x1000, n)
    ref, rlkd, _ = po.witgen_mem(list(range(23)) + [23], True, recs, np.arange(n), 0, 0x1000, n)
    assert np.array_equal(got[:, :23], ref)                                     # SW's columns
    g = got.astype(np.int64)
    addr, prev, rs2 = g[:, 21] + (g[:, 22] << 16), g[:, 19] + (g[:, 20] << 16), g[:, 15] + (g[:, 16] << 16)
    after = d["mem_after"].astype(np.int64)
    if kind == 2:
        bit1 = g[:, 23]
        assert np.array_equal(bit1, (addr >> 1) & 1) and not (addr & 1).any()
        new_limbs = [np.where(bit1 == k, rs2 & 0xFFFF, (prev >> (16 * k)) & 0xFFFF) for k in range(2)]
        assert np.array_equal(int(lkd.sum()), int(rlkd.sum()))
    else:
        bit0, bit1, pb0, pb1, sb, exp = (g[:, 23 + k] for k in range(6))
        assert np.array_equal(bit0 + 2 * bit1, addr & 3)
        limb = np.where(bit1 == 1, prev >> 16, prev & 0xFFFF)
        assert np.array_equal(pb0 + (pb1 << 8), limb) and np.array_equal(sb, rs2 & 0xFF)
        assert np.array_equal(exp, np.where(bit0 == 1, (sb << 8) + pb0, (pb1 << 8) + sb))
        new_limbs = [np.where(bit1 == k, exp, (prev >> (16 * k)) & 0xFFFF) for k in range(2)]
        assert int(lkd.sum()) == int(rlkd.sum()) + 4 * n and int(lkd[256:512].sum()) == 4 * n
    assert np.array_equal(new_limbs[0] + (new_limbs[1] << 16), after)           # what the memory write records
    assert int(lkf.sum()) == n


def _sub_load_steps(n, width, signed):
    """LH / LHU / LB / LBU: offsets that visit every halfword / byte of a word, memory words with set and clear sign bits in every position"""
    d = _mem_steps(n, False)
    i = np.arange(n, dtype=np.int64)
    off = (2 * (i % 2)) if width == 16 else (i % 4)
    rs1 = 0x1000 + 16 * i
    imm = np.array([0, 4, -4, -8], dtype=np.int64)[(i // 4) % 4] + off
    word = (0x7F80FF01 + 0x01810283 * i) & 0xFFFFFFFF
    addr = (rs1 + imm) & 0xFFFFFFFF
    mask = 0xFFFF if width == 16 else 0xFF
    val = (word >> (8 * (addr & 3))) & mask
    if signed:
        val = np.where(val >> (width - 1), val | (0xFFFFFFFF ^ mask), val)
    d.update(rs1_vals=rs1.astype(np.uint64), imms=imm, mem_addrs=addr.astype(np.uint64), mem_before=word.astype(np.uint64), mem_after=word.astype(np.uint64),
             rd_after=val.astype(np.uint64), loaded=val)
    return d


def _sub_load_records(d, width, signed):
    kind = {(16, True): po.INSN_LH, (16, False): po.INSN_LHU, (8, True): po.INSN_LB, (8, False): po.INSN_LBU}[(width, signed)]
    return po.step_records_mem(False, d["cycles"], d["pcs"], kind, 2, 4, d["imms"], d["rs1_vals"], d["rs2_vals"], d["rd_before"], d["rd_after"], d["mem_addrs"],
                               d["mem_before"], d["mem_after"], d["prev_cycles"], d["mem_prev_cycles"])


@pytest.mark.parametrize("signed", [False, True])
@pytest.mark.parametrize("width", [16, 8])
def test_load_sub_oracle_rows_satisfy_the_circuit_relations(width, signed):
    """load_v2.rs:95-170: target_limb = memory limb[bit 1]; byte loads: target_limb = target_byte * 2^(8 bit0) + dummy_byte * 2^(8 (1 - bit0));
    the value written to rd is the loaded value extended by msb"""
    n = 400
    d = _sub_load_steps(n, width, signed)
    recs = _sub_load_records(d, width, signed)
    nc = 25 + (3 if width == 8 else 0) + int(signed)
    cols = po.load_sub_cols(range(nc), width, signed, nc)
    got, lkd, lkf = po.witgen_load_sub(cols, width, signed, recs, np.arange(n), 0, 0x1000, n)
    ref, rlkd, _ = po.witgen_mem(list(range(23)) + [23], False, recs, np.arange(n), 0, 0x1000, n)
    assert np.array_equal(got[:, :23], ref)                                          # LW's columns
    g = got.astype(np.int64)
    addr, word = g[:, 19] + (g[:, 20] << 16), g[:, 21] + (g[:, 22] << 16)
    bit1, limb = g[:, 23], g[:, 24]
    assert np.array_equal(bit1, (addr >> 1) & 1) and np.array_equal(limb, np.where(bit1 == 1, word >> 16, word & 0xFFFF))
    val, extra = limb, 0
    if width == 8:
        bit0, target, other = g[:, 25], g[:, 26], g[:, 27]
        assert np.array_equal(bit0, addr & 1)
        assert np.array_equal(limb, np.where(bit0 == 1, (target << 8) + other, (other << 8) + target))
        val, extra = target, 2
    else:
        assert not (addr & 1).any()
    if signed:
        msb = g[:, nc - 1]
        assert np.array_equal(msb, val >> (width - 1))
        val = val + msb * ((1 << 32) - (1 << width))
        extra += 1
        assert int(lkd[(1 << width):(2 << width)].sum()) >= n
    assert np.array_equal(val, d["loaded"])                                           # what the register write records
    assert int(lkd.sum()) == int(rlkd.sum()) + extra * n and int(lkf.sum()) == n


def test_load_sub_oracle_rejects_columns_a_variant_does_not_have():
    d = _sub_load_steps(8, 16, False)
    recs = _sub_load_records(d, 16, False)
    with pytest.raises(ValueError):
        po.witgen_load_sub(list(range(29)) + [29], 16, False, recs, np.arange(8))       # LHU with byte and msb columns
    with pytest.raises(ValueError):
        po.witgen_load_sub(po.load_sub_cols(range(25), 16, False, 25), 12, False, recs, np.arange(8))


def _mul_steps(n, kind):
    """chips/mul.rs tests' shape (rs1 = 3 i + 1-like words) plus operands with set sign bits and limb-boundary values"""
    i = np.arange(n, dtype=np.int64)
    a = (0x9E3779B1 * (i + 1)) & 0xFFFFFFFF
    b = (0x85EBCA77 * (i + 3)) & 0xFFFFFFFF
    if n >= 10:
        a[:10] = [0, 1, 0xFFFFFFFF, 0x80000000, 0x7FFFFFFF, 0xFFFF0000, 0x0000FFFF, 0x80000000, 0xFFFFFFFF, 0x00010000]
        b[:10] = [5, 0xFFFFFFFF, 0xFFFFFFFF, 0x80000000, 0x7FFFFFFF, 0x0000FFFF, 0xFFFF0000, 1, 1, 0x00010000]
    sa, sb = np.where(a >> 31, a - (1 << 32), a), np.where(b >> 31, b - (1 << 32), b)
    full = {0: a.astype(object) * b.astype(object), 1: sa.astype(object) * sb.astype(object), 2: a.astype(object) * b.astype(object),
            3: sa.astype(object) * b.astype(object)}[kind]
    prod = np.array([int(v) % (1 << 64) for v in full], dtype=object)
    rd = np.array([int(v) & 0xFFFFFFFF if kind == 0 else int(v) >> 32 for v in prod], dtype=np.uint64)
    return dict(cycles=(4 + 4 * i).astype(np.uint64), pcs=(0x1000 + 4 * i).astype(np.uint64), rs1_vals=a.astype(np.uint64), rs2_vals=b.astype(np.uint64),
                rd_before=(i % 53).astype(np.uint64), rd_after=rd, prev_cycles=np.zeros(n, dtype=np.uint64), prod=prod)


def _mul_cols(ids, kind, num_cols):
    ids = list(ids)
    return [ids.pop(0) for _ in range(22)] + ([ids.pop(0) for _ in range(4)] if kind else [po.NO_COLUMN] * 4) + [num_cols]


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_mul_oracle_rows_hold_the_full_product(kind):
    """mulh_circuit_v2.rs:60-200: rd_low | rd_high are the limbs of rs1 * rs2 with the signedness of the opcode (MULH: both signed, MULHSU: rs1 signed,
    MULHU / MUL: unsigned), the extensions are 0 or 0xffff by the operands' signs"""
    n = 300
    d = _mul_steps(n, kind)
    recs = po.step_records_r(d["cycles"], d["pcs"], [po.INSN_MUL, po.INSN_MULH, po.INSN_MULHU, po.INSN_MULHSU][kind], 2, 3, 4, d["rs1_vals"], d["rs2_vals"],
                             d["rd_before"], d["rd_after"], d["prev_cycles"])
    nc = 26 if kind else 22
    got, lkd, lkf = po.witgen_mul(_mul_cols(range(nc), kind, nc), kind, recs, np.arange(n), 0, 0x1000, n)
    g = got.astype(np.int64)
    assert np.array_equal(g[:, 16] + (g[:, 17] << 16), d["rs1_vals"].astype(np.int64)) and np.array_equal(g[:, 18] + (g[:, 19] << 16), d["rs2_vals"].astype(np.int64))
    low = g[:, 20] + (g[:, 21] << 16)
    assert [int(v) for v in low] == [int(p) & 0xFFFFFFFF for p in d["prod"]]
    n_lk = 6 + 4
    if kind:
        high = g[:, 22] + (g[:, 23] << 16)
        assert [int(v) for v in high] == [int(p) >> 32 for p in d["prod"]]
        assert np.array_equal(high, d["rd_after"].astype(np.int64))
        s1, s2 = d["rs1_vals"].astype(np.int64) >> 31, d["rs2_vals"].astype(np.int64) >> 31
        assert np.array_equal(g[:, 24], np.where(kind == 2, 0, s1 * 0xFFFF)) and np.array_equal(g[:, 25], np.where(kind == 1, s2 * 0xFFFF, 0))
        n_lk += 4 + (2 if kind in (1, 3) else 0)
    else:
        assert np.array_equal(low, d["rd_after"].astype(np.int64))
    assert int(lkd.sum()) == n_lk * n and int(lkf.sum()) == n
    assert int(lkd[(1 << 18):].sum()) == (4 if kind else 2) * n                          # the carries: 18-bit range lookups


def test_mul_oracle_rejects_option_columns_on_mul():
    d = _mul_steps(8, 0)
    recs = po.step_records_r(d["cycles"], d["pcs"], po.INSN_MUL, 2, 3, 4, d["rs1_vals"], d["rs2_vals"], d["rd_before"], d["rd_after"], d["prev_cycles"])
    with pytest.raises(ValueError):
        po.witgen_mul(list(range(26)) + [26], 0, recs, np.arange(8))
    with pytest.raises(ValueError):
        po.witgen_mul(_mul_cols(range(22), 0, 22), 4, recs, np.arange(8))


def _div_steps(n, kind):
    """operands that reach every branch of run_divrem: a zero divisor, the signed overflow (INT_MIN / -1), exact divisions (remainder 0), both signs of
    both operands, |dividend| < |divisor|, divisors of one limb and of two"""
    i = np.arange(n, dtype=np.int64)
    a = (0x9E3779B1 * (i + 1)) & 0xFFFFFFFF
    b = np.where(i % 3 == 0, (0x85EBCA77 * (i + 3)) & 0xFFFF, (0x85EBCA77 * (i + 3)) & 0xFFFFFFFF)
    b = np.where(i % 5 == 0, (0 - b) & 0xFFFFFFFF, b)
    b[b == 0] = 7
    if n >= 12:
        a[:12] = [100, 0x80000000, 0x80000000, 7, 0xFFFFFFF9, 0xFFFFFFF9, 21, 0, 5, 0xFFFFFFFF, 0x7FFFFFFF, 0x00010000]
        b[:12] = [0, 0xFFFFFFFF, 1, 0xFFFFFFFE, 2, 0xFFFFFFFE, 7, 9, 0x00020000, 0xFFFFFFFF, 0x80000000, 0x0000FFFF]
    signed = kind in (0, 2)
    sa = np.where((a >> 31) & signed, a - (1 << 32), a)
    sb = np.where((b >> 31) & signed, b - (1 << 32), b)
    q, r = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
    for k in range(n):
        x, y = int(sa[k]), int(sb[k])
        if y == 0:
            qq, rr = -1, x
        elif signed and x == -(1 << 31) and y == -1:
            qq, rr = x, 0
        else:
            qq = abs(x) // abs(y) * (1 if (x < 0) == (y < 0) else -1)       # C division: truncates toward zero
            rr = x - qq * y
        q[k], r[k] = qq & 0xFFFFFFFF, rr & 0xFFFFFFFF
    rd = q if kind in (0, 1) else r
    return dict(cycles=(4 + 4 * i).astype(np.uint64), pcs=(0x1000 + 4 * i).astype(np.uint64), rs1_vals=a.astype(np.uint64), rs2_vals=b.astype(np.uint64),
                rd_before=(i % 59).astype(np.uint64), rd_after=rd.astype(np.uint64), prev_cycles=np.zeros(n, dtype=np.uint64), q=q, r=r)


def _div_records(d, kind):
    return po.step_records_r(d["cycles"], d["pcs"], [po.INSN_DIV, po.INSN_DIVU, po.INSN_REM, po.INSN_REMU][kind], 2, 3, 4, d["rs1_vals"], d["rs2_vals"],
                             d["rd_before"], d["rd_after"], d["prev_cycles"])


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_div_oracle_rows_satisfy_the_circuit_relations(kind):
    """div_circuit_v2.rs:60-380: quotient and remainder are RISC-V's; the zero flags are proved by their inverse witnesses (flag = 1 - sum * inv);
    remainder' carries the divisor's sign and differs from the divisor at the marked limb by lt_diff; (remainder'_i - 2^16) * remainder_inv_i = 1"""
    n = 400
    d = _div_steps(n, kind)
    recs = _div_records(d, kind)
    got, lkd, lkf = po.witgen_div(list(range(39)) + [39], kind, recs, np.arange(n), 0, 0x1000, n)
    signed = kind in (0, 2)
    for r in range(n):
        row = [int(v) for v in got[r]]
        x, y, q, rem = (row[16 + 2 * k] + (row[17 + 2 * k] << 16) for k in range(4))
        assert (x, y, q, rem) == (int(d["rs1_vals"][r]), int(d["rs2_vals"][r]), int(d["q"][r]), int(d["r"][r]))
        xs, ys, qs, rz, dz = row[24:29]
        assert xs == (signed and x >> 31) and ys == (signed and y >> 31) and dz == (y == 0)
        overflow = signed and x == 0x80000000 and y == 0xFFFFFFFF
        # a zero divisor's quotient -1 counts as negative when signed; the overflow's quotient INT_MIN does not (run_divrem :663-671)
        assert qs == (signed and not overflow and (q >> 31 or y == 0))
        assert rz == (rem == 0 and y != 0)
        dsum, rsum = row[18] + row[19], row[22] + row[23]
        assert (dsum * row[29]) % P_GL == (0 if dsum == 0 else 1) and (rsum * row[30]) % P_GL == (0 if rsum == 0 else 1)
        sx, rp = row[33], row[34] + (row[35] << 16)
        assert sx == (xs ^ ys) and rp == ((-rem) & 0xFFFFFFFF if sx else rem)
        for k in range(2):
            assert ((row[34 + k] - (1 << 16)) % P_GL) * row[31 + k] % P_GL == 1
        m0, m1, lt = row[36], row[37], row[38]
        special = y == 0 or (signed and x == 0x80000000 and y == 0xFFFFFFFF)
        if special or rz:
            assert (m0, m1, lt) == (0, 0, 0)
        else:
            assert m0 + m1 == 1
            k = 1 if m1 else 0
            assert all(row[18 + j] == row[34 + j] for j in range(k + 1, 2))                  # equal above the marked limb
            assert lt == (row[34 + k] - row[18 + k] if ys else row[18 + k] - row[34 + k]) and lt >= 1
    assert int(lkd.sum()) == (6 + 4 + 4 + 1 + (2 if signed else 0)) * n and int(lkf.sum()) == n and int(lkd[(1 << 18):].sum()) == 4 * n
    # DIV and REM (DIVU and REMU) assign the same row
    other, _, _ = po.witgen_div(list(range(39)) + [39], {0: 2, 2: 0, 1: 3, 3: 1}[kind], recs, np.arange(n), 0, 0x1000, n)
    assert np.array_equal(other, got)
