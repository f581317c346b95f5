"""Shared inputs for the ADD / SUB witness-generation tests.

`reference_test_steps` reproduces the INPUT DATA of the reference's own test of this path
(make_test_steps, ceno_zkvm/src/instructions/gpu/chips/add.rs:62-100): the eight edge-case operand pairs, then
((i % 1000) + 1, (i % 500) + 3); rd_before = i % 200; cycle = 4 + 4 i; pc = 0x1000 + 4 i; registers x2, x3 -> x4;
previous cycle 0.  `model_row` is an independent pure-Python statement of the assignment, written from the circuit's
constraints rather than from the C restatement."""
import numpy as np

EDGE_CASES = [(0, 0), (0, 1), (1, 0), (0xFFFFFFFF, 1), (0xFFFFFFFF, 0xFFFFFFFF), (0x80000000, 0x80000000), (0x7FFFFFFF, 1),
              (0xFFFF0000, 0x0000FFFF)]
# column ids in AddColumnMap order for a circuit that creates its witnesses in this order (illustrative; the real ids come
# from the Rust circuit builder through extract_add_column_map and are passed in by the caller)
NATURAL_COLS = list(range(22)) + [22]


def reference_test_steps(n, sub=False):
    i = np.arange(n, dtype=np.uint64)
    a = (i % 1000 + 1).astype(np.uint64)
    b = (i % 500 + 3).astype(np.uint64)
    for k, (x, y) in enumerate(EDGE_CASES[:n]):
        a[k], b[k] = x, y
    rd_before = i % 200
    if sub:
        rd_after = (a - b) & np.uint64(0xFFFFFFFF)
    else:
        rd_after = (a + b) & np.uint64(0xFFFFFFFF)
    return dict(cycles=4 + 4 * i, pcs=0x1000 + 4 * i, rs1_vals=a, rs2_vals=b, rd_before=rd_before, rd_after=rd_after,
                prev_cycles=np.zeros(n, dtype=np.uint64))


def model_row(cols, sub, cycle, pc, rs1, rs2, rd, v1, v2, rd_before, rd_after, prev, offset):
    """one row as {column id: value} plus the list of (table, key) lookups, from first principles"""
    row, lk = {}, []
    ts = cycle - offset
    row[cols[0]], row[cols[1]] = pc, ts
    lk.append(("fetch", pc))

    def access(base, reg, sub_cycle, extra=()):
        p = max(prev - offset, 0)
        p = 0 if p < 4 else p
        rhs = ts + sub_cycle
        diff = p - rhs + ((1 << 29) if p < rhs else 0)
        assert 0 <= diff < (1 << 29)
        row[cols[base]], row[cols[base + 1]] = reg, p
        k = base + 2
        for e in extra:
            row[cols[k]] = e
            k += 1
        row[cols[k]], row[cols[k + 1]] = diff & 0xFFFF, diff >> 16
        lk.append(("dyn", (1 << 16) + (diff & 0xFFFF)))
        lk.append(("dyn", (1 << 13) + (diff >> 16)))

    access(2, rs1, 0)
    access(6, rs2, 1)
    access(10, rd, 2, extra=(rd_before & 0xFFFF, rd_before >> 16))
    x, y = (v2, rd_after) if sub else (v1, v2)
    if sub:
        lk += [("dyn", (1 << 16) + (rd_after & 0xFFFF)), ("dyn", (1 << 16) + (rd_after >> 16))]
    row[cols[16]], row[cols[17]], row[cols[18]], row[cols[19]] = x & 0xFFFF, x >> 16, y & 0xFFFF, y >> 16
    s = x + y
    c0 = ((x & 0xFFFF) + (y & 0xFFFF)) >> 16
    c1 = s >> 32
    row[cols[20]], row[cols[21]] = c0, c1
    res = s & 0xFFFFFFFF
    lk += [("dyn", (1 << 16) + (res & 0xFFFF)), ("dyn", (1 << 16) + (res >> 16))]
    return row, lk


# ---- AND / OR / XOR (chips/logic_r.rs:85-118): eight edge cases, then (0xDEAD0000 | i, 0x00FFFF00 | i << 8) ----
LOGIC_EDGE_CASES = [(0, 0), (0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0), (0, 0xFFFFFFFF), (0xAAAAAAAA, 0x55555555), (0xFFFF0000, 0x0000FFFF),
                    (0xDEADBEEF, 0xFFFFFFFF), (0x12345678, 0)]
LOGIC_NATURAL_COLS = list(range(28)) + [28]
LOGIC_OPS = {0: lambda a, b: a & b, 1: lambda a, b: a | b, 2: lambda a, b: a ^ b}


def reference_logic_steps(n, kind=0):
    i = np.arange(n, dtype=np.uint64)
    a = (np.uint64(0xDEAD0000) | i) & np.uint64(0xFFFFFFFF)
    b = (np.uint64(0x00FFFF00) | (i << np.uint64(8))) & np.uint64(0xFFFFFFFF)
    for k, (x, y) in enumerate(LOGIC_EDGE_CASES[:n]):
        a[k], b[k] = x, y
    return dict(cycles=4 + 4 * i, pcs=0x1000 + 4 * i, rs1_vals=a, rs2_vals=b, rd_before=i % 200, rd_after=LOGIC_OPS[kind](a, b),
                prev_cycles=np.zeros(n, dtype=np.uint64))


def model_logic_row(cols, cycle, pc, rs1, rs2, rd, v1, v2, rd_before, rd_after, prev, offset):
    """one row of a logic chip as {column id: value} plus its (table, key) lookups, from the circuit's constraints: the R-instruction
    base of model_row and the byte decompositions; the table key of byte pair (a, b) is a | b << 8"""
    base, lk = model_row(list(cols[:16]) + [10 ** 6 + k for k in range(6)] + [cols[28]], False, cycle, pc, rs1, rs2, rd, 0, 0, rd_before, rd_after, prev, offset)
    row = {c: v for c, v in base.items() if c in set(cols[:16])}
    lk = [x for x in lk[:7]]  # fetch + the six timestamp-difference limbs (the arithmetic chips' own lookups follow them)
    for b in range(4):
        row[cols[16 + b]] = (v1 >> (8 * b)) & 0xFF
        row[cols[20 + b]] = (v2 >> (8 * b)) & 0xFF
        row[cols[24 + b]] = (rd_after >> (8 * b)) & 0xFF
        lk.append(("logic", ((v1 >> (8 * b)) & 0xFF) | (((v2 >> (8 * b)) & 0xFF) << 8)))
    return row, lk


# ---- ADDI (chips/addi.rs:80-97): rs1 = 137 i + 1, imm = i % 2048 - 1024, rd_before = i % 200 ----
ADDI_NATURAL_COLS = list(range(18)) + [18]


def reference_addi_steps(n):
    i = np.arange(n, dtype=np.int64)
    rs1 = ((i * 137 + 1) & 0xFFFFFFFF).astype(np.uint64)
    imm = (i % 2048 - 1024).astype(np.int64)
    rd_after = ((rs1.astype(np.int64) + imm) & 0xFFFFFFFF).astype(np.uint64)
    return dict(cycles=(4 + 4 * i).astype(np.uint64), pcs=(0x1000 + 4 * i).astype(np.uint64), rs1_vals=rs1, imms=imm, rd_before=(i % 200).astype(np.uint64),
                rd_after=rd_after, prev_cycles=np.zeros(n, dtype=np.uint64))


def model_addi_row(cols, cycle, pc, rs1, rd, v1, imm, rd_before, prev, offset):
    """one ADDI row as {column id: value} plus its lookups, from the constraints: rs1 + sign_extend(imm16) = rd with limb carries"""
    row, lk = {}, [("fetch", pc)]
    ts = cycle - offset
    row[cols[0]], row[cols[1]] = pc, ts

    def access(base, reg, sub_cycle, extra=()):
        p = max(prev - offset, 0)
        p = 0 if p < 4 else p
        rhs = ts + sub_cycle
        diff = p - rhs + ((1 << 29) if p < rhs else 0)
        row[cols[base]], row[cols[base + 1]] = reg, p
        k = base + 2
        for e in extra:
            row[cols[k]] = e
            k += 1
        row[cols[k]], row[cols[k + 1]] = diff & 0xFFFF, diff >> 16
        lk.append(("dyn", (1 << 16) + (diff & 0xFFFF)))
        lk.append(("dyn", (1 << 13) + (diff >> 16)))

    access(2, rs1, 0)
    access(6, rd, 2, extra=(rd_before & 0xFFFF, rd_before >> 16))
    imm16 = imm & 0xFFFF
    neg = 1 if imm16 & 0x8000 else 0
    ext = (0xFFFF0000 if neg else 0) | imm16
    row[cols[12]], row[cols[13]], row[cols[14]], row[cols[15]] = v1 & 0xFFFF, v1 >> 16, imm16, neg
    c0 = ((v1 & 0xFFFF) + imm16) >> 16
    c1 = (v1 + ext) >> 32
    row[cols[16]], row[cols[17]] = c0, c1
    res = (v1 + ext) & 0xFFFFFFFF
    lk += [("dyn", (1 << 16) + (res & 0xFFFF)), ("dyn", (1 << 16) + (res >> 16))]
    return row, lk


# ---- ANDI / ORI / XORI (chips/logic_i.rs:79-100): edge cases, then (i * 0x01010101 ^ 0xabed5eff, i % 4096) ----
LOGIC_I_EDGE_CASES = [(0, 0), (0xFFFFFFFF, 0xFFF), (0xFFFFFFFF, 0), (0, 0xFFF), (0xAAAAAAAA, 0x555), (0xFFFF0000, 0xFFF), (0x12345678, 0), (0xDEADBEEF, 0xABC)]
LOGIC_I_NATURAL_COLS = list(range(24)) + [24]


def reference_logic_i_steps(n, kind=0):
    i = np.arange(n, dtype=np.uint64)
    a = ((i * np.uint64(0x01010101)) & np.uint64(0xFFFFFFFF)) ^ np.uint64(0xABED5EFF)
    imm = (i % 4096).astype(np.int64)
    for k, (x, y) in enumerate(LOGIC_I_EDGE_CASES[:n]):
        a[k], imm[k] = x, y
    return dict(cycles=4 + 4 * i, pcs=0x1000 + 4 * i, rs1_vals=a, imms=imm, rd_before=i % 200,
                rd_after=LOGIC_OPS[kind](a, imm.astype(np.uint64) & np.uint64(0xFFFFFFFF)), prev_cycles=np.zeros(n, dtype=np.uint64))


def model_logic_i_row(cols, cycle, pc, rs1, rd, v1, imm, rd_before, rd_after, prev, offset):
    """one ANDI / ORI / XORI row: the I-instruction base of model_addi_row, byte columns, and the immediate as the circuit sees it: low
    half as is, high half 0xffff exactly when the (sign-extended, 32-bit) immediate is negative"""
    base, lk = model_addi_row(list(cols[:12]) + [10 ** 6 + k for k in range(6)] + [cols[24]], cycle, pc, rs1, rd, 0, 0, rd_before, prev, offset)
    row = {c: v for c, v in base.items() if c < 10 ** 6}
    lk = lk[:5]  # fetch + four timestamp-difference limbs
    imm32 = imm & 0xFFFFFFFF
    eff = (imm32 & 0xFFFF) | (0xFFFF0000 if imm32 >> 31 else 0)
    for b in range(4):
        row[cols[12 + b]] = (v1 >> (8 * b)) & 0xFF
        row[cols[16 + b]] = (rd_after >> (8 * b)) & 0xFF
        row[cols[20 + b]] = (eff >> (8 * b)) & 0xFF
        lk.append(("logic", ((v1 >> (8 * b)) & 0xFF) | (((eff >> (8 * b)) & 0xFF) << 8)))
    return row, lk
