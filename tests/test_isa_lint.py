"""ISA hazard lint over the gfx950 assembly that is assembled into libceno_hip.so (round-5 verdict item 7).

hipcc's hazard recogniser pads the code IT schedules; it does not look inside an `asm` statement, and it does not know what an `asm` statement
left pending when its own code resumes after `;;#ASMEND`.  Round 5 found that by accident: eleven inline-assembly 16-byte stores without their
wait states, boundary values corrupted in up to 87 % of the runs under load.  This test reads the device assembly the build
keeps beside every object (ceno_amd/_build/<unit>.hip.gfx950.s, written by the same compile that produced the object: -save-temps) and fails when

  R1  a VMEM store of more than 64 bits issued from an `asm` statement (`*_store_dwordx3 / x4`) is followed — inside the statement or in the
      compiler's code after it — by a VALU write of one of its data registers before two wait states have passed (the store reads its data late;
      /opt/skills/guides/cdna_hip_programming.md section 5.7 item 1: "an asm ..._store_dwordx3/x4 ends with `s_nop 1` inside the string");
  R2  a VALU instruction writes an SGPR pair (a carry-out of v_mad_u64_u32 / v_*_co_*, a v_cmp mask) and a VALU instruction reads that pair as
      carry-in, mask or operand fewer than two wait states later, with producer or consumer inside an `asm` statement (gfx90a+: "VALU writes SGPR ->
      VALU reads that SGPR: 2 wait states"; ceno_amd/csrc/gl64.hpp:109-114);
  R3  (the tree's own convention for its hand-written carry chains, gl64.hpp) inside ONE `asm` statement a VALU write of VCC is followed by a VALU
      read of VCC as carry-in or mask fewer than two wait states later.

CPU-only: no GPU is needed to read assembly."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STORE_RE = re.compile(r"^(global|flat|scratch|buffer)_store_dwordx[34]\b")
VREG_RE = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
SREG_RE = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")


def _vregs(tok):
    out = set()
    for m in VREG_RE.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def _sregs(tok):
    out = set()
    for m in SREG_RE.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    if re.search(r"\bvcc\b", tok):
        out.add("vcc")
    return out


def parse(path):
    """[(mnemonic, [operands], in_asm, asm_id, line number, text)] per function (split at labels that start a kernel / function body is not needed: the
    windows are two wait states long, a branch target in between only makes the check conservative)"""
    ins, in_asm, asm_id = [], False, 0
    for ln, raw in enumerate(open(path, errors="replace"), 1):
        line = raw.strip()
        if line.startswith(";;#ASMSTART"):
            in_asm, asm_id = True, asm_id + 1
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not raw.startswith("\t") or not line or line[0] in ".;#":
            if line.endswith(":") and not in_asm:
                ins.append(("<label>", [], False, 0, ln, line))
            continue
        code = line.split(";")[0].strip()
        if not code:
            continue
        parts = code.split(None, 1)
        mn = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        ins.append((mn, ops, in_asm, asm_id if in_asm else 0, ln, code))
    return ins


def wait_states(mn, ops):
    if mn == "s_nop":
        return int(ops[0], 0) + 1
    return 1


def is_valu(mn):
    return mn.startswith("v_") and not mn.startswith("v_nop")


def valu_writes_vregs(mn, ops):
    if not is_valu(mn) or not ops or mn.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
        return set()
    return _vregs(ops[0])


def sgpr_dest(mn, ops):
    """SGPRs (and "vcc") a VALU instruction writes"""
    if not is_valu(mn) or not ops:
        return set()
    if mn.startswith(("v_mad_u64_u32", "v_mad_i64_i32")) or re.match(r"v_(add|sub|subrev|addc|subb|subbrev)_co_", mn) or mn.startswith("v_div_scale"):
        return _sregs(ops[1]) if len(ops) > 1 else set()
    if mn.startswith("v_cmp"):
        if mn.endswith("_e32") or not (ops[0].startswith("s") or ops[0] == "vcc"):
            return {"vcc"}
        return _sregs(ops[0])
    if mn.startswith(("v_readlane", "v_readfirstlane")):
        return _sregs(ops[0])
    return set()


def sgpr_sources(mn, ops):
    """SGPRs (and "vcc") a VALU instruction reads: its source operands; e32 carry / mask consumers read VCC implicitly"""
    if not is_valu(mn):
        return set()
    n_dst = 2 if (mn.startswith(("v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale")) or re.match(r"v_(add|sub|subrev|addc|subb|subbrev)_co_", mn)) else 1
    src = set()
    for o in ops[n_dst:]:
        src |= _sregs(o)
    if mn in ("v_cndmask_b32_e32",) or re.match(r"v_(addc|subb|subbrev)_co_u32_e32", mn):
        src.add("vcc")
    return src


def carry_or_mask_read_of_vcc(mn, ops):
    if mn.startswith("v_cndmask_b32"):
        return mn.endswith("_e32") or (len(ops) >= 4 and ops[3] == "vcc")
    if re.match(r"v_(addc|subb|subbrev)_co_u32", mn):
        return mn.endswith("_e32") or (len(ops) >= 5 and ops[4] == "vcc")
    return False


def lint(ins):
    problems = []
    n = len(ins)
    for i, (mn, ops, in_asm, aid, ln, text) in enumerate(ins):
        # ---- R1: an asm store of > 64 bits and the next two wait states ----
        if in_asm and STORE_RE.match(mn):
            data = _vregs(ops[0] if mn.startswith("buffer") else ops[1])
            ws, j = 0, i + 1
            while j < n and ws < 2:
                m2, o2, *_rest = ins[j]
                if m2 == "<label>" or m2 in ("s_endpgm", "s_branch", "s_setpc_b64"):
                    break
                hit = valu_writes_vregs(m2, o2) & data
                if hit:
                    problems.append(f"R1 line {ln}: `{text}` — data register(s) v{sorted(hit)} rewritten by `{ins[j][5]}` (line {ins[j][4]}) after {ws} wait state(s)")
                    break
                ws += wait_states(m2, o2)
                j += 1
        # ---- R2 / R3: a VALU read of an SGPR pair / VCC and the two wait states before it ----
        if is_valu(mn):
            need = sgpr_sources(mn, ops)
            if need:
                ws, j = 0, i - 1
                while j >= 0 and ws < 2:
                    m2, o2, asm2, aid2, ln2, text2 = ins[j]
                    if m2 == "<label>":
                        break
                    wrote = sgpr_dest(m2, o2) & need
                    sg = {r for r in wrote if r != "vcc"}
                    if sg and (in_asm or asm2):
                        problems.append(f"R2 line {ln}: `{text}` reads s{sorted(sg)} written by `{text2}` (line {ln2}) {ws} wait state(s) earlier")
                        break
                    if "vcc" in wrote and in_asm and asm2 and aid == aid2 and carry_or_mask_read_of_vcc(mn, ops):
                        problems.append(f"R3 line {ln}: `{text}` takes VCC from `{text2}` (line {ln2}) {ws} wait state(s) earlier, inside one asm statement")
                        break
                    if wrote:
                        break  # the nearest producer decides
                    ws += wait_states(m2, o2)
                    j -= 1
    return problems


def _units():
    from ceno_amd import build

    return build.device_asm_files(build=True)


def test_every_unit_with_inline_assembly_is_covered():
    """the lint sees what ships: an assembly file per .hip unit, newer than its source, and the units known to hold `asm` statements have blocks"""
    units = _units()
    assert len(units) >= 13
    with_asm = 0
    for src, path in units:
        assert os.path.exists(path), f"{path} missing: python -m ceno_amd.build keeps it"
        assert os.path.getmtime(path) >= os.path.getmtime(src), f"{path} is older than {src}"
        if ";;#ASMSTART" in open(path, errors="replace").read():
            with_asm += 1
    assert with_asm >= 6   # sumcheck*, mle, poseidon2, basefold, ntt ...: everything that multiplies in the field


@pytest.mark.parametrize("unit", sorted(os.path.basename(p) for p in __import__("glob").glob(os.path.join(ROOT, "ceno_amd", "csrc", "*.hip"))))
def test_no_unpadded_hazard_around_inline_assembly(unit):
    from ceno_amd import build

    build.build_hip()
    path = build.device_asm_path(os.path.join(ROOT, "ceno_amd", "csrc", unit))
    problems = lint(parse(path))
    assert not problems, f"{unit}: {len(problems)} hazard(s), first: " + " | ".join(problems[:5])


def test_the_lint_catches_the_round5_bug_and_the_sgpr_rule(tmp_path):
    """what the rules are for, as text: the round-5 store without its wait states, the same store padded, a carry pair read too early"""
    bad_store = "\t;;#ASMSTART\n\tglobal_store_dwordx4 v[2:3], v[4:7], off sc0 sc1\n\t;;#ASMEND\n\tv_mov_b32_e32 v4, v9\n\ts_endpgm\n"
    ok_store = "\t;;#ASMSTART\n\tglobal_store_dwordx4 v[2:3], v[4:7], off sc0 sc1\n\ts_nop 1\n\t;;#ASMEND\n\tv_mov_b32_e32 v4, v9\n\ts_endpgm\n"
    late_store = "\t;;#ASMSTART\n\tglobal_store_dwordx4 v[2:3], v[4:7], off sc0 sc1\n\t;;#ASMEND\n\tv_mov_b32_e32 v8, v9\n\tv_mov_b32_e32 v10, v9\n\tv_mov_b32_e32 v4, v9\n"
    bad_sgpr = "\t;;#ASMSTART\n\tv_mad_u64_u32 v[0:1], s[4:5], v2, -1, v[0:1]\n\t;;#ASMEND\n\t;;#ASMSTART\n\tv_cndmask_b32 v3, 0, -1, s[4:5]\n\t;;#ASMEND\n"
    ok_sgpr = "\t;;#ASMSTART\n\tv_mad_u64_u32 v[0:1], s[4:5], v2, -1, v[0:1]\n\ts_nop 1\n\t;;#ASMEND\n\t;;#ASMSTART\n\tv_cndmask_b32 v3, 0, -1, s[4:5]\n\t;;#ASMEND\n"
    bad_vcc = "\t;;#ASMSTART\n\tv_add_co_u32 v0, vcc, v0, v1\n\tv_addc_co_u32 v2, vcc, 0, v2, vcc\n\t;;#ASMEND\n"
    ok_vcc = "\t;;#ASMSTART\n\tv_add_co_u32 v0, vcc, v0, v1\n\ts_nop 1\n\tv_addc_co_u32 v2, vcc, 0, v2, vcc\n\t;;#ASMEND\n"
    compiler_vcc = "\tv_add_co_u32_e32 v0, vcc, v0, v1\n\tv_addc_co_u32_e32 v2, vcc, 0, v2, vcc\n"   # hipcc's own pairs are its business
    for name, text, want in (("bad_store", bad_store, "R1"), ("ok_store", ok_store, None), ("late_store", late_store, None), ("bad_sgpr", bad_sgpr, "R2"),
                             ("ok_sgpr", ok_sgpr, None), ("bad_vcc", bad_vcc, "R3"), ("ok_vcc", ok_vcc, None), ("compiler_vcc", compiler_vcc, None)):
        p = tmp_path / (name + ".s")
        p.write_text(text)
        got = lint(parse(str(p)))
        if want is None:
            assert not got, (name, got)
        else:
            assert got and got[0].startswith(want), (name, got)
