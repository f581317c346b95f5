"""Minimal parser of the C ABI headers (include/ceno_hip.h, include/ceno_prover.h): prototypes, structs, enums, #defines.
Used by tools/gen_rust_sys.py (emits rust/ceno_hip-sys/src/lib.rs) and by tests/test_rust_shim.py (checks the Rust
`extern "C"` block against the headers prototype by prototype)."""
from __future__ import annotations

import os
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = [os.path.join(ROOT, "include", "ceno_hip.h"), os.path.join(ROOT, "include", "ceno_prover.h")]


@dataclass
class CType:
    """base type name + pointer levels, outermost last: `const T* const*` -> base T, ptrs [const-pointee, const-pointee]"""
    base: str
    ptr_const: List[bool] = field(default_factory=list)  # per pointer level: is the POINTEE const
    array: Optional[int] = None
    fn: Optional[Tuple["CType", List["CType"]]] = None  # function pointer: (return, args)

    @property
    def is_ptr(self) -> bool:
        return bool(self.ptr_const) or self.fn is not None

    def width_class(self, typedef_ptrs=()) -> str:
        """'ptr', 'i32', 'u32', 'u64', 'usize', 'f64', 'u8', 'u16', 'void' — what must agree across the boundary"""
        if self.is_ptr or self.base in typedef_ptrs:
            return "ptr"
        return {"int": "i32", "unsigned": "u32", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "double": "f64",
                "uint8_t": "u8", "uint16_t": "u16", "void": "void", "char": "i8", "int64_t": "i64", "int32_t": "i32",
                "c_ulong": "usize"}.get(self.base, "struct:" + self.base)


@dataclass
class Proto:
    name: str
    ret: CType
    args: List[Tuple[str, CType]]


@dataclass
class Struct:
    name: str
    fields: List[Tuple[str, CType]]
    opaque: bool = False


def _strip(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    out = []
    for line in text.split("\n"):
        if line.strip().startswith("#"):
            continue
        out.append(line)
    text = "\n".join(out)
    text = text.replace('extern "C" {', " ")
    return text


def parse_type(decl: str) -> Tuple[CType, str]:
    """`const uint64_t* const* name` -> (CType, name); the name may be empty"""
    decl = decl.strip()
    m = re.match(r"^(.*?)\(\s*\*\s*(\w*)\s*\)\s*\((.*)\)$", decl, flags=re.S)  # function pointer
    if m:
        ret, _ = parse_type(m.group(1))
        args = [parse_type(a)[0] for a in split_args(m.group(3))]
        return CType("fn", fn=(ret, args)), m.group(2)
    arr = None
    ma = re.search(r"\[(\d+)\]\s*$", decl)
    if ma:
        arr = int(ma.group(1))
        decl = decl[: ma.start()].strip()
    # multi-word integer types become one token BEFORE the declarator name is split off (`unsigned long long f(...)` has no name to pop)
    for spelled, canon in (("unsigned long long int", "uint64_t"), ("unsigned long long", "uint64_t"), ("long long int", "int64_t"),
                           ("long long", "int64_t"), ("unsigned long", "c_ulong"), ("unsigned int", "unsigned"), ("unsigned char", "uint8_t"),
                           ("unsigned short", "uint16_t")):
        decl = re.sub(r"\b" + spelled.replace(" ", r"\s+") + r"\b", canon, decl)
    toks = re.findall(r"\w+|\*", decl)
    name = ""
    if toks and toks[-1] != "*" and toks[-1] != "const" and len([t for t in toks if t not in ("const", "*", "struct", "unsigned")]) > 1:
        name = toks.pop()
    # base type = first non-const identifier(s)
    base_toks, i = [], 0
    pending_const = False
    while i < len(toks) and toks[i] != "*":
        if toks[i] == "const":
            pending_const = True
        elif toks[i] != "struct":
            base_toks.append(toks[i])
        i += 1
    base = " ".join(base_toks)
    if base == "unsigned int":
        base = "unsigned"
    if base == "unsigned long long":
        base = "uint64_t"
    ptrs = []
    while i < len(toks):
        if toks[i] == "*":
            ptrs.append(pending_const)
            pending_const = False
        elif toks[i] == "const":
            pending_const = True
        i += 1
    return CType(base, ptrs, arr), name


def split_args(s: str) -> List[str]:
    s = s.strip()
    if s in ("", "void"):
        return []
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "(":
            depth += 1
        if ch == ")":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    return [a.strip() for a in out]


def parse_headers(paths=HEADERS):
    protos: Dict[str, Proto] = {}
    structs: Dict[str, Struct] = {}
    enums: Dict[str, List[Tuple[str, int]]] = {}
    defines: Dict[str, int] = {}
    typedefs: Dict[str, str] = {}
    fn_typedefs: Dict[str, CType] = {}
    for path in paths:
        raw = open(path).read()
        for m in re.finditer(r"^#define\s+(CENO_\w+)\s+\(?([0-9a-fx]+)u?\s*(?:<<\s*(\d+))?\)?\s*$", raw, flags=re.M):
            v = int(m.group(2), 0)
            if m.group(3):
                v <<= int(m.group(3))
            defines[m.group(1)] = v
        text = _strip(raw)
        # enums
        for m in re.finditer(r"typedef\s+enum\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
            vals, nxt = [], 0
            for item in m.group(2).split(","):
                item = item.strip()
                if not item:
                    continue
                if "=" in item:
                    k, v = item.split("=")
                    nxt = int(v.strip(), 0)
                    vals.append((k.strip(), nxt))
                else:
                    vals.append((item, nxt))
                nxt += 1
            enums[m.group(3)] = vals
        text = re.sub(r"typedef\s+enum\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
        # struct definitions
        for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
            fields = []
            for stmt in m.group(2).split(";"):
                stmt = stmt.strip()
                if not stmt:
                    continue
                if "(" in stmt:  # function pointer member
                    t, n = parse_type(stmt)
                    fields.append((n, t))
                    continue
                parts = split_args(stmt)  # `uint32_t a, b[2]` -> declarators share the base type
                t0, n0 = parse_type(parts[0])
                fields.append((n0, t0))
                for extra in parts[1:]:
                    ex = extra.strip()
                    stars = len(ex) - len(ex.lstrip("* "))
                    t, n = parse_type(t0.base + " " + ex)
                    fields.append((n, t))
            structs[m.group(3)] = Struct(m.group(3), fields)
        text = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
        # opaque structs and plain typedefs
        for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", text):
            structs.setdefault(m.group(2), Struct(m.group(2), [], opaque=True))
        text = re.sub(r"typedef\s+struct\s+\w+\s+\w+\s*;", " ", text)
        for m in re.finditer(r"typedef\s+([^;(]+?)\s*\(\s*\*\s*(\w+)\s*\)\s*\(([^;]*)\)\s*;", text, flags=re.S):
            ret, _ = parse_type(m.group(1))
            fn_typedefs[m.group(2)] = CType("fn", fn=(ret, [parse_type(a)[0] for a in split_args(m.group(3))]))
        text = re.sub(r"typedef\s+[^;(]+?\(\s*\*\s*\w+\s*\)\s*\([^;]*\)\s*;", " ", text, flags=re.S)
        for m in re.finditer(r"typedef\s+([\w\s\*]+?)\s*(\w+)\s*;", text):
            typedefs[m.group(2)] = m.group(1).strip()
        text = re.sub(r"typedef\s+[\w\s\*]+?\w+\s*;", " ", text)
        # prototypes
        for m in re.finditer(r"([\w\s\*]+?)\b(ceno_\w+)\s*\(([^;{]*)\)\s*;", text, flags=re.S):
            ret, _ = parse_type(m.group(1))
            args = []
            for a in split_args(m.group(3)):
                t, n = parse_type(a)
                args.append((n, t))
            protos[m.group(2)] = Proto(m.group(2), ret, args)
    return protos, structs, enums, defines, typedefs, fn_typedefs
