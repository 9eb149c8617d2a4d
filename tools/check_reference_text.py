#!/usr/bin/env python3
"""Pins the oracle's encoder arithmetic to the TEXT of the reference's CUDA sources (which cannot be compiled or run here).

Runs only where /root/reference exists (the build container); nothing of the reference is copied: its text is read,
transformed mechanically and compared.

 1. Hash-grid index.  The bodies of `fast_hash` and `get_grid_index` (gridencoder/src/gridencoder.cu:45-79) are transliterated
    token by token into Python (C for-loops -> while-loops, uint32_t -> a wrapping 32-bit integer class, template arguments ->
    globals) and EXECUTED; the oracle's row function (oracle.c:grid_row, exported as orc_grid_row) must return the same index on
    random vertices for D = 2..5, hashed and tiled grids, power-of-two and odd table sizes, resolutions up to 2^20 (products wrap).
 2. Spherical harmonics, forward.  Every `outputs[k] = ...;` of `write_sh` (shencoder/src/shencoder.cu:50-120) is parsed into a
    polynomial (sympy) and compared coefficient by coefficient with (a) the oracle's forward polynomials (oracle.c:ORC_SH_BODY)
    and (b) the device basis (csrc/sh_basis.inc, generated independently by tools/gen_sh.py; fp32 literals: 1e-7 relative).
    (The derivatives are covered the same way by tools/gen_oracle_sh_grad.py.)

  python tools/check_reference_text.py        # prints what was verified; exit code 0 = all equal
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


# ------------------------------------------------------------------------------------------------
# 1. grid index
# ------------------------------------------------------------------------------------------------
class U32:
    """uint32_t with C's wrap-around semantics."""
    __slots__ = ("v",)

    def __init__(self, v=0):
        self.v = int(v.v if isinstance(v, U32) else v) & 0xFFFFFFFF

    def _o(self, o):
        return int(o.v if isinstance(o, U32) else o) & 0xFFFFFFFF

    def __add__(self, o): return U32(self.v + self._o(o))
    __radd__ = __add__
    def __mul__(self, o): return U32(self.v * self._o(o))
    __rmul__ = __mul__
    def __xor__(self, o): return U32(self.v ^ self._o(o))
    def __mod__(self, o): return U32(self.v % self._o(o))
    def __lt__(self, o): return self.v < self._o(o)
    def __le__(self, o): return self.v <= self._o(o)
    def __gt__(self, o): return self.v > self._o(o)
    def __ge__(self, o): return self.v >= self._o(o)
    def __eq__(self, o): return self.v == self._o(o)
    def __index__(self): return self.v
    def __int__(self): return self.v
    def __hash__(self): return hash(self.v)


def c_function(src: str, name: str):
    """(parameter names, body text) of the C function `name` in src."""
    m = re.search(r"\b" + re.escape(name) + r"\s*\(([^)]*)\)\s*\{", src)
    assert m, name
    params = [re.sub(r"\[.*?\]", "", p).split()[-1] for p in m.group(1).split(",")]
    depth, i = 1, m.end()
    while depth:
        depth += {"{": 1, "}": -1}.get(src[i], 0)
        i += 1
    return params, src[m.end():i - 1]


def transliterate(body: str, indent: str = "    ") -> str:
    """C statements -> Python statements, purely syntactic."""
    body = re.sub(r"//[^\n]*", "", body)
    body = re.sub(r"#pragma[^\n]*", "", body)
    out = []
    pos = 0

    def expr(e):
        e = re.sub(r"(\d+)u\b", r"U32(\1)", e)                   # 2654435761u
        e = re.sub(r"\b(\w+)\s*<[^<>]*>\s*\(", r"\1(", e)          # template arguments of a call
        return e.replace("&&", " and ").replace("||", " or ")

    def block(text, ind):
        nonlocal out
        i = 0
        while i < len(text):
            rest = text[i:].lstrip()
            i = len(text) - len(rest)
            if not rest:
                break
            m = re.match(r"for\s*\(([^;]*);([^;]*);([^)]*)\)\s*\{", rest)
            if m:
                end = match_brace(rest, m.end() - 1)
                stmt(m.group(1), ind)
                out.append(f"{ind}while {expr(m.group(2).strip())}:")
                block(rest[m.end():end], ind + indent)
                stmt(m.group(3), ind + indent)
                i += end + 1
                continue
            m = re.match(r"if\s*\((.*?)\)\s*\{", rest, re.S)
            if m:
                end = match_brace(rest, m.end() - 1)
                out.append(f"{ind}if {expr(m.group(1).strip())}:")
                block(rest[m.end():end], ind + indent)
                i += end + 1
                continue
            semi = rest.index(";")
            stmt(rest[:semi], ind)
            i += semi + 1

    def match_brace(text, open_at):
        depth, k = 0, open_at
        while True:
            depth += {"{": 1, "}": -1}.get(text[k], 0)
            if depth == 0:
                return k
            k += 1

    def stmt(s, ind):
        s = " ".join(s.split())
        if not s:
            return
        m = re.match(r"(?:constexpr\s+)?uint32_t\s+(\w+)\[\d+\]\s*=\s*\{(.*)\}$", s)
        if m:
            out.append(f"{ind}{m.group(1)} = [{expr(m.group(2))}]")
            return
        s = re.sub(r"^(?:const\s+)?uint32_t\s+", "", s)             # declarations: `uint32_t a = 1` / `uint32_t a = 1, b = 0`
        m = re.match(r"(\+\+|--)?(\w+)(\+\+|--)?$", s)
        if m and (m.group(1) or m.group(3)):
            out.append(f"{ind}{m.group(2)} = U32({m.group(2)}) + {1 if '+' in (m.group(1) or m.group(3)) else -1}")
            return
        m = re.match(r"return\s+(.*)$", s)
        if m:
            out.append(f"{ind}return U32({expr(m.group(1))})")
            return
        m = re.match(r"(\w+)\s*(\^|\+|\*)?=\s*(.*)$", s)
        assert m, s
        lhs, op, rhs = m.group(1), m.group(2), expr(m.group(3))
        out.append(f"{ind}{lhs} = U32({lhs}) {op} ({rhs})" if op else f"{ind}{lhs} = U32({rhs})")

    block(body, indent)
    return "\n".join(out)


def check_grid_index(trials: int = 200000) -> int:
    import numpy as np
    import oracle as orc
    src = open(os.path.join(REF, "gridencoder/src/gridencoder.cu")).read()
    head = src[:src.index("kernel_grid(")]                      # the helper functions precede the kernels (lines 45-79)
    code = []
    for name in ("fast_hash", "get_grid_index"):
        params, body = c_function(head, name)
        code.append(f"def {name}({', '.join(params)}):\n{transliterate(body)}\n")
    text = "\n".join(code)
    assert "2654435761" in text and "805459861" in text and "% hashmap_size" in text and "stride <= hashmap_size" in text, text
    env = {"U32": U32}
    exec(text, env)
    rng = np.random.default_rng(7)
    n = 0
    for D in (2, 3, 4, 5):
        env["D"] = D
        for _ in range(trials // 4):
            gridtype = int(rng.integers(0, 2))
            size = int(rng.choice([1 << int(rng.integers(3, 25)), int(rng.integers(8, 1 << 22)) | 1, 8 * int(rng.integers(1, 1 << 18))]))
            res = int(rng.choice([int(rng.integers(2, 64)), int(rng.integers(64, 1 << 20))]))
            pg = [int(v) for v in rng.integers(0, res, size=D)]
            C = int(rng.choice([1, 2, 4, 8]))
            ch = int(rng.integers(0, C))
            env["C"] = C
            want = int(env["get_grid_index"](gridtype, ch, size, res, [U32(v) for v in pg]))
            got = (orc.grid_row(gridtype, size, res, pg) * C + ch) & 0xFFFFFFFF
            assert want == got, (D, gridtype, size, res, pg, C, ch, want, got)
            n += 1
    print(f"grid index: the oracle's grid_row equals the executed text of gridencoder.cu:45-79 on {n} random vertices "
          f"(D = 2..5, hash + tiled, wrap-around products)")
    return n


# ------------------------------------------------------------------------------------------------
# 2. SH forward
# ------------------------------------------------------------------------------------------------
def check_sh_forward():
    import sympy as sp
    import gen_oracle_sh_grad as g
    x, y, z = g.x, g.y, g.z
    names = dict(g.NAMES, xyz=x * y * z, x3=x**3, y3=y**3, z3=z**3, x5=x**5, y5=y**5, z5=z**5, x7=x**7, y7=y**7, z7=z**7)

    def poly(expr):
        e = re.sub(r"(\d)f\b", r"\1", expr)
        return g.monomials(sp.sympify(e, locals=names, rational=False))

    src = open(os.path.join(REF, "shencoder/src/shencoder.cu")).read()
    block = src[src.index("auto write_sh = "):src.index("write_sh();")]
    ref = {int(m.group(1)): poly(m.group(2)) for m in re.finditer(r"outputs\[(\d+)\] = (.*?);", block)}
    assert sorted(ref) == list(range(64)), sorted(ref)
    orc_polys = [g.monomials(p) for p in g.oracle_polys()]
    dev_src = open(os.path.join(ROOT, "sanerf-hq_amd/csrc/sh_basis.inc")).read()
    dev_block = dev_src[dev_src.index("#define SN_SH_VALUES(o)"):dev_src.index("#define SN_SH_DX(o)")]
    dev = {int(m.group(1)): poly(m.group(2)) for m in re.finditer(r"o\[(\d+)\] = (.*?);", dev_block)}
    assert sorted(dev) == list(range(64)), sorted(dev)
    worst = {"oracle": 0.0, "device": 0.0}
    for k in range(64):
        for label, got, tol in (("oracle", orc_polys[k], 1e-12), ("device", dev[k], 2e-7)):
            for key in set(ref[k]) | set(got):
                a, b = ref[k].get(key, 0.0), got.get(key, 0.0)
                err = abs(a - b) / max(abs(a), abs(b), 1e-300)
                worst[label] = max(worst[label], err)
                assert err < tol, (label, k, key, a, b)
    print(f"SH forward: 64 polynomials of shencoder.cu:50-120 equal the oracle's (worst relative coefficient difference "
          f"{worst['oracle']:.1e}) and the device basis sh_basis.inc ({worst['device']:.1e}, fp32 literals)")
    return worst


def main():
    if not os.path.isdir(REF):
        print("reference not present: nothing checked")
        return 0
    check_grid_index()
    check_sh_forward()
    return 0


if __name__ == "__main__":
    sys.exit(main())
