#!/usr/bin/env python3
"""Fold compile-time knobs whose alternative is a recorded loser into the source (a small `unifdef`): for the given MACRO=value pairs every
`#if / #elif / #else / #endif` group whose conditions mention ONLY those macros (and literals) is resolved to its taken branch, the
`#ifndef MACRO / #define MACRO v / #endif` default blocks are dropped and remaining uses of the macro become the literal.  Everything else is left
untouched.  usage: python tools/fold_knobs.py file MACRO=value [MACRO=value ...]   (rewrites the file; check with the preprocessor diff it prints)"""
import re
import sys


def main():
    path = sys.argv[1]
    vals = dict(a.split("=") for a in sys.argv[2:])
    src = open(path).read().split("\n")
    tok = re.compile(r"[A-Za-z_][A-Za-z0-9_]*")

    def resolvable(expr):
        e = re.sub(r"defined\s*\(\s*([A-Za-z_][A-Za-z0-9_]*)\s*\)", lambda m: "1" if m.group(1) in vals else "defined(" + m.group(1) + ")", expr)
        names = set(tok.findall(re.sub(r"//.*", "", e)))
        return all(n in vals for n in names), e

    def evaluate(e):
        e = re.sub(r"//.*", "", e)
        e = tok.sub(lambda m: vals[m.group(0)], e)
        e = e.replace("&&", " and ").replace("||", " or ")
        e = re.sub(r"!(?!=)", " not ", e)
        return bool(eval(e))
    out = []
    i = 0
    # stack entries: dict(resolved=bool, taken=bool (a branch was already taken), emitting=bool)
    stack = []
    emitting = lambda: all(s["emit"] for s in stack)   # noqa: E731
    n = len(src)
    while i < n:
        ln = src[i]
        st = ln.strip()
        m_if = re.match(r"#\s*if\s+(.*)", st)
        m_ifndef = re.match(r"#\s*ifndef\s+([A-Za-z_][A-Za-z0-9_]*)", st)
        m_ifdef = re.match(r"#\s*ifdef\s+([A-Za-z_][A-Za-z0-9_]*)", st)
        if m_ifndef and m_ifndef.group(1) in vals and emitting():
            # the default block: #ifndef M / #define M v [// comment possibly continued on the #endif line] / #endif
            j = i + 1
            while j < n and not re.match(r"#\s*endif", src[j].strip()):
                j += 1
            i = j + 1
            continue
        if m_if or m_ifdef or m_ifndef:
            if m_if:
                ok, e = resolvable(m_if.group(1))
            else:
                ok, e = False, ""
            if ok and emitting():
                v = evaluate(e)
                stack.append({"res": True, "taken": v, "emit": v})
            else:
                stack.append({"res": False, "taken": False, "emit": True})
                if emitting():
                    out.append(ln)
            i += 1
            continue
        m_elif = re.match(r"#\s*elif\s+(.*)", st)
        if m_elif and stack:
            top = stack[-1]
            if top["res"]:
                ok, e = resolvable(m_elif.group(1))
                if not ok:
                    raise SystemExit(f"{path}:{i + 1}: #elif mixes folded and other macros")
                v = (not top["taken"]) and evaluate(e)
                top["emit"] = v
                top["taken"] = top["taken"] or v
            elif emitting():
                out.append(ln)
            i += 1
            continue
        if re.match(r"#\s*else\b", st) and stack:
            top = stack[-1]
            if top["res"]:
                top["emit"] = not top["taken"]
                top["taken"] = True
            elif emitting():
                out.append(ln)
            i += 1
            continue
        if re.match(r"#\s*endif\b", st) and stack:
            top = stack.pop()
            if not top["res"] and emitting():
                out.append(ln)
            i += 1
            continue
        if emitting():
            out.append(tok.sub(lambda m: vals.get(m.group(0), m.group(0)), ln) if any(k in ln for k in vals) and not st.startswith("//") else ln)
        i += 1
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
