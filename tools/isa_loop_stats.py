#!/usr/bin/env python3
"""usage: isa_loop_stats.py <file.s> <mangled-kernel-substring>  -> instruction mix of the kernel's hottest (largest) loop.
Loop = the largest backward-branch span inside the function."""
import collections
import re
import sys

src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith(key.split("$")[-1]) or (l.startswith("_Z") and key in l and ":" in l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end + 1]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
best = None
for i, l in enumerate(body):
    m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        span = (labels[m.group(1)], i)
        if best is None or span[1] - span[0] > best[1] - best[0]:
            best = span
lo, hi = best
cnt = collections.Counter()
ops = collections.Counter()
for l in body[lo:hi + 1]:
    t = l.strip()
    if not t or t.startswith((";", ".")) or t.endswith(":"):
        continue
    op = t.split()[0]
    ops[op] += 1
    if op.startswith("v_mfma"): cnt["mfma"] += 1
    elif op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): cnt["valu_lane(spill)"] += 1
    elif op.startswith("v_"): cnt["valu"] += 1
    elif op.startswith(("global_", "flat_", "buffer_", "scratch_")): cnt["vmem"] += 1
    elif op.startswith("ds_"): cnt["lds"] += 1
    elif op.startswith("s_waitcnt"): cnt["waitcnt"] += 1
    elif op.startswith(("s_load", "s_buffer")): cnt["smem"] += 1
    elif op.startswith("s_nop"): cnt["nop"] += 1
    elif op.startswith("s_"): cnt["salu"] += 1
    else: cnt["other"] += 1
print("loop lines", lo, hi, "instructions", sum(cnt.values()))
print(dict(cnt))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 45
for op, c in ops.most_common(n):
    print(f"  {op:28s} {c}")
