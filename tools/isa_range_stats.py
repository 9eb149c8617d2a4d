#!/usr/bin/env python3
"""usage: isa_range_stats.py <kernel.s> <from-line> <to-line> [...more ranges]  -> instruction mix of those line ranges (1-based, inclusive)"""
import collections, sys
lines = open(sys.argv[1]).read().split("\n")
cnt = collections.Counter(); ops = collections.Counter()
for a, b in zip(sys.argv[2::2], sys.argv[3::2]):
    for l in lines[int(a) - 1:int(b)]:
        t = l.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"): continue
        op = t.split()[0]; ops[op] += 1
        if op.startswith("v_mfma"): cnt["mfma"] += 1
        elif op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): cnt["lane"] += 1
        elif op.startswith("v_"): cnt["valu"] += 1
        elif op.startswith(("global_", "flat_", "buffer_", "scratch_")): cnt["vmem"] += 1
        elif op.startswith("ds_"): cnt["lds"] += 1
        elif op.startswith("s_waitcnt"): cnt["waitcnt"] += 1
        elif op.startswith(("s_load", "s_buffer")): cnt["smem"] += 1
        elif op.startswith("s_nop"): cnt["nop"] += 1
        elif op.startswith("s_"): cnt["salu"] += 1
print(sum(cnt.values()), dict(cnt))
print(", ".join(f"{o}:{c}" for o, c in ops.most_common(60)))
