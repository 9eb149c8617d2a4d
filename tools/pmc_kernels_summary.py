#!/usr/bin/env python3
"""One line per kernel from the counter passes of tools/gpu_pmc_kernels.sh (profiles/<round>/pmc_kernels_*.txt): cycles per launch (GRBM_GUI_ACTIVE / 8 XCDs),
MfmaUtil, VALUBusy, texture-address busy (TA_TA_BUSY_sum / 256 CUs / cycles), gather instructions, L1 accesses per gather instruction, L2 hit rate, FETCH / WRITE.
usage: python tools/pmc_kernels_summary.py profiles/r05 > profiles/r05/pmc_kernels_summary.txt"""
import os
import re
import sys

import hashlib
import json

d0 = sys.argv[1]
sets = (("pmc_kernels_mask_head.txt", ("k_mask16",)), ("pmc_kernels_c3_sam_head.txt", ("k_feat_stage",)),
        ("pmc_kernels_train_mask.txt", ("k_bin_refs", "k_bin_pull", "k_bin_scatter", "k_bin_accum", "k_linear_wgrad_mfma", "k_mlp_wide", "k_grid_forward")),
        ("pmc_kernels_train_rgb.txt", ("k_mlp_small<fwd 32-64-64-16>", "k_mlp_small<bwd 16-64-64-32>", "k_mlp_small<fwd 10-16-1>", "k_mlp_small<bwd 1-16-10>",
                                       "k_bin_refs", "k_bin_pull", "k_grid_forward", "k_linear_wgrad_mfma")),
        ("pmc_kernels_ref_f16.txt", ("k_prop_stage", "k_final_stage")), ("pmc_kernels_flat128_f16.txt", ("k_final_stage",)))
as_json = {}
print("workload      kernel                       cycles/launch  (ms at 2.3 GHz)  MfmaUtil VALUBusy TA-busy  gather-instr  L1-acc/instr  L2-hit  FETCH_SIZE  WRITE_SIZE")
for f, ks in sets:
    path = os.path.join(d0, f)
    if not os.path.exists(path):
        continue
    d = {}
    for ln in open(path):
        m = re.match(r"(.*?)\s+([A-Za-z_][A-Za-z0-9_]*)\s+dispatches=\s*(\d+) mean=(\S+)", ln)
        if m:
            d[(m.group(1).strip(), m.group(2))] = float(m.group(4))
    names = sorted({n for n, _ in d})
    for k in ks:
        full = [n for n in names if n == k or (k + "<") in n or ("::" + k + "(") in n]       # templated kernels print with their signature
        if not full:
            continue
        g = lambda c: d.get((full[0], c), float("nan"))   # noqa: E731
        act = g("GRBM_GUI_ACTIVE") / 8
        hit = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")) * 100
        as_json.setdefault(f[12:-4], {})[k] = {
            "shader_cycles_per_launch": act, "MfmaUtil_pct": g("MfmaUtil"), "VALUBusy_pct": g("VALUBusy"),
            "TA_busy_pct": g("TA_TA_BUSY_sum") / 256 / act * 100, "L2_hit_pct": hit, "gather_instructions": g("SQ_INSTS_VMEM_RD"),
            "fetch_bytes_gfx950_corrected": 2 * 1024 * g("FETCH_SIZE"), "write_bytes": 1024 * g("WRITE_SIZE")}
        print(f"{f[12:-4]:13s} {k:28s} {act:13.0f}  {act / 2.3e6:15.3f}  {g('MfmaUtil'):8.1f} {g('VALUBusy'):8.1f} {g('TA_TA_BUSY_sum') / 256 / act * 100:6.1f}%  {g('SQ_INSTS_VMEM_RD'):12.4g}  "
              f"{g('TCP_TOTAL_CACHE_ACCESSES_sum') / max(g('SQ_INSTS_VMEM_RD'), 1):12.1f}  {hit:5.1f}%  {g('FETCH_SIZE'):10.4g}  {g('WRITE_SIZE'):10.4g}")

if len(sys.argv) > 2:       # second argument: a JSON twin of the table for bench.py (`also.*.roofline.binding`), tied to the kernel sources by their hash
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "sanerf-hq_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith((".hip", ".h", ".inc")):
            h.update(open(os.path.join(csrc, fn), "rb").read())
    clean = {w: {k: {kk: (None if vv != vv else round(vv, 3)) for kk, vv in v.items()} for k, v in ks.items()} for w, ks in as_json.items()}
    json.dump({"source_fingerprint": h.hexdigest()[:16], "from": d0, "workloads": clean,
               "note": "means per launch under rocprofv3 --pmc (one counter group per pass, --kernel-trace only); fetch bytes = 2 x FETCH_SIZE KiB "
                       "(the gfx950 correction of MI355X_MICROARCH.md: 128-byte requests tallied at 64 B); VALUBusy by the gfx94x formula"},
              open(sys.argv[2], "w"), indent=1)
