#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs into the small text tables committed under profiles/.

  python tools/rocpd_summary.py stats <results.db>      per-kernel count / total / avg / min / max (kernel-trace --stats)
  python tools/rocpd_summary.py pmc   <results.db>      per-kernel mean of every collected counter
"""
import sqlite3
import sys
from collections import defaultdict


def short(name: str) -> str:
    import re
    m = re.search(r"k_mlp_small<sn::Chain<(true|false), (\d), ([\d, ]+)>", name)
    if m:       # the small training perceptrons: direction and the chain's widths (backward chains list the widths reversed)
        dims = [d.strip() for d in m.group(3).split(",")][: int(m.group(2)) + 1]
        return f"k_mlp_small<{'bwd' if m.group(1) == 'true' else 'fwd'} {'-'.join(dims)}>"
    for key in ("k_mask16", "k_bin_scatter", "k_bin_accum", "k_bin_count", "k_bin_merge", "k_final_stage_sp", "k_prop_stage_sp", "k_final_stage", "k_prop_stage", "k_feat_stage", "k_linear_wgrad_mfma", "k_linear_wgrad_sum4", "k_pack_grid_mlp_f16", "k_pack_grid_mlp", "k_pack_mlp_wide", "k_mlp_wide",
                "k_grid_composite", "k_grid_forward", "k_grid_backward", "k_bwd_reduce", "k_bwd_keys",
                "k_composite", "k_generate_rays", "k_sample_pdf", "k_weights"):
        if key in name:
            tag = ""
            if key == "k_final_stage":
                m = re.search(r"Li32ELi(\d)ELi(n?\d+)E", name)
                if m:
                    tag = {"0": "<valu>", "1": "<mfma_f32>", "2": "<mfma_f16x3>"}[m.group(1)]
                if "ELb1EEE" in name:
                    tag += "<aux>"
            if key == "k_prop_stage_sp":
                m = re.search(r"Li16ELi(n?\d+)E", name)
                if m:
                    tag = {"3": "<prop0,K=3>", "2": "<prop1,K=2>"}.get(m.group(1), "<generic>")
            if key == "k_linear_wgrad_mfma":
                m = re.search(r"<(\d), (true|false), (true|false)>", name) or re.search(r"ILi(\d)ELb([01])ELb([01])E", name)
                if m:
                    tag = f"<{m.group(1)} column blocks{', vector loads' if m.group(2) in ('true', '1') else ''}{', flat' if m.group(3) in ('true', '1') else ''}>"
            if key == "k_prop_stage":
                m = re.search(r"Li16ELi(n?\d+)E", name)
                if m:
                    tag = {"3": "<prop0,K=3>", "2": "<prop1,K=2>"}.get(m.group(1), "<generic>")
            if "6__half" in name:
                tag += "<f16 tables>"
            return key + tag
    return name[:70]


def stats(db):
    con = sqlite3.connect(db)
    rows = con.execute("select name, (end - start) from kernels").fetchall()
    agg = defaultdict(list)
    for n, d in rows:
        agg[short(n)].append(d)
    tot = sum(sum(v) for v in agg.values())
    print(f"{'kernel':42s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k:42s} {len(v):6d} {sum(v) / 1e6:10.3f} {sum(v) / len(v) / 1e3:10.2f} {min(v) / 1e3:10.2f} {max(v) / 1e3:10.2f} {100 * sum(v) / tot:6.2f}")


def pmc(db):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    rows = con.execute("select * from counters_collection").fetchall()
    ki, ci, vi = cols.index("kernel_name"), cols.index("counter_name"), cols.index("value")
    agg = defaultdict(lambda: defaultdict(list))
    for r in rows:
        agg[short(r[ki])][r[ci]].append(r[vi])
    for k, cs in agg.items():
        if not any(t in k for t in ("k_final", "k_prop", "k_pack", "k_mlp_wide", "k_feat", "k_mask16", "k_bin_", "k_linear_wgrad", "k_grid_", "k_adam", "k_mlp_small",
                                    "k_ray_composite", "k_weights", "k_sample", "k_proposal_loss", "k_jitter", "k_zero16")):
            continue
        for c, v in sorted(cs.items()):
            print(f"{k:30s} {c:28s} dispatches={len(v):4d} mean={sum(v) / len(v):.6g} min={min(v):.6g} max={max(v):.6g}")


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2])
