#!/usr/bin/env python3
"""profiles/<round>/pmc_<schedule>.txt -> the HBM-side bytes per launch of the dominant kernel (profiles/latest_traffic.json,
read by bench.py for roofline.traffic).  Follows MI355X_MICROARCH.md's HBM section: separate --pmc passes; on gfx950
FETCH_SIZE tallies 128-byte requests at 64 B, so read bytes = 128 B x TCC_EA0_RDREQ_128B + 64 B x (RDREQ - RDREQ_128B)
(= 2 x FETCH_SIZE KiB when every request is 128 B); WRITE_SIZE KiB as reported."""
import json
import os
import re
import sys


def parse(path):
    vals = {}
    for line in open(path):
        m = re.match(r"(\S+)\s+(\S+)\s+dispatches=\s*(\d+) mean=(\S+)", line)
        if m and m.group(1).startswith("k_final_stage"):
            vals[m.group(2)] = float(m.group(4))
    return vals


def source_fingerprint():
    """Same sha256 over the kernel sources as bench.py's: ties this PMC profile to the code it was taken from."""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "sanerf-hq_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h", ".inc")):
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def main():
    d = sys.argv[1]
    out = {}
    fp = source_fingerprint()
    for sch, tb in (("flat128", "f32"), ("ref", "f32"), ("flat128", "f16"), ("ref", "f16")):
        p = os.path.join(d, f"pmc_{sch}.txt" if tb == "f32" else f"pmc_{sch}_{tb}.txt")
        if not os.path.exists(p):
            continue
        v = parse(p)
        if "TCC_EA0_RDREQ_sum" not in v:
            continue
        rd, rd128 = v["TCC_EA0_RDREQ_sum"], v.get("TCC_EA0_RDREQ_128B_sum", 0.0)
        read_b = 128 * rd128 + 64 * (rd - rd128)
        write_b = v.get("WRITE_SIZE", 0.0) * 1024
        n_cu, xcd = 256, 8
        cyc = v.get("GRBM_GUI_ACTIVE", 0.0) / xcd                   # shader cycles of the launch (the counter sums the 8 XCDs)
        counters = {
            "MfmaUtil_pct": v.get("MfmaUtil"), "VALUBusy_pct": v.get("VALUBusy"),
            "TA_busy_pct": round(100.0 * v["TA_TA_BUSY_sum"] / (n_cu * cyc), 1) if cyc and "TA_TA_BUSY_sum" in v else None,
            "L2_hit_pct": round(100.0 * v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 1) if v.get("TCC_HIT_sum") and v.get("TCC_MISS_sum") is not None else None,
            "SQ_INSTS_VMEM_RD": v.get("SQ_INSTS_VMEM_RD"), "SQ_INSTS_VALU": v.get("SQ_INSTS_VALU"), "SQ_INSTS_LDS": v.get("SQ_INSTS_LDS"),
            "SQ_WAIT_ANY_pct_of_wave_cycles": round(100.0 * v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 1) if v.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in v else None,
            "shader_cycles_per_launch": round(cyc) if cyc else None,
            "note": "means per launch of the same command under rocprofv3 --pmc (separate passes); VALUBusy uses the gfx94x formula "
                    "(4 cycles per instruction), which overstates on gfx950"}
        out[f"{sch}_{tb}"] = {
            "counters": counters,
            "rays": 640000, "kernel": "k_final_stage", "source_fingerprint": fp,
            "hbm_read_bytes_per_launch": int(read_b), "hbm_write_bytes_per_launch": int(write_b),
            "hbm_bytes_per_launch": int(read_b + write_b),
            "raw": {"TCC_EA0_RDREQ_sum": rd, "TCC_EA0_RDREQ_128B_sum": rd128, "FETCH_SIZE_KiB": v.get("FETCH_SIZE"),
                    "WRITE_SIZE_KiB": v.get("WRITE_SIZE"), "TCC_HIT_sum": v.get("TCC_HIT_sum"), "TCC_MISS_sum": v.get("TCC_MISS_sum")},
            "method": f"separate rocprofv3 --pmc passes ({os.path.basename(d)}/{os.path.basename(p)}); read bytes = 128 B x TCC_EA0_RDREQ_128B + "
                      "64 B x the rest (= 2 x FETCH_SIZE KiB: the gfx950 correction of MI355X_MICROARCH.md, FETCH_SIZE tallies 128-B requests "
                      "at 64 B); WRITE_SIZE KiB as reported (<0.1% of the total)"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
