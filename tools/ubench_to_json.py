#!/usr/bin/env python3
"""tools/ubench/gathers_ub output -> profiles/latest_ubench.json: the measured ceiling of the texture addressers (wave-wide
gather instructions per second, chip-wide) that bench.py prices the final stage's gather stream against, with the sha256 of the
micro-benchmark's source so the number can be tied to the code that produced it.
usage: python tools/ubench_to_json.py gpurun_out/<round>/ubench_gathers.txt > profiles/latest_ubench.json"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = {}
for line in open(sys.argv[1]):
    m = re.match(r"\s*(.*?)\s+([\d.]+) ms\s+([\d.]+) G lane-gathers/s\s+([\d.]+) G wave-instr/s", line)
    if m:
        rows[m.group(1).strip()] = float(m.group(4))
src = open(os.path.join(ROOT, "tools", "ubench", "gathers.hip"), "rb").read()
pick = {k: rows[k] for k in ("coherent, 4 bytes per lane (dword)", "coherent, 8 bytes per lane (dwordx2)", "coherent, 16 bytes per lane (dwordx4)",
                             "coherent (1 line), lanes 0-31 active") if k in rows}
lines = {k: v for k, v in rows.items() if "distinct lines per instruction" in k}
peak = pick.get("coherent, 16 bytes per lane (dwordx4)")
print(json.dumps({
    "source": "tools/ubench/gathers.hip", "source_sha256_16": hashlib.sha256(src).hexdigest()[:16],
    "wave_gather_instr_per_s_peak": peak * 1e9 if peak else None,
    "cycles_per_instr_per_cu_at_2p4GHz": round(256 * 2.4e9 / (peak * 1e9), 2) if peak else None,
    "G_wave_instr_per_s": pick, "by_distinct_lines": lines,
    "note": "chip-wide rate of wave-wide gather instructions whose 64 lanes share <= 4 cache lines: the address-rate ceiling of the texture "
            "path (16-byte and 8-byte loads cost the same, masked lanes do not help)"}, indent=1))
