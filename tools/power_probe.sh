#!/bin/bash
# usage (GPU box, repo root): tools/power_probe.sh <label> [VAR=value ...] -- [bench args]
# Runs the bench line long enough for the power management to settle (--steps 1500, ~10 s of back-to-back frames) and samples
# socket power, shader clock and temperature twice a second with rocm-smi meanwhile: which build is clock-limited by power?
label=$1; shift
envs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
[ "$1" = "--" ] && shift
out=gpurun_out/power_$label.txt
: > $out
( while true; do rocm-smi --showpower --showclocks --showtemp --csv 2>/dev/null | tail -n +2 | head -1 >> $out; sleep 0.5; done ) &
spid=$!
line=$(env "${envs[@]}" python bench.py --no-cpu-baseline --primary-only --steps 1500 --warmup 20 "$@" 2>/dev/null | grep "^{")
kill $spid
python - "$label" "$out" <<PY
import json, sys, re, statistics as st
d = json.loads('''$line'''); r = d['roofline']
rows = [l.strip().split(',') for l in open(sys.argv[2]) if l.startswith('card')]
pw = [float(x[-1]) for x in rows]; clk = [float(re.sub(r'[^0-9.]', '', x[7])) for x in rows]; tj = [float(x[1]) for x in rows]
busy = [i for i, p in enumerate(pw) if p > 0.8 * max(pw)]
print('%-14s %-8s %7.2f Mrays/s %7.3f ms | final %.3f ms | in-kernel clk %6.0f MHz | smi: %4.0f W  sclk %4.0f MHz  Tj %2.0f C  (%d samples)' % (
    sys.argv[1], d['config']['schedule'], d['value'] / 1e6, d['ms_per_step'], r['avg_kernel_ms'], r['shader_clock_mhz'],
    st.median(pw[i] for i in busy), st.median(clk[i] for i in busy), max(tj), len(busy)))
PY
