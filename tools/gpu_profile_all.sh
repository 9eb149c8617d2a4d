#!/bin/bash
# usage (GPU box, repo root): tools/gpu_profile_all.sh [round-dir]   -> gpurun_out/<round-dir>/*  (copy what is judged into profiles/<round-dir>/)
# Regenerates every profile artefact of the round from the code as it is: rocprofv3 --kernel-trace --stats summaries of the
# bench line(s) and of the other configurations, the --pmc passes of the dominant kernel (separate passes, kernel-trace only),
# the HBM-side traffic json bench.py reads, the micro-benchmarks and the un-profiled bench lines.
R=${1:-r06}; out=$GRAFT_REPO_ROOT/gpurun_out/$R; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
stats() {   # name, command...
  local name=$1; shift
  rm -rf $out/_t; rocprofv3 --kernel-trace --stats -d $out/_t -o t -- "$@" > $out/${name}_under_rocprof.log 2>&1
  python $root/tools/rocpd_summary.py stats $out/_t/t_results.db > $out/kernel_stats_${name}.txt 2>&1
  rm -rf $out/_t
}
for sch in flat128 ref; do for tb in f32 f16; do
  stats ${sch}_${tb} python $root/bench.py --steps 20 --warmup 5 --schedule $sch --tables $tb --no-cpu-baseline --primary-only
  grep "^{" $out/${sch}_${tb}_under_rocprof.log | tail -1 > $out/bench_under_rocprof_${sch}_${tb}.json; rm -f $out/${sch}_${tb}_under_rocprof.log
done; done
stats c3_sam_head python $root/tools/c3_profile.py
stats train_rgb python $root/tools/train_profile.py rgb
stats train_rgb_noprop python $root/tools/train_profile.py rgb_noprop
stats train_mask python $root/tools/train_profile.py mask
stats mask_head python $root/tools/mask_profile.py mask
stats compact_live python $root/tools/mask_profile.py compact
rm -f $out/*_under_rocprof.log
# PMC passes (both schedules with fp32 tables, the headline schedule with fp16 tables as well)
for cfg in "flat128 f32" "ref f32" "flat128 f16"; do
  set -- $cfg; sch=$1; tb=$2
  pf=$out/pmc_$sch.txt; [ $tb = f16 ] && pf=$out/pmc_${sch}_f16.txt
  rm -f $pf
  while read -r c; do
    [ -z "$c" ] && continue
    rm -rf $out/_p; rocprofv3 --pmc $c --kernel-trace -d $out/_p -o pmc -- python $root/bench.py --steps 2 --warmup 1 --schedule $sch --tables $tb --no-cpu-baseline --primary-only > /dev/null 2>&1
    echo "== pass: $c" >> $pf
    python $root/tools/rocpd_summary.py pmc $out/_p/pmc_results.db | grep -v k_pack >> $pf 2>&1
    rm -rf $out/_p
  done <<LIST
MfmaUtil VALUBusy
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum
TCC_HIT_sum TCC_MISS_sum
FETCH_SIZE
WRITE_SIZE
LIST
done
python $root/tools/traffic_from_pmc.py $out > $out/latest_traffic.json
cp $out/latest_traffic.json $root/profiles/latest_traffic.json      # the un-profiled bench line at the end of this script reads it (same sources: fingerprint matches)
# micro-benchmarks
[ -x $root/tools/ubench/gathers_ub ] && timeout 300 $root/tools/ubench/gathers_ub > $out/ubench_gathers.txt 2>&1 && \
  python $root/tools/ubench_to_json.py $out/ubench_gathers.txt > $out/latest_ubench.json && cp $out/latest_ubench.json $root/profiles/latest_ubench.json
[ -x $root/tools/ubench/valu_rate_ub ] && timeout 120 $root/tools/ubench/valu_rate_ub > $out/ubench_valu_rate.txt 2>&1
python $root/tools/mlp_bench.py > $out/ubench_head_mlp.txt 2>&1
python $root/tools/wgrad_bench.py 2>/dev/null | tail -1 > $out/ubench_wgrad.json
python $root/tools/prop_sp_ab.py 2>/dev/null | tail -1 > $out/small_batch_kernels_ab.json
SN_AB_STEPS=128 python $root/tools/prop_sp_ab.py 2>/dev/null | tail -1 > $out/small_batch_kernels_ab_flat128.json
# robustness evidence: every fused path repeated on identical inputs (bit-equal?), random configurations against the oracle
python $root/tools/stress_determinism.py 30 2>&1 | grep -v amdgpu.ids > $out/stress_determinism.txt
python $root/tools/fuzz_parity.py 60 31 2>&1 | grep -v amdgpu.ids | tail -12 > $out/fuzz_parity_tail.txt
python $root/tools/fuzz_parity.py any 150 5 2>&1 | grep -v amdgpu.ids | grep -E "MISMATCH|mismatching|case 1[0-9]:" | cut -c1-300 > $out/fuzz_parity_any.txt
# un-profiled numbers
cd $root
python tools/bench_configs.py 2>/dev/null | tail -1 > $out/bench_configs.json
python tools/train_bench.py both 2>/dev/null | tail -1 > $out/train_bench.json
python tools/mlp_modes_ab.py 2>/dev/null | tail -1 > $out/mlp_modes_ab_final.json
python tools/feat_patch_ab.py 2>/dev/null | tail -1 > $out/feat_patch_ab_final.json
python tools/prop_pair_ab.py 2>/dev/null | tail -1 > $out/prop_pair_ab.json
python tools/band_small_ab.py 2>/dev/null | tail -1 > $out/band_small_ab.json
python tools/gemm_f32_bench.py 2>/dev/null | tail -1 > $out/gemm_f32_bench.json
python bench.py > $out/bench_default.json 2> $out/bench_default.err
ls -la $out
