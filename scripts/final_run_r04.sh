# Round 4: the round's final measurements, all from ONE gpurun call on the final build (run from the repository root on the GPU box):
#   bash scripts/final_run.sh <commit>
# bench lines -> gpurun_out/r04_*.json, rocprofv3 --kernel-trace --stats summaries -> gpurun_out/r04_*_kernel_stats.csv,
# PMC passes (counters in their own runs, --kernel-trace only) -> summarised into gpurun_out/r04_*.json ; copied into profiles/ afterwards.
set -e
export TMPDIR=/tmp
C=${1:-unknown}
O=gpurun_out
mkdir -p $O
python bench.py > $O/r04_bench.json 2> $O/r04_bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_prof -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-shapes --no-secondary > $O/r04_bench_under_rocprof.json 2> $O/r04_prof.err
cp $(find $O/r04_prof -name "*kernel_stats.csv" | head -1) $O/r04_kernel_stats.csv; rm -rf $O/r04_prof
echo "kernel trace done"
for p in f16x3 f32; do
  python bench.py --workload log_prob_grad --precision $p --no-cpu-baseline > $O/r04_grad_$p.json 2>> $O/r04_bench.err
  python bench.py --workload log_prob_grad --precision $p --batch 30 --steps 20 --warmup 3 --no-cpu-baseline > $O/r04_grad30_$p.json 2>> $O/r04_bench.err
  for b in 32 256; do python bench.py --workload train --precision $p --batch $b --no-cpu-baseline > $O/r04_train_${p}_b$b.json 2>> $O/r04_bench.err; done
  echo "$p done"
done
python bench.py --workload basis --steps 20 --warmup 3 > $O/r04_basis_f16x3.json 2>> $O/r04_bench.err
python bench.py --workload basis --basis-crop 64 --steps 20 --warmup 3 > $O/r04_basis64_f16x3.json 2>> $O/r04_bench.err
python bench.py --workload basis --basis-levels 10 --basis-sigma1 1.0 --basis-T 100 --basis-K 32 --batch 30 --steps 20 --warmup 2 > $O/r04_basis_full_ladder_f16x3.json 2>> $O/r04_bench.err
echo "basis done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_prof_train -- python3 bench.py --workload train --precision f16x3 --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>> $O/r04_prof.err
cp $(find $O/r04_prof_train -name "*kernel_stats.csv" | head -1) $O/r04_train_kernel_stats.csv; rm -rf $O/r04_prof_train
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_prof_train256 -- python3 bench.py --workload train --precision f16x3 --batch 256 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>> $O/r04_prof.err
cp $(find $O/r04_prof_train256 -name "*kernel_stats.csv" | head -1) $O/r04_train256_kernel_stats.csv; rm -rf $O/r04_prof_train256
GLOWK_AB_N=30 GLOWK_GRAD=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_prof_small -- python3 scripts/small_batch.py > $O/r04_small_batch.log 2>> $O/r04_prof.err
cp $(find $O/r04_prof_small -name "*kernel_stats.csv" | head -1) $O/r04_grad30_kernel_stats.csv; rm -rf $O/r04_prof_small
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_prof_basis -- python3 bench.py --workload basis --basis-levels 3 --basis-train-steps 30 --basis-T 20 --steps 10 --warmup 2 > /dev/null 2>> $O/r04_prof.err
cp $(find $O/r04_prof_basis -name "*kernel_stats.csv" | head -1) $O/r04_basis_kernel_stats.csv; rm -rf $O/r04_prof_basis
echo "traces done"
GLOWK_BENCH_REHEARSAL=1 python bench.py --gpus 2 --steps 3 --warmup 1 --batch 256 --no-secondary > $O/r04_bench_gpus2_rehearsal.json 2>> $O/r04_bench.err
GLOWK_BENCH_FORCE_DIST=1 python bench.py --steps 3 --warmup 1 --batch 256 --no-secondary --no-cpu-baseline --no-other-shapes > $O/r04_bench_rccl_1rank.json 2>> $O/r04_bench.err
echo "rehearsal done"
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-shapes --no-secondary"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/r04_pmc_fetch -- $B > /dev/null 2>> $O/r04_prof.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/r04_pmc_write -- $B > /dev/null 2>> $O/r04_prof.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/r04_pmc_mfma -- $B > /dev/null 2>> $O/r04_prof.err
python scripts/pmc_summary.py traffic $(find $O/r04_pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/r04_pmc_write -name "*counter_collection.csv" | head -1) $C $O/r04_roofline_traffic.json > /dev/null
python scripts/pmc_summary.py mfma $(find $O/r04_pmc_mfma -name "*counter_collection.csv" | head -1) $C $O/r04_mfma_utilisation.json > /dev/null
rm -rf $O/r04_pmc_fetch $O/r04_pmc_write $O/r04_pmc_mfma
# the one-workgroup-per-CU kernel of round 3 on the same build and box (GLOWK_CO_OFF=1), for the MFMA-busy comparison
export GLOWK_CO_OFF=1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/r04_pmc_mfma_u -- $B > /dev/null 2>> $O/r04_prof.err
unset GLOWK_CO_OFF
python scripts/pmc_summary.py mfma $(find $O/r04_pmc_mfma_u -name "*counter_collection.csv" | head -1) $C $O/r04_mfma_utilisation_one_per_cu.json > /dev/null
rm -rf $O/r04_pmc_mfma_u
echo "pmc done"
