# The round's final measurements, all from ONE gpurun call on the final build (run from the repository root on the GPU box):
#   bash scripts/final_run.sh <commit>
# bench lines -> gpurun_out/r02_*.json, rocprofv3 --kernel-trace --stats summaries -> gpurun_out/r02_*_kernel_stats.csv,
# PMC passes (counters in their own runs, --kernel-trace only) -> gpurun_out/r02_pmc_*/ ; copied into profiles/ afterwards.
set -e
export TMPDIR=/tmp
C=${1:-unknown}
O=gpurun_out
mkdir -p $O
python bench.py > $O/r02_bench.json 2> $O/r02_bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_prof -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-shapes > $O/r02_bench_under_rocprof.json 2> $O/r02_prof.err
cp $(find $O/r02_prof -name "*kernel_stats.csv" | head -1) $O/r02_kernel_stats.csv
echo "kernel trace done"
for p in f16x3 f32; do
  python bench.py --workload log_prob_grad --precision $p --no-cpu-baseline > $O/r02_grad_$p.json 2>> $O/r02_bench.err
  python bench.py --workload basis --batch 30 --steps 20 --warmup 3 --precision $p --no-cpu-baseline > $O/r02_basis_$p.json 2>> $O/r02_bench.err
  echo "$p done"
done
for p in f16x3 f32; do for b in 32 256; do python bench.py --workload train --precision $p --batch $b --no-cpu-baseline > $O/r02_train_${p}_b$b.json 2>> $O/r02_bench.err; done; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_prof_train -- python3 bench.py --workload train --precision f16x3 --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>> $O/r02_prof.err
cp $(find $O/r02_prof_train -name "*kernel_stats.csv" | head -1) $O/r02_train_kernel_stats.csv
echo "train done"
GLOWK_BENCH_REHEARSAL=1 python bench.py --gpus 2 --steps 3 --warmup 1 --batch 256 > $O/r02_bench_gpus2_rehearsal.json 2>> $O/r02_bench.err
echo "rehearsal done"
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-shapes"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/r02_pmc_fetch -- $B > /dev/null 2>> $O/r02_prof.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/r02_pmc_write -- $B > /dev/null 2>> $O/r02_prof.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/r02_pmc_mfma -- $B > /dev/null 2>> $O/r02_prof.err
python scripts/pmc_summary.py traffic $(find $O/r02_pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/r02_pmc_write -name "*counter_collection.csv" | head -1) $C $O/r02_roofline_traffic.json > /dev/null
python scripts/pmc_summary.py mfma $(find $O/r02_pmc_mfma -name "*counter_collection.csv" | head -1) $C $O/r02_mfma_utilisation.json > /dev/null
echo "pmc done"
