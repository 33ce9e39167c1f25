set -e
export TMPDIR=/tmp
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
echo bench done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_prof -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final_bench_prof.json 2> gpurun_out/final_prof.err
echo prof done
for p in f16x3 f32; do
  python bench.py --workload log_prob_grad --precision $p --no-cpu-baseline > gpurun_out/final_grad_$p.json 2>> gpurun_out/final_bench.err
  python bench.py --workload basis --batch 30 --precision $p --no-cpu-baseline > gpurun_out/final_basis_$p.json 2>> gpurun_out/final_bench.err
  echo $p done
done
