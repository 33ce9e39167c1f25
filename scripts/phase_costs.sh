#!/bin/bash
# Per-kernel times of the 30-tile gradient chain for the product build and the diagnostic builds (-DGLOWK_EXP_NOX / NOBARRIER, built with __graft_entry__.build(tag=..., extra_flags=...):
# wrong results on purpose), one rocprofv3 kernel trace each: what does each phase kind cost a pass-workgroup of a small grid?
export TMPDIR=/tmp
for v in "" _nox _nobarrier; do
  lib=$PWD/audiosourcesep_amd/libglowk$v.so
  [ -f $lib ] || continue
  rm -rf gpurun_out/pc$v
  GLOWK_LIB=$lib GLOWK_IGNORE_RANGE=1 GLOWK_AB_N=30 GLOWK_GRAD=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pc$v -o pc -- python3 scripts/small_batch.py > gpurun_out/pc$v.log 2>&1
  echo "== build '$v': $(grep 'ms per call' gpurun_out/pc$v.log)"
  python scripts/kstats.py $(find gpurun_out/pc$v -name "*kernel_stats.csv" | head -1) 6
done
