#!/bin/bash
# A/B of two builds of libglowk.so on one box, training sweep (variant built with __graft_entry__.build(tag=..., extra_flags=[...])):
#   bash scripts/ab_two_builds_train.sh audiosourcesep_amd/libglowk_<tag>.so [pytest]
# Every step runs under its own timeout and a failing step ends the script (a variant that hangs or faults must not be retried).
set -e -o pipefail
V=$PWD/$1
if [ "$2" = "pytest" ]; then
  GLOWK_LIB=$V timeout -k 10 300 python -m pytest tests/test_gpu_training.py -x -q > gpurun_out/ab_variant_test.log 2>&1; tail -1 gpurun_out/ab_variant_test.log
fi
for rep in 1 2; do
  echo product; timeout -k 10 120 python scripts/time_param_grad.py 2> gpurun_out/ab_variant.err | grep f16x3
  echo variant; GLOWK_LIB=$V timeout -k 10 120 python scripts/time_param_grad.py 2>> gpurun_out/ab_variant.err | grep f16x3
done
