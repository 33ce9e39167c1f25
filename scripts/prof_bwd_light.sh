#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lpp1g -o g -- python3 $R/bench.py --workload log_prob_grad --steps 3 --warmup 1 > $R/gpurun_out/prof_lpp1g.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lpp1t -o t -- python3 $R/bench.py --workload train --batch 256 --steps 3 --warmup 1 > $R/gpurun_out/prof_lpp1t.log 2>&1
find $R/gpurun_out/prof_lpp1g $R/gpurun_out/prof_lpp1t -name '*kernel_stats.csv' | while read f; do echo $f; head -30 "$f" | cut -c1-160; done
find $R/gpurun_out/prof_lpp1g $R/gpurun_out/prof_lpp1t -name '*kernel_trace.csv' -delete
