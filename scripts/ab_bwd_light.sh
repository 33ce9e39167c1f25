#!/bin/bash
# A/B of the one-lane-per-pixel form of k_bwd_light (large grids) against the four-lane form: log_prob_grad at 1024 tiles, training at 256
set -e
o=gpurun_out
python bench.py --workload log_prob_grad --steps 5 --warmup 2 > $o/r3_lpp1_grad.json 2> $o/r3_lpp1.err
GLOWK_BWD_LIGHT_4=1 python bench.py --workload log_prob_grad --steps 5 --warmup 2 > $o/r3_lpp4_grad.json 2>> $o/r3_lpp1.err
python bench.py --workload train --batch 256 --steps 5 --warmup 2 > $o/r3_lpp1_train256.json 2>> $o/r3_lpp1.err
GLOWK_BWD_LIGHT_4=1 python bench.py --workload train --batch 256 --steps 5 --warmup 2 > $o/r3_lpp4_train256.json 2>> $o/r3_lpp1.err
for f in lpp1_grad lpp4_grad lpp1_train256 lpp4_train256; do python -c "
import json
d=json.loads(open('$o/r3_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
