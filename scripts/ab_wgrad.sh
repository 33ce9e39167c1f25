#!/bin/bash
# A/B of the interleaved round of k_wgrad_h3 (square shapes) against the plain order, same box: training step at 256 and 32 tiles
set -e
o=gpurun_out
for rep in 1 2; do
python bench.py --workload train --batch 256 --steps 5 --warmup 2 > $o/r3_wgi_train256_$rep.json 2> $o/r3_wgi.err
GLOWK_WGRAD_PLAIN=1 python bench.py --workload train --batch 256 --steps 5 --warmup 2 > $o/r3_wgp_train256_$rep.json 2>> $o/r3_wgi.err
python bench.py --workload train --batch 32 --steps 10 --warmup 3 > $o/r3_wgi_train32_$rep.json 2>> $o/r3_wgi.err
GLOWK_WGRAD_PLAIN=1 python bench.py --workload train --batch 32 --steps 10 --warmup 3 > $o/r3_wgp_train32_$rep.json 2>> $o/r3_wgi.err
done
for f in wgi_train256_1 wgp_train256_1 wgi_train256_2 wgp_train256_2 wgi_train32_1 wgp_train32_1 wgi_train32_2 wgp_train32_2; do python -c "
import json
d=json.loads(open('$o/r3_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
