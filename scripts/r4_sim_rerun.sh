#!/bin/bash
# the projections again (medians over the timed steps), same commands as scripts/r4_artifacts.sh
out=gpurun_out/r4art
for w in 2 4 8; do python bench.py --sim-world $w --steps 30 --warmup 5 > $out/sim${w}_n20.json 2> $out/sim${w}_n20.err; done
python bench.py --sim-world 8 --steps 30 --warmup 5 --n 4194304 --chunk-factor 6 > $out/sim8_n22_grid24.json 2> $out/sim8_n22.err
python bench.py --sim-world 8 --steps 8 --warmup 2 --n 16777216 --chunk-factor 10 > $out/sim8_n24_grid40.json 2> $out/sim8_n24.err
python bench.py --sim-world 8 --all-pairs --n 1048576 --steps 3 --warmup 1 > $out/ap_n20_sim8.json 2> $out/ap_n20_sim8.err
for f in sim2_n20 sim4_n20 sim8_n20 sim8_n22_grid24 sim8_n24_grid40 ap_n20_sim8; do python -c "
import json; d=json.load(open('$out/$f.json')); print('$f', round(d['modelled_step_ms'],4), round(d['modelled_step_ms_optimistic'],4), {k:max(v) for k,v in d['stage_ms_per_rank'].items()})"; done
