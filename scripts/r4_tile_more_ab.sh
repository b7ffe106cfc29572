#!/bin/bash
# (a) tolerance mode, small share: 16 bodies per group in the tile walk?  (b) N = 2^22 on eight ranks: the tile walk (2048 waves) instead of the scalar walk (4096)?
O=gpurun_out
python bench.py --sim-world 8 --fast-math --steps 30 --warmup 5 > $O/r4_tm_fast8.json 2>> $O/r4_tm.err
PSAMD_TILE_NQ=16 python bench.py --sim-world 8 --fast-math --steps 30 --warmup 5 > $O/r4_tm_fast16.json 2>> $O/r4_tm.err
python bench.py --sim-world 8 --steps 20 --warmup 5 --n 4194304 --chunk-factor 6 > $O/r4_tm_n22_scalar.json 2>> $O/r4_tm.err
PSAMD_TILE=1 PSAMD_WAVES=2048 python bench.py --sim-world 8 --steps 20 --warmup 5 --n 4194304 --chunk-factor 6 > $O/r4_tm_n22_tile.json 2>> $O/r4_tm.err
python bench.py --sim-world 8 --steps 30 --warmup 5 > $O/r4_tm_n20.json 2>> $O/r4_tm.err
python - <<'PY'
import json
for f in ("fast8","fast16","n22_scalar","n22_tile","n20"):
    try:
        d=json.load(open('gpurun_out/r4_tm_%s.json'%f)); print(f, round(d['modelled_step_ms'],4), d['stage_ms_per_rank']['pairs'])
    except Exception as e: print(f,'ERR',e)
PY
