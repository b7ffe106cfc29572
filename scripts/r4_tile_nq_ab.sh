#!/bin/bash
# small shares (tile walk): 16 bodies per group (PSAMD_TILE_NQ=16) against 8
O=gpurun_out
for i in 1 2; do
  python bench.py --sim-world 8 --steps 30 --warmup 5 > $O/r4_tilenq8_$i.json 2>> $O/r4_tilenq.err
  PSAMD_TILE_NQ=16 python bench.py --sim-world 8 --steps 30 --warmup 5 > $O/r4_tilenq16_$i.json 2>> $O/r4_tilenq.err
done
python bench.py --sim-world 4 --steps 30 --warmup 5 > $O/r4_tilenq8_w4.json 2>> $O/r4_tilenq.err
PSAMD_TILE_NQ=16 python bench.py --sim-world 4 --steps 30 --warmup 5 > $O/r4_tilenq16_w4.json 2>> $O/r4_tilenq.err
PSAMD_TILE_NQ=16 python -m pytest tests/test_gpu_slab.py -m gpu -x -q 2>&1 | tail -2
python - <<'PY'
import json
for f in ("8_1","16_1","8_2","16_2","8_w4","16_w4"):
    try:
        d=json.load(open('gpurun_out/r4_tilenq%s.json'%f)); print(f, round(d['modelled_step_ms'],4), d['stage_ms_per_rank']['pairs'])
    except Exception as e: print(f,'ERR',e)
PY
