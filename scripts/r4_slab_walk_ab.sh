#!/bin/bash
O=gpurun_out
for w in 2 4; do
  python bench.py --sim-world $w --steps 30 --warmup 5 > $O/r4_slabwalk_w${w}_list.json 2>> $O/r4_slabwalk.err
  PSAMD_SLAB_WALK0=1 python bench.py --sim-world $w --steps 30 --warmup 5 > $O/r4_slabwalk_w${w}_walk0.json 2>> $O/r4_slabwalk.err
done
python - <<'PY'
import json
for w in (2,4):
    for v in ("list","walk0"):
        try:
            d=json.load(open('gpurun_out/r4_slabwalk_w%d_%s.json'%(w,v))); print(w, v, round(d['modelled_step_ms'],4), d['stage_ms_per_rank']['pairs'])
        except Exception as e: print(w,v,'ERR',e)
PY
python -m pytest tests -m gpu -x -q > $O/r4_t6.log 2>&1; tail -3 $O/r4_t6.log
