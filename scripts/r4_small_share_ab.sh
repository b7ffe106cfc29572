#!/bin/bash
O=gpurun_out
run() { name=$1; shift; env "$@" python bench.py --sim-world 8 --steps 30 --warmup 5 > $O/r4_small_$name.json 2>> $O/r4_small.err; }
run default X=1
run scalar2048 PSAMD_TILE=0 PSAMD_WAVES=2048
run scalar4096 PSAMD_TILE=0 PSAMD_WAVES=4096
run scalar7168 PSAMD_TILE=0 PSAMD_WAVES=7168
run scalar4096_nopace PSAMD_TILE=0 PSAMD_WAVES=4096 PSAMD_PACE=0
python - <<'PY'
import json
for f in ("default","scalar2048","scalar4096","scalar7168","scalar4096_nopace"):
    try:
        d=json.load(open('gpurun_out/r4_small_%s.json'%f)); print(f, round(d['modelled_step_ms'],4), d['stage_ms_per_rank']['pairs'])
    except Exception as e: print(f,'ERR',e)
PY
