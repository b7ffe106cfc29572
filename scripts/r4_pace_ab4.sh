#!/bin/bash
O=gpurun_out
run() { name=$1; shift; env "$@" python bench.py --no-side-runs --no-cpu $EXTRA > $O/r4_pace4_$name.json 2>> $O/r4_pace4.err; }
EXTRA="--n 4194304 --chunk-factor 6 --steps 30" run n22_base PSAMD_PACE=0 PSAMD_NW_PACKS=0
EXTRA="--n 4194304 --chunk-factor 6 --steps 30" run n22_b20 PSAMD_PACE=20
EXTRA="--n 4194304 --steps 30" run n22g16_base PSAMD_PACE=0 PSAMD_NW_PACKS=0
EXTRA="--n 4194304 --steps 30" run n22g16_b20 PSAMD_PACE=20
EXTRA="--n 262144 --steps 100" run n18_base PSAMD_PACE=0 PSAMD_NW_PACKS=0
EXTRA="--n 262144 --steps 100" run n18_b20 PSAMD_PACE=20
EXTRA="--steps 100" run n20_b20 PSAMD_PACE=20
python - <<'PY'
import json
for f in ("n22_base","n22_b20","n22g16_base","n22g16_b20","n18_base","n18_b20","n20_b20"):
    try:
        d=json.load(open('gpurun_out/r4_pace4_%s.json'%f)); print(f, round(d['ms_per_step'],4), round(d['kernel_us_per_step']['pairs'],1), round(d['roofline']['frac'],4))
    except Exception as e: print(f,'ERR',e)
PY
