#!/bin/bash
O=gpurun_out
run() { name=$1; shift; env "$@" python bench.py --no-side-runs --no-cpu $EXTRA > $O/r4_packs_$name.json 2>> $O/r4_packs.err; }
EXTRA="--steps 100" run n20_off PSAMD_PACE=0 PSAMD_NW_PACKS=0
EXTRA="--steps 100" run n20_c10 PSAMD_PACK_COST=1.0
EXTRA="--steps 100" run n20_c14 PSAMD_PACK_COST=1.4
EXTRA="--steps 100" run n20_c18 PSAMD_PACK_COST=1.8
EXTRA="--steps 100" run n20_c24 PSAMD_PACK_COST=2.4
EXTRA="--steps 100 --fast-math" run n20f_off PSAMD_PACE=0 PSAMD_NW_PACKS=0
EXTRA="--steps 100 --fast-math" run n20f_c14 PSAMD_PACK_COST=1.4
EXTRA="--steps 30 --n 4194304 --chunk-factor 6" run n22_off PSAMD_PACE=0 PSAMD_NW_PACKS=0
EXTRA="--steps 30 --n 4194304 --chunk-factor 6" run n22_c14 PSAMD_PACK_COST=1.4
EXTRA="--steps 30 --n 4194304 --chunk-factor 6" run n22_c20 PSAMD_PACK_COST=2.0
EXTRA="--steps 30 --n 4194304" run n22g16_off PSAMD_PACE=0 PSAMD_NW_PACKS=0
EXTRA="--steps 30 --n 4194304" run n22g16_c14 PSAMD_PACK_COST=1.4
EXTRA="--steps 100 --n 262144" run n18_off PSAMD_PACE=0 PSAMD_NW_PACKS=0
EXTRA="--steps 100 --n 262144" run n18_c14 PSAMD_PACK_COST=1.4
python - <<'PY'
import json
for f in ("n20_off","n20_c10","n20_c14","n20_c18","n20_c24","n20f_off","n20f_c14","n22_off","n22_c14","n22_c20","n22g16_off","n22g16_c14","n18_off","n18_c14"):
    try:
        d=json.load(open('gpurun_out/r4_packs_%s.json'%f)); print(f, round(d['ms_per_step'],4), round(d['kernel_us_per_step']['pairs'],1), round(d['roofline']['frac'],4), (d.get('shader_clock_mhz') or {}).get('median'))
    except Exception as e: print(f,'ERR',e)
PY
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -2
