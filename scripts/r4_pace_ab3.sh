#!/bin/bash
O=gpurun_out
run() { name=$1; shift; env "$@" python bench.py --no-side-runs --no-cpu --steps 100 $EXTRA > $O/r4_pace3_$name.json 2>> $O/r4_pace3.err; }
run base PSAMD_PACE=0 PSAMD_NW_PACKS=0
run b10 PSAMD_PACE=10
run b20 PSAMD_PACE=20
run b30 PSAMD_PACE=30
EXTRA=--fast-math run fast_base PSAMD_PACE=0 PSAMD_NW_PACKS=0
EXTRA=--fast-math run fast_b20 PSAMD_PACE=20
EXTRA="--n 4194304 --chunk-factor 6 --steps 30" run n22_base PSAMD_PACE=0 PSAMD_NW_PACKS=0
EXTRA="--n 4194304 --chunk-factor 6 --steps 30" run n22_b20 PSAMD_PACE=20
python - <<'PY'
import json
for f in ("base","b10","b20","b30","fast_base","fast_b20","n22_base","n22_b20"):
    try:
        d=json.load(open('gpurun_out/r4_pace3_%s.json'%f)); print(f, round(d['ms_per_step'],4), round(d['kernel_us_per_step']['pairs'],1), round(d['roofline']['frac'],4))
    except Exception as e: print(f,'ERR',e)
PY
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fast.py tests/test_gpu_slab.py -m gpu -x -q 2>&1 | tail -2
