#!/bin/bash
# the DPP walk against the scalar-load / LDS-tile walks: parity first, then one GPU and an eighth's share
O=gpurun_out
PSAMD_DPP=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_slab.py tests/test_gpu_fast.py -m gpu -x -q > $O/r4_dpp_parity.log 2>&1; tail -3 $O/r4_dpp_parity.log
for v in "0 0" "1 6" "1 5" "1 4"; do set -- $v
  PSAMD_DPP=$1 PSAMD_DPP_WAVES_PER_SIMD=$2 python bench.py --no-side-runs --no-cpu --steps 100 > $O/r4_dpp_one_$1_$2.json 2>> $O/r4_dpp.err
  PSAMD_DPP=$1 PSAMD_DPP_WAVES_PER_SIMD=$2 python bench.py --sim-world 8 --steps 30 --warmup 5 > $O/r4_dpp_w8_$1_$2.json 2>> $O/r4_dpp.err
done
PSAMD_DPP=1 python bench.py --no-side-runs --no-cpu --steps 100 --fast-math > $O/r4_dpp_one_fast.json 2>> $O/r4_dpp.err
PSAMD_DPP=0 python bench.py --no-side-runs --no-cpu --steps 100 --fast-math > $O/r4_dpp_one_fast0.json 2>> $O/r4_dpp.err
for w in 1 2 3; do PSAMD_DPP=1 PSAMD_WAVES=$((1024*w)) python bench.py --sim-world 8 --steps 30 --warmup 5 > $O/r4_dpp_w8_waves$w.json 2>> $O/r4_dpp.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4_dpp_one_*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['ms_per_step'],4), {k:round(v,1) for k,v in d['kernel_us_per_step'].items()}, round(d['roofline']['frac'],4))
    except Exception as e: print(f,'ERR',e)
for f in sorted(glob.glob('gpurun_out/r4_dpp_w8_*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['modelled_step_ms'],4), d['stage_ms_per_rank']['pairs'])
    except Exception as e: print(f,'ERR',e)
PY
