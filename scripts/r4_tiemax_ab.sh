#!/bin/bash
# A/B: the tie test of the one-transcendental reciprocal as "largest residual >= 2^-24" (v_max3; -DPSAMD_TIE_MAX, the tree's lib) against a compare per residual (scripts/libpsamd_base.so)
O=gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_math.py -m gpu -x -q > $O/r4_tiemax_tests.txt 2>&1 || { tail -20 $O/r4_tiemax_tests.txt; exit 1; }
tail -2 $O/r4_tiemax_tests.txt
for i in 1 2 3; do
  PSAMD_LIB=$PWD/scripts/libpsamd_base.so python bench.py --no-side-runs --no-cpu --steps 100 > $O/r4_tiemax_base$i.json 2>> $O/r4_tiemax.err
  python bench.py --no-side-runs --no-cpu --steps 100 > $O/r4_tiemax_new$i.json 2>> $O/r4_tiemax.err
done
python - <<'PY'
import json
for f in ("base1","new1","base2","new2","base3","new3"):
    try:
        d=json.load(open('gpurun_out/r4_tiemax_%s.json'%f)); print(f, round(d['ms_per_step'],4), round(d['roofline']['frac'],4))
    except Exception as e: print(f,'ERR',e)
PY
