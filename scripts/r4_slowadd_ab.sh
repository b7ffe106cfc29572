#!/bin/bash
# A/B: the double-precision EPS2 add per body (new) against per group (scripts/libpsamd_prev.so = HEAD before it)
O=gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fast.py -m gpu -x -q > $O/r4_slowadd_tests.txt 2>&1 || { tail -20 $O/r4_slowadd_tests.txt; exit 1; }
tail -2 $O/r4_slowadd_tests.txt
for i in 1 2 3; do
  PSAMD_LIB=$PWD/scripts/libpsamd_prev.so python bench.py --no-side-runs --no-cpu --steps 100 > $O/r4_slowadd_prev$i.json 2>> $O/r4_slowadd.err
  python bench.py --no-side-runs --no-cpu --steps 100 > $O/r4_slowadd_new$i.json 2>> $O/r4_slowadd.err
done
PSAMD_LIB=$PWD/scripts/libpsamd_prev.so python bench.py --all-pairs --no-side-runs --no-cpu --steps 10 > $O/r4_slowadd_ap_prev.json 2>> $O/r4_slowadd.err
python bench.py --all-pairs --no-side-runs --no-cpu --steps 10 > $O/r4_slowadd_ap_new.json 2>> $O/r4_slowadd.err
python - <<'PY'
import json
for f in ("prev1","new1","prev2","new2","prev3","new3","ap_prev","ap_new"):
    try:
        d=json.load(open('gpurun_out/r4_slowadd_%s.json'%f)); print(f, round(d['ms_per_step'],4), d['roofline']['frac'])
    except Exception as e: print(f,'ERR',e)
PY
