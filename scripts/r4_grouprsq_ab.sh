#!/bin/bash
# A/B: the eight v_rsq of a group of bodies issued back to back (-DPSAMD_GROUP_RSQ, the tree's libpsamd.so) against interleaved (scripts/libpsamd_base.so)
O=gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fast.py -m gpu -x -q > $O/r4_grouprsq_tests.txt 2>&1 || { tail -20 $O/r4_grouprsq_tests.txt; exit 1; }
tail -2 $O/r4_grouprsq_tests.txt
for i in 1 2 3; do
  PSAMD_LIB=$PWD/scripts/libpsamd_base.so python bench.py --no-side-runs --no-cpu --steps 100 > $O/r4_grouprsq_base$i.json 2>> $O/r4_grouprsq.err
  python bench.py --no-side-runs --no-cpu --steps 100 > $O/r4_grouprsq_new$i.json 2>> $O/r4_grouprsq.err
  PSAMD_LIB=$PWD/scripts/libpsamd_base.so python bench.py --fast-math --no-side-runs --no-cpu --steps 100 > $O/r4_grouprsq_fbase$i.json 2>> $O/r4_grouprsq.err
  python bench.py --fast-math --no-side-runs --no-cpu --steps 100 > $O/r4_grouprsq_fnew$i.json 2>> $O/r4_grouprsq.err
done
python - <<'PY'
import json
for f in ("base1","new1","base2","new2","base3","new3","fbase1","fnew1","fbase2","fnew2","fbase3","fnew3"):
    try:
        d=json.load(open('gpurun_out/r4_grouprsq_%s.json'%f)); print(f, round(d['ms_per_step'],4), round(d['roofline']['frac'],4))
    except Exception as e: print(f,'ERR',e)
PY
