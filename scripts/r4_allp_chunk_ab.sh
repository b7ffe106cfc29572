#!/bin/bash
# all-pairs far pass: cells per chain (ALLP_CHUNK) 4 (tree) / 8 / 16
O=gpurun_out
for ch in 4 8 16; do
  L=""; [ $ch != 4 ] && L=$PWD/scripts/libpsamd_chunk$ch.so
  PSAMD_LIB=$L python bench.py --all-pairs --no-side-runs --no-cpu --steps 10 > $O/r4_allp_chunk${ch}_n18.json 2>> $O/r4_allp_chunk.err
  PSAMD_LIB=$L python bench.py --all-pairs --n 1048576 --no-side-runs --no-cpu --steps 3 --warmup 1 > $O/r4_allp_chunk${ch}_n20.json 2>> $O/r4_allp_chunk.err
done
PSAMD_LIB=$PWD/scripts/libpsamd_chunk16.so python -m pytest tests/test_gpu_extras.py -m gpu -x -q -s -k all_pairs 2>&1 | grep -E "passed|failed|deviation|sum a"
python - <<'PY'
import json
for ch in (4,8,16):
  for f in ("n18","n20"):
    try:
        d=json.load(open('gpurun_out/r4_allp_chunk%d_%s.json'%(ch,f))); print(ch, f, round(d['ms_per_step'],3), round(d['roofline']['frac'],4))
    except Exception as e: print(ch,f,'ERR',e)
PY
