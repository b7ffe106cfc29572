#!/bin/bash
# the all-pairs far pass by dense tasks: tests, then BASELINE configs[1] (N = 2^18) and N = 2^20
O=gpurun_out
python -m pytest tests/test_gpu_extras.py tests/test_host_driver.py tests/test_gpu_bench_multi.py tests/test_gpu_parity.py -m gpu -x -q -s > $O/r4_allp_tests.txt 2>&1 || { tail -30 $O/r4_allp_tests.txt; exit 1; }
tail -3 $O/r4_allp_tests.txt; grep -h "all-pairs N=2^18" $O/r4_allp_tests.txt
python bench.py --all-pairs --no-side-runs --no-cpu --steps 10 > $O/r4_allp_n18.json 2>> $O/r4_allp.err
python bench.py --all-pairs --fast-math --no-side-runs --no-cpu --steps 10 > $O/r4_allp_n18_fast.json 2>> $O/r4_allp.err
python bench.py --all-pairs --n 1048576 --no-side-runs --no-cpu --steps 3 --warmup 1 > $O/r4_allp_n20.json 2>> $O/r4_allp.err
python - <<'PY'
import json
for f in ("n18","n18_fast","n20"):
    try:
        d=json.load(open('gpurun_out/r4_allp_%s.json'%f)); print(f, round(d['ms_per_step'],3), round(d['roofline']['frac'],4), d['kernel_us_per_step'])
    except Exception as e: print(f,'ERR',e)
PY
