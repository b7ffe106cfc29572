#!/bin/bash
O=gpurun_out
python -m pytest tests/test_gpu_slab.py tests/test_gpu_fuzz.py tests/test_gpu_extras.py tests/test_host_driver.py -m gpu -x -q > $O/r4_fuse_tests.txt 2>&1 || { tail -30 $O/r4_fuse_tests.txt; exit 1; }
tail -2 $O/r4_fuse_tests.txt
python bench.py --sim-world 8 --steps 30 --warmup 5 --n 4194304 --chunk-factor 6 > $O/r4_fuse_n22.json 2> $O/r4_fuse.err
python bench.py --sim-world 8 --steps 30 --warmup 5 > $O/r4_fuse_n20.json 2>> $O/r4_fuse.err
python -c "
import json
for f in ('n22','n20'):
    d=json.load(open('gpurun_out/r4_fuse_%s.json'%f)); print(f, d['stage_ms_per_rank']['pairs'], d['stage_ms_per_rank']['build'], round(d['modelled_step_ms'],4))
"
