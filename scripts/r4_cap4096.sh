#!/bin/bash
# the 4096-operation replay instance on the last step's hint: parity, then the N = 2^22 projection on eight ranks
O=gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_slab.py tests/test_gpu_graphs.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/r4_cap4096_tests.txt 2>&1 || { tail -30 $O/r4_cap4096_tests.txt; exit 1; }
tail -2 $O/r4_cap4096_tests.txt
python bench.py --sim-world 8 --steps 30 --warmup 5 --n 4194304 --chunk-factor 6 > $O/r4_cap4096_n22.json 2> $O/r4_cap4096.err
python bench.py --sim-world 8 --steps 30 --warmup 5 > $O/r4_cap4096_n20.json 2>> $O/r4_cap4096.err
python -c "
import json
for f in ('n22','n20'):
    d=json.load(open('gpurun_out/r4_cap4096_%s.json'%f)); print(f, d['max_ops_one_queue_per_rank'], d['stage_ms_per_rank']['finish'], round(d['modelled_step_ms'],4))
"
