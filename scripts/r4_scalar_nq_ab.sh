#!/bin/bash
# scalar walk at <= 4 waves per SIMD (N = 2^22 on eight ranks; a slab of two at N = 2^20): 16 bodies per group (PSAMD_SCALAR_NQ=16)?
O=gpurun_out
PSAMD_SCALAR_NQ=16 python -m pytest tests/test_gpu_slab.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
python bench.py --sim-world 8 --steps 20 --warmup 5 --n 4194304 --chunk-factor 6 > $O/r4_snq8_n22_$i.json 2>> $O/r4_snq.err
PSAMD_SCALAR_NQ=16 python bench.py --sim-world 8 --steps 20 --warmup 5 --n 4194304 --chunk-factor 6 > $O/r4_snq16_n22_$i.json 2>> $O/r4_snq.err
done
python bench.py --sim-world 2 --steps 30 --warmup 5 > $O/r4_snq8_w2.json 2>> $O/r4_snq.err
PSAMD_SCALAR_NQ=16 PSAMD_WAVES=4096 python bench.py --sim-world 2 --steps 30 --warmup 5 > $O/r4_snq16_w2.json 2>> $O/r4_snq.err
python - <<'PY'
import json
for f in ("8_n22_1","16_n22_1","8_n22_2","16_n22_2","8_w2","16_w2"):
    try:
        d=json.load(open('gpurun_out/r4_snq%s.json'%f)); print(f, round(d['modelled_step_ms'],4), d['stage_ms_per_rank']['pairs'])
    except Exception as e: print(f,'ERR',e)
PY
