#!/bin/bash
# waves of the balanced force pass paced against the clock (issue priority by lag): A/B, parity, trace
O=gpurun_out
for p in 0 1; do PSAMD_PACE=$p python bench.py --no-side-runs --no-cpu --steps 100 > $O/r4_pace_$p.json 2>> $O/r4_pace.err; done
PSAMD_PACE=1 python bench.py --no-side-runs --no-cpu --steps 100 --fast-math > $O/r4_pace_1_fast.json 2>> $O/r4_pace.err
PSAMD_PACE=0 python bench.py --no-side-runs --no-cpu --steps 100 --fast-math > $O/r4_pace_0_fast.json 2>> $O/r4_pace.err
PSAMD_LIB=$PWD/scripts/libpsamd_trace.so python scripts/wave_trace.py 1 0 > $O/r4_wave_trace_paced.txt 2>&1; tail -11 $O/r4_wave_trace_paced.txt
python - <<'PY'
import json
for f in ("0","1","0_fast","1_fast"):
    try:
        d=json.load(open('gpurun_out/r4_pace_%s.json'%f)); print(f, round(d['ms_per_step'],4), round(d['kernel_us_per_step']['pairs'],1), round(d['roofline']['frac'],4))
    except Exception as e: print(f,'ERR',e)
PY
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -2
