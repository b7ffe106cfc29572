#!/bin/bash
O=gpurun_out
run() { name=$1; shift; env "$@" python bench.py --no-side-runs --no-cpu --steps 100 > $O/r4_pace2_$name.json 2>> $O/r4_pace2.err; }
run base PSAMD_PACE=0 PSAMD_NW_PACKS=0
run nw PSAMD_PACE=0 PSAMD_NW_PACKS=1
run pace20 PSAMD_PACE=20 PSAMD_NW_PACKS=0
run both20 PSAMD_PACE=20 PSAMD_NW_PACKS=1
run both5 PSAMD_PACE=5 PSAMD_NW_PACKS=1
run both50 PSAMD_PACE=50 PSAMD_NW_PACKS=1
run both150 PSAMD_PACE=150 PSAMD_NW_PACKS=1
run base2 PSAMD_PACE=0 PSAMD_NW_PACKS=0
PSAMD_LIB=$PWD/scripts/libpsamd_trace.so python scripts/wave_trace.py 1 0 > $O/r4_wave_trace_paced2.txt 2>&1; tail -9 $O/r4_wave_trace_paced2.txt
python - <<'PY'
import json
for f in ("base","nw","pace20","both20","both5","both50","both150","base2"):
    try:
        d=json.load(open('gpurun_out/r4_pace2_%s.json'%f)); print(f, round(d['ms_per_step'],4), round(d['kernel_us_per_step']['pairs'],1), round(d['roofline']['frac'],4))
    except Exception as e: print(f,'ERR',e)
PY
