#!/bin/bash
# more wave slots than resident waves in the balanced force pass (oldest-first issue makes the seven waves of a SIMD end one after the other)
O=gpurun_out
PSAMD_LIB=$PWD/scripts/libpsamd_trace.so python scripts/wave_trace.py 1 0 > $O/r4_wave_trace2.txt 2>&1; tail -12 $O/r4_wave_trace2.txt
for w in 0 8192 10240 12288 14336 16384; do
  PSAMD_WAVES=$w python bench.py --no-side-runs --no-cpu --steps 100 > $O/r4_waves_$w.json 2>> $O/r4_waves.err
done
python - <<'PY'
import json
for w in (0,8192,10240,12288,14336,16384):
    try:
        d=json.load(open('gpurun_out/r4_waves_%d.json'%w)); print(w, round(d['ms_per_step'],4), round(d['kernel_us_per_step']['pairs'],1), round(d['roofline']['frac'],4))
    except Exception as e: print(w,'ERR',e)
PY
