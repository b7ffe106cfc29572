#!/bin/bash
O=gpurun_out
for i in 1 2 3; do
  PSAMD_PACE=0 PSAMD_NW_PACKS=0 python bench.py --no-side-runs --no-cpu --steps 100 > $O/r4_pace5_base$i.json 2>> $O/r4_pace5.err
  python bench.py --no-side-runs --no-cpu --steps 100 > $O/r4_pace5_new$i.json 2>> $O/r4_pace5.err
done
python - <<'PY'
import json
for f in ("base1","new1","base2","new2","base3","new3"):
    try:
        d=json.load(open('gpurun_out/r4_pace5_%s.json'%f)); print(f, round(d['ms_per_step'],4), round(d['kernel_us_per_step']['pairs'],1), round(d['roofline']['frac'],4), d['config']['particles_with_a_force_term'])
    except Exception as e: print(f,'ERR',e)
PY
