#!/bin/bash
# A/B: how many persistent pack workgroups the balanced force pass gets (PSAMD_PACK_COST: what one pack of partly filled
# slices is taken to cost, in ordinary tasks) -- the wave trace of round 5 showed the pack waves ending at 35-45 % of the launch
O=gpurun_out
for c in 1.4 1.0 0.75 0.55 0.4; do
  for i in 1 2; do
    PSAMD_PACK_COST=$c python bench.py --host python --no-side-runs --no-cpu --steps 100 > $O/r5_pc_${c}_$i.json 2>> $O/r5_pc.err
  done
done
python - <<'PY'
import json
for c in ("1.4","1.0","0.75","0.55","0.4"):
    for i in (1,2):
        try:
            d=json.loads(open('gpurun_out/r5_pc_%s_%d.json'%(c,i)).read().strip().splitlines()[-1]); print(c, i, round(d['ms_per_step'],4), round(d['kernel_us_per_step']['pairs'],1), round(d['roofline']['frac'],4))
        except Exception as e: print(c,i,'ERR',e)
PY
