#!/bin/bash
# A/B: two waves to a task on a small share (PSAMD_PAIRED=1: WALK 3, pairs_task_tile2) against the tile walk with one wave per task.
# An eighth and a quarter of the N = 2^20 cloud (bench.py --sim-world 8 / 4: every rank's stages one after the other on one GPU).
O=gpurun_out
for w in 8 4; do
  for v in 0 1; do
    for i in 1 2; do
      PSAMD_PAIRED=$v timeout -k 10 120 python bench.py --sim-world $w --steps 30 --warmup 5 > $O/r5_paired_w${w}_v${v}_$i.json 2>> $O/r5_paired.err
    done
  done
done
python - <<'PY'
import json
for w in (8,4):
    for v in (0,1):
        for i in (1,2):
            try:
                d=json.loads(open('gpurun_out/r5_paired_w%d_v%d_%d.json'%(w,v,i)).read().strip().splitlines()[-1])
                pr=d['stage_ms_per_rank']['pairs']
                print("world %d paired %d run %d: pair stage per rank (median ms): min %.4f median %.4f max %.4f; modelled step %.4f ms" % (w, v, i, min(pr), sorted(pr)[len(pr)//2], max(pr), d['modelled_step_ms']))
            except Exception as e: print(w,v,i,'ERR',e)
PY
