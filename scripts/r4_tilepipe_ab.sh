#!/bin/bash
O=gpurun_out
for i in 1 2; do
  PSAMD_LIB=$PWD/scripts/libpsamd_nopipe.so python bench.py --sim-world 8 --steps 30 --warmup 5 > $O/r4_tilepipe_off$i.json 2>> $O/r4_tilepipe.err
  python bench.py --sim-world 8 --steps 30 --warmup 5 > $O/r4_tilepipe_on$i.json 2>> $O/r4_tilepipe.err
done
PSAMD_LIB=$PWD/scripts/libpsamd_nopipe.so python bench.py --sim-world 4 --steps 30 --warmup 5 > $O/r4_tilepipe_w4_off.json 2>> $O/r4_tilepipe.err
python bench.py --sim-world 4 --steps 30 --warmup 5 > $O/r4_tilepipe_w4_on.json 2>> $O/r4_tilepipe.err
PSAMD_WAVES=2048 python bench.py --sim-world 8 --steps 30 --warmup 5 > $O/r4_tilepipe_on_2048.json 2>> $O/r4_tilepipe.err
python - <<'PY'
import json
for f in ("off1","on1","off2","on2","w4_off","w4_on","on_2048"):
    try:
        d=json.load(open('gpurun_out/r4_tilepipe_%s.json'%f)); print(f, round(d['modelled_step_ms'],4), d['stage_ms_per_rank']['pairs'])
    except Exception as e: print(f,'ERR',e)
PY
python -m pytest tests/test_gpu_slab.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
