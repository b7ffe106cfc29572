#!/bin/bash
# small shares (tile walk): the next group's distances between a group's scale factors and its additions (scripts/libpsamd_addpipe.so), 8 / 16 bodies per group
O=gpurun_out
L=$PWD/scripts/libpsamd_addpipe.so
PSAMD_LIB=$L python -m pytest tests/test_gpu_slab.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
  PSAMD_TILE_NQ=16 python bench.py --sim-world 8 --steps 30 --warmup 5 > $O/r4_addpipe_base16_$i.json 2>> $O/r4_addpipe.err
  PSAMD_LIB=$L python bench.py --sim-world 8 --steps 30 --warmup 5 > $O/r4_addpipe_p8_$i.json 2>> $O/r4_addpipe.err
  PSAMD_LIB=$L PSAMD_TILE_NQ=16 python bench.py --sim-world 8 --steps 30 --warmup 5 > $O/r4_addpipe_p16_$i.json 2>> $O/r4_addpipe.err
done
python - <<'PY'
import json
for f in ("base16_1","p8_1","p16_1","base16_2","p8_2","p16_2"):
    try:
        d=json.load(open('gpurun_out/r4_addpipe_%s.json'%f)); print(f, round(d['modelled_step_ms'],4), d['stage_ms_per_rank']['pairs'])
    except Exception as e: print(f,'ERR',e)
PY
