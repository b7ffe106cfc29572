#!/bin/bash
# A/B: k_apply / k_scatter_lds leave runs of 1024 free slots unread (block_live from k_hist_lds); PSAMD_NO_BLOCK_SKIP=1: k_apply as before
O=gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_slab.py -m gpu -x -q > $O/r4_blockskip_tests.txt 2>&1 || { tail -20 $O/r4_blockskip_tests.txt; exit 1; }
tail -2 $O/r4_blockskip_tests.txt
for i in 1 2; do
  PSAMD_NO_BLOCK_SKIP=1 python bench.py --no-side-runs --no-cpu --steps 100 > $O/r4_blockskip_off$i.json 2>> $O/r4_blockskip.err
  python bench.py --no-side-runs --no-cpu --steps 100 > $O/r4_blockskip_on$i.json 2>> $O/r4_blockskip.err
done
python - <<'PY'
import json
for f in ("off1","on1","off2","on2"):
    try:
        d=json.load(open('gpurun_out/r4_blockskip_%s.json'%f)); k=d['kernel_us_per_step']; print(f, round(d['ms_per_step'],4), {x:round(k[x],1) for x in k})
    except Exception as e: print(f,'ERR',e)
PY
