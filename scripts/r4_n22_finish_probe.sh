#!/bin/bash
# which kernel makes slab_finish twice as long on the end ranks at N = 2^22?  max / mean duration per kernel of an 8-rank projection
cd /tmp 2>/dev/null; cd - >/dev/null; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_n22_sim8 -o sim -- python3 bench.py --sim-world 8 --steps 6 --warmup 2 --n 4194304 --chunk-factor 6 > gpurun_out/n22_sim8_prof.json 2> gpurun_out/n22_sim8_prof.err
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_n22_sim8/**/sim_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:32]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(6), ('%.1f' % (float(r['AverageNs'])/1e3)).rjust(9), ('%.1f' % (float(r['MaxNs'])/1e3)).rjust(9), r['Percentage'])
PY
