#!/bin/bash
# k_sort_cells' workgroups in the cells' own order (PSAMD_CELL_ORDER=0) or segment-major, one contiguous run per XCD (1)
out=gpurun_out/r5_cellorder_ab.txt; : > $out
for rep in 1 2; do for v in 0 1; do
  export PSAMD_CELL_ORDER=$v
  timeout -k 10 200 bash scripts/profile_bench.sh co_${v}_$rep --host python --steps 60 --warmup 5 --no-side-runs > /dev/null 2>&1
  echo "== PSAMD_CELL_ORDER=$v run $rep" >> $out
  python3 - "gpurun_out/prof_co_${v}_$rep" >> $out <<'PY'
import sys, csv, glob, json
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    for k in ('k_sort_cells', 'k_collide_cell', 'k_apply', 'k_scatter_lds', 'k_pairs_balanced'):
        if k in n: print('   %-18s avg %.2f us' % (k, float(r['AverageNs']) / 1e3))
d = json.loads([l for l in open(sys.argv[1] + '/bench_stdout.json') if l.startswith('{')][-1])
print('   ms_per_step %.4f' % d['ms_per_step'])
PY
done; done
cat $out
