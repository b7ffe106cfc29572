#!/bin/bash
out=gpurun_out/r5_host_ab3.txt; : > $out
R="host/ps_ring_rccl --world 1 --rank 0 --device 0 --id-file /tmp/psamd_ab_$$ --job 4242 --bench --n 1048576 --seed 2026 --max-particles 1048576 --settle-seconds 0.5 --steps 150 --warmup 5"
for rep in 1 2; do
echo "== ring (C++, system HIP runtime)" >> $out; rm -f /tmp/psamd_ab_$$*
timeout -k 10 120 $R 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernel_us_median']
        print('   ms_per_step %.4f  pairs %.1f' % (1e3 * d['elapsed_s'] / d['steps'], k['pairs']))" >> $out
echo "== Python, no torch in the process (system HIP runtime): scripts/r5_stage_calls_ab.py" >> $out
timeout -k 10 120 python scripts/r5_stage_calls_ab.py 2>&1 | head -3 >> $out
echo "== bench.py --host python (torch imported first: its bundled HIP runtime)" >> $out
timeout -k 10 120 python bench.py --steps 150 --warmup 5 --no-side-runs --no-cpu --host python 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ms_per_step %.4f  pairs %.1f' % (d['ms_per_step'], d['kernel_us_per_step']['pairs']))" >> $out
done
cat $out
