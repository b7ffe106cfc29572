#!/bin/bash
# one GPU: where the C++ host's step differs from the Python host's (same library, same kernels)
out=gpurun_out/r5_host_ab.txt; : > $out
R="host/ps_ring_rccl --world 1 --rank 0 --device 0 --id-file /tmp/psamd_ab_$$ --job 4242 --bench --n 1048576 --seed 2026 --max-particles 1048576 --settle-seconds 0.5 --steps 150 --warmup 5"
one() { echo "== $1" >> $out; shift; rm -f /tmp/psamd_ab_$$*; timeout -k 10 120 "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        k = d.get('kernel_us_median') or d.get('kernel_us_per_step')
        ms = d['ms_per_step'] if 'ms_per_step' in d else 1e3 * d['elapsed_s'] / d['steps']
        print('   ms_per_step %.4f  pairs %.1f apply %.1f  clock %s' % (ms, k['pairs'], k['apply'], (d.get('shader_clock_mhz') or {}).get('median')))
" >> $out; }
for rep in 1 2; do
one "ring default" $R
one "python host" python bench.py --steps 150 --warmup 5 --no-side-runs --no-cpu --host python
one "ring, no timing events in the region" $R --timing-period 100000
one "ring, no clock watch" $R --clock-period-ms 0
one "ring, neither" $R --clock-period-ms 0 --timing-period 100000
done
cat $out
