#!/bin/bash
out=gpurun_out/r5_host_ab2.txt; : > $out
R="host/ps_ring_rccl --world 1 --rank 0 --device 0 --id-file /tmp/psamd_ab_$$ --job 4242 --bench --n 1048576 --seed 2026 --max-particles 1048576 --settle-seconds 0.5 --steps 150 --warmup 5"
one() { echo "== $1" >> $out; shift; rm -f /tmp/psamd_ab_$$*; env "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        k = d.get('kernel_us_median') or d.get('kernel_us_per_step')
        ms = d['ms_per_step'] if 'ms_per_step' in d else 1e3 * d['elapsed_s'] / d['steps']
        print('   ms_per_step %.4f  pairs %.1f apply %.1f' % (ms, k['pairs'], k['apply']))
" >> $out; }
for rep in 1 2 3; do
one "ring default" X=1 timeout -k 10 120 $R
one "ring, no communicator" PSAMD_RING_NO_RCCL=1 timeout -k 10 120 $R
one "python host" X=1 timeout -k 10 120 python bench.py --steps 150 --warmup 5 --no-side-runs --no-cpu --host python
done
cat $out
