#!/bin/bash
# the C++ host's compute stream: created before the contexts (0), the first context's own (1), created after the first context (2)
out=gpurun_out/r5_host_ab4.txt; : > $out
R="host/ps_ring_rccl --world 1 --rank 0 --device 0 --id-file /tmp/psamd_ab_$$ --job 4242 --bench --n 1048576 --seed 2026 --max-particles 1048576 --settle-seconds 0.5 --steps 150 --warmup 5"
L="host/ps_ring_rccl --loopback --world 8 --bench --n 1048576 --steps 50 --warmup 5 --halo-cap-cell 310 --xfer-cap 9216"
one() { echo "== $1" >> $out; shift; rm -f /tmp/psamd_ab_$$*; env "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernel_us_median']
        print('   ms_per_step %.4f  pairs %.1f' % (1e3 * d['elapsed_s'] / d['steps'], k['pairs']))" >> $out; }
for rep in 1 2; do
for m in 0 1 3 4; do one "one rank, stream mode $m" PSAMD_RING_STREAM=$m timeout -k 10 120 $R; done
done

cat $out
