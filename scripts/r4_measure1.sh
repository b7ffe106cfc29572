#!/bin/bash
# round 4, first measurement set: the driver's line, graphs on one GPU, the C++ host's step in loopback (eight slabs on this GPU)
set -o pipefail
O=gpurun_out
python -m pytest tests/test_gpu_bench_multi.py -m gpu -x -q > $O/r4_t4.log 2>&1; tail -3 $O/r4_t4.log
python bench.py --steps 20 > $O/r4_bench_steps20.json 2> $O/r4_bench_steps20.err; tail -c 600 $O/r4_bench_steps20.json; echo
python bench.py --no-side-runs --no-cpu > $O/r4_bench_plain.json 2>> $O/r4_bench_steps20.err
python bench.py --no-side-runs --no-cpu --graphs > $O/r4_bench_graphs.json 2>> $O/r4_bench_steps20.err
for g in 1 0; do for s in 1 0; do
  host/ps_ring_rccl --loopback --world 8 --bench --n 1048576 --steps 50 --warmup 5 --halo-cap-cell 448 --xfer-cap 9216 --graphs $g --side-stream $s 2>> $O/r4_ring.err | grep psamd_ring > $O/r4_ring_w8_g${g}_s${s}.json
done; done
host/ps_ring_rccl --loopback --world 8 --bench --n 1048576 --steps 50 --warmup 5 --halo-cap-cell 448 --xfer-cap 9216 --wait 0 2>> $O/r4_ring.err | grep psamd_ring > $O/r4_ring_w8_spin.json
python bench.py --sim-world 8 --steps 30 --warmup 5 > $O/r4_sim_world8_first.json 2>> $O/r4_ring.err
python bench.py --sim-world 8 --steps 30 --warmup 5 --graphs > $O/r4_sim_world8_first_graphs.json 2>> $O/r4_ring.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4_bench_*.json')):
    try:
        d=json.load(open(f)); print(f, d['ms_per_step'], d['value'], d.get('sustained'), d['config'].get('graphs'), {k:round(v,1) for k,v in d['kernel_us_per_step'].items()})
    except Exception as e: print(f, 'ERR', e)
for f in sorted(glob.glob('gpurun_out/r4_ring_w8_*.json')):
    try:
        d=json.load(open(f)); print(f, 'ms/step', 1e3*d['elapsed_s']/d['steps'], d['graph_replays'], d['kernel_us'])
    except Exception as e: print(f, 'ERR', e)
for f in sorted(glob.glob('gpurun_out/r4_sim_world8_first*.json')):
    try:
        d=json.load(open(f)); print(f, d['modelled_step_ms'], d['modelled_step_ms_optimistic'], d['compute_ms_per_rank'])
    except Exception as e: print(f, 'ERR', e)
PY
