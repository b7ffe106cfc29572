#!/bin/bash
# Round-5 measurement set (on the GPU box): bench lines, rocprofv3 kernel stats, PMC passes, projections, the C++ host in loopback.
# Everything lands under gpurun_out/r5art/; scripts/r5_collect.sh copies the judged summaries to profiles/.
out=gpurun_out/r5art
mkdir -p $out
export TMPDIR=/tmp
say() { echo "[$(date +%H:%M:%S)] $*" | tee -a $out/progress.log; }
RING="host/ps_ring_rccl --world 1 --rank 0 --device 0 --bench --n 1048576 --seed 2026 --max-particles 1048576 --settle-seconds 0.5"
say "PMC fetch / write"
bash scripts/pmc_bench.sh r5art_fetch FETCH_SIZE --host python --steps 3 --warmup 1 --settle-seconds 0 --no-side-runs > $out/pmc_fetch.log 2>&1
bash scripts/pmc_bench.sh r5art_write WRITE_SIZE --host python --steps 3 --warmup 1 --settle-seconds 0 --no-side-runs > $out/pmc_write.log 2>&1
# (the lines below quote the counted traffic: the file they read is made from the two passes above, here on the box;
#  scripts/r5_collect.sh makes the same file from the same CSVs for the repository)
python scripts/make_traffic_json.py gpurun_out/pmc_r5art_fetch/pmc_counter_collection.csv gpurun_out/pmc_r5art_write/pmc_counter_collection.csv profiles/r5_traffic.json >> $out/pmc_write.log 2>&1
say "default bench line (the C++ host), the driver's form of it (--steps 20), and the Python host beside it"
python bench.py > $out/bench_line.json 2> $out/bench_line.err
python bench.py --steps 20 --warmup 5 --no-cpu --no-side-runs > $out/bench_steps20.json 2>> $out/bench_line.err
python bench.py --host python --no-cpu --no-side-runs > $out/bench_python_host.json 2>> $out/bench_line.err
say "kernel stats: the headline's own host (C++), then the Python host (same kernels), then the tolerance mode"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r5art_ring -o bench -- $RING --steps 100 --warmup 5 --timing-period 8 > $out/prof_ring_stdout.json 2> $out/prof_ring.err
bash scripts/profile_bench.sh r5art_exact --host python --steps 100 --warmup 5 --no-side-runs > $out/prof_exact.log 2>&1
bash scripts/profile_bench.sh r5art_fast --host python --fast-math --steps 50 --warmup 5 --no-side-runs > $out/prof_fast.log 2>&1
say "PMC SQ (exact)"
bash scripts/pmc_bench.sh r5art_sq "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE" --host python --steps 3 --warmup 2 --settle-seconds 0 --no-side-runs > $out/pmc_sq.log 2>&1
bash scripts/pmc_bench.sh r5art_wc "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" --host python --steps 3 --warmup 2 --settle-seconds 0 --no-side-runs > $out/pmc_wc.log 2>&1
say "all-pairs (configs[1])"
python bench.py --all-pairs --steps 10 --warmup 3 > $out/allpairs_line.json 2> $out/allpairs_line.err
say "projections, N = 2^20"
python bench.py --host python --no-cpu --no-side-runs --steps 30 --warmup 5 > $out/one_n20_s30.json 2> /dev/null
for w in 2 4 8; do python bench.py --sim-world $w --steps 30 --warmup 5 > $out/sim${w}_n20.json 2> $out/sim${w}_n20.err; done
say "projections, N = 2^22 in 24^3 cells"
python bench.py --host python --no-cpu --no-side-runs --steps 30 --warmup 5 --n 4194304 --chunk-factor 6 > $out/one_n22_grid24_s30.json 2> /dev/null
python bench.py --sim-world 8 --steps 30 --warmup 5 --n 4194304 --chunk-factor 6 > $out/sim8_n22_grid24.json 2> $out/sim8_n22.err
say "projections, N = 2^24 in 40^3 cells (BASELINE configs[4])"
python bench.py --host python --no-cpu --no-side-runs --steps 8 --warmup 2 --n 16777216 --chunk-factor 10 > $out/one_n24_grid40.json 2> $out/one_n24.err
python bench.py --sim-world 8 --steps 8 --warmup 2 --n 16777216 --chunk-factor 10 > $out/sim8_n24_grid40.json 2> $out/sim8_n24.err
say "all-pairs across eight ranks (BASELINE configs[3]'s exchange), N = 2^20"
python bench.py --host python --all-pairs --n 1048576 --no-cpu --no-side-runs --steps 3 --warmup 1 > $out/ap_n20_one.json 2> /dev/null
python bench.py --sim-world 8 --all-pairs --n 1048576 --steps 3 --warmup 1 > $out/ap_n20_sim8.json 2> $out/ap_n20_sim8.err
say "the C++ host: eight slabs in one process, every message through RCCL (loopback); the three stream modes"
for s in 0 1 2; do
  host/ps_ring_rccl --loopback --world 8 --bench --n 1048576 --steps 50 --warmup 5 --halo-cap-cell 310 --xfer-cap 9216 --side-stream $s 2>> $out/ring.err | grep psamd_ring > $out/ring_w8_s$s.json
done
host/ps_ring_rccl --loopback --world 8 --bench --all-pairs --n 262144 --steps 10 --warmup 2 --xfer-cap 24576 2>> $out/ring.err | grep psamd_ring > $out/ring_w8_allpairs.json
say "what a cross-stream dependency costs: ONE rank, no messages, the three stream modes (--side-stream 2 waits on events between all stages)"
for s in 0 2; do
  $RING --steps 100 --warmup 5 --side-stream $s 2>> $out/ring.err | grep psamd_ring > $out/ring_w1_s$s.json
done
say "done"
