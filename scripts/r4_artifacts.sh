#!/bin/bash
# Round-4 measurement set (on the GPU box): bench lines, rocprofv3 kernel stats, PMC passes, projections, the C++ host in loopback.
# Everything lands under gpurun_out/r4art/; scripts/r4_collect.sh copies the judged summaries to profiles/.
out=gpurun_out/r4art
mkdir -p $out
export TMPDIR=/tmp
say() { echo "[$(date +%H:%M:%S)] $*" | tee -a $out/progress.log; }
say "default bench line, and the driver's form of it (--steps 20)"
python bench.py > $out/bench_line.json 2> $out/bench_line.err
python bench.py --steps 20 --no-cpu --no-side-runs > $out/bench_steps20.json 2>> $out/bench_line.err
say "kernel stats (exact)"
bash scripts/profile_bench.sh r4art_exact --steps 20 --warmup 3 --settle-seconds 0 --no-side-runs > $out/prof_exact.log 2>&1
say "kernel stats (tolerance mode)"
bash scripts/profile_bench.sh r4art_fast --fast-math --steps 20 --warmup 3 --settle-seconds 0 --no-side-runs > $out/prof_fast.log 2>&1
say "PMC fetch / write"
bash scripts/pmc_bench.sh r4art_fetch FETCH_SIZE --steps 3 --warmup 1 --settle-seconds 0 --no-side-runs > $out/pmc_fetch.log 2>&1
bash scripts/pmc_bench.sh r4art_write WRITE_SIZE --steps 3 --warmup 1 --settle-seconds 0 --no-side-runs > $out/pmc_write.log 2>&1
say "PMC SQ (exact)"
bash scripts/pmc_bench.sh r4art_sq "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE" --steps 3 --warmup 2 --settle-seconds 0 --no-side-runs > $out/pmc_sq.log 2>&1
bash scripts/pmc_bench.sh r4art_wc "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" --steps 3 --warmup 2 --settle-seconds 0 --no-side-runs > $out/pmc_wc.log 2>&1
say "PMC SQ (tolerance mode)"
bash scripts/pmc_bench.sh r4art_fast_sq "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE" --fast-math --steps 3 --warmup 2 --settle-seconds 0 --no-side-runs > $out/pmc_fast_sq.log 2>&1
say "all-pairs (configs[1])"
python bench.py --all-pairs --steps 10 --warmup 3 > $out/allpairs_line.json 2> $out/allpairs_line.err
say "kernel stats (all-pairs)"
bash scripts/profile_bench.sh r4art_allpairs --all-pairs --steps 5 --warmup 2 --settle-seconds 0 --no-side-runs > $out/prof_allpairs.log 2>&1
say "projections, N = 2^20"
python bench.py --no-cpu --no-side-runs --steps 30 --warmup 5 > $out/one_n20_s30.json 2> /dev/null
for w in 2 4 8; do python bench.py --sim-world $w --steps 30 --warmup 5 > $out/sim${w}_n20.json 2> $out/sim${w}_n20.err; done
say "projections, N = 2^22 in 24^3 cells"
python bench.py --no-cpu --no-side-runs --steps 30 --warmup 5 --n 4194304 --chunk-factor 6 > $out/one_n22_grid24_s30.json 2> /dev/null
python bench.py --sim-world 8 --steps 30 --warmup 5 --n 4194304 --chunk-factor 6 > $out/sim8_n22_grid24.json 2> $out/sim8_n22.err
say "projections, N = 2^24 in 40^3 cells (BASELINE configs[4])"
python bench.py --no-cpu --no-side-runs --steps 8 --warmup 2 --n 16777216 --chunk-factor 10 > $out/one_n24_grid40.json 2> $out/one_n24.err
python bench.py --sim-world 8 --steps 8 --warmup 2 --n 16777216 --chunk-factor 10 > $out/sim8_n24_grid40.json 2> $out/sim8_n24.err
say "all-pairs across eight ranks (BASELINE configs[3]'s exchange), N = 2^20"
python bench.py --all-pairs --n 1048576 --no-cpu --no-side-runs --steps 3 --warmup 1 > $out/ap_n20_one.json 2> /dev/null
python bench.py --sim-world 8 --all-pairs --n 1048576 --steps 3 --warmup 1 > $out/ap_n20_sim8.json 2> $out/ap_n20_sim8.err
say "the C++ host: eight slabs in one process, every message through RCCL (loopback)"
for v in "0 1" "1 1" "0 0"; do set -- $v
  host/ps_ring_rccl --loopback --world 8 --bench --n 1048576 --steps 50 --warmup 5 --halo-cap-cell 310 --xfer-cap 9216 --graphs $1 --side-stream $2 2>> $out/ring.err | grep psamd_ring > $out/ring_w8_g$1_s$2.json
done
host/ps_ring_rccl --loopback --world 8 --bench --all-pairs --n 262144 --steps 10 --warmup 2 --xfer-cap 24576 2>> $out/ring.err | grep psamd_ring > $out/ring_w8_allpairs.json
say "rocprof of the 8-rank projection"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r4art_sim8 -o sim -- python3 bench.py --sim-world 8 --steps 10 --warmup 3 > $out/sim8_prof.json 2> $out/sim8_prof.err
say "done"
