#!/bin/bash
# Round-3 measurement set (on the GPU box): bench lines, rocprofv3 kernel stats, PMC passes, projections.
# Everything lands under gpurun_out/r3art/; the summaries that are judged get copied to profiles/ by hand.
out=gpurun_out/r3art
mkdir -p $out
export TMPDIR=/tmp
say() { echo "[$(date +%H:%M:%S)] $*" | tee -a $out/progress.log; }
say "default bench line"
python bench.py > $out/bench_line.json 2> $out/bench_line.err
say "kernel stats (exact)"
bash scripts/profile_bench.sh r3art_exact --steps 20 --warmup 3 --settle-seconds 0 --no-side-runs > $out/prof_exact.log 2>&1
say "kernel stats (tolerance mode)"
bash scripts/profile_bench.sh r3art_fast --fast-math --steps 20 --warmup 3 --settle-seconds 0 --no-side-runs > $out/prof_fast.log 2>&1
say "PMC fetch / write"
bash scripts/pmc_bench.sh r3art_fetch FETCH_SIZE --steps 3 --warmup 1 --settle-seconds 0 --no-side-runs > $out/pmc_fetch.log 2>&1
bash scripts/pmc_bench.sh r3art_write WRITE_SIZE --steps 3 --warmup 1 --settle-seconds 0 --no-side-runs > $out/pmc_write.log 2>&1
say "PMC SQ (exact)"
bash scripts/pmc_bench.sh r3art_sq "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE" --steps 3 --warmup 1 --settle-seconds 0 --no-side-runs > $out/pmc_sq.log 2>&1
bash scripts/pmc_bench.sh r3art_wc "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" --steps 3 --warmup 1 --settle-seconds 0 --no-side-runs > $out/pmc_wc.log 2>&1
say "PMC SQ (tolerance mode)"
bash scripts/pmc_bench.sh r3art_fast_sq "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE" --fast-math --steps 3 --warmup 1 --settle-seconds 0 --no-side-runs > $out/pmc_fast_sq.log 2>&1
bash scripts/pmc_bench.sh r3art_fast_wc "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" --fast-math --steps 3 --warmup 1 --settle-seconds 0 --no-side-runs > $out/pmc_fast_wc.log 2>&1
say "all-pairs (configs[1])"
python bench.py --all-pairs --steps 10 --warmup 3 > $out/allpairs_line.json 2> $out/allpairs_line.err
bash scripts/profile_bench.sh r3art_allpairs --all-pairs --steps 5 --warmup 2 --settle-seconds 0 > $out/prof_allpairs.log 2>&1
say "projections"
python bench.py --no-cpu --no-side-runs --steps 30 --warmup 5 > $out/one_n20_s30.json 2> /dev/null
python bench.py --sim-world 2 --steps 30 --warmup 5 > $out/sim2_n20.json 2> $out/sim2_n20.err
python bench.py --sim-world 4 --steps 30 --warmup 5 > $out/sim4_n20.json 2> $out/sim4_n20.err
python bench.py --sim-world 8 --steps 30 --warmup 5 > $out/sim8_n20.json 2> $out/sim8_n20.err
python bench.py --no-cpu --no-side-runs --steps 30 --warmup 5 --n 4194304 --chunk-factor 6 > $out/one_n22_grid24_s30.json 2> /dev/null
python bench.py --sim-world 8 --steps 30 --warmup 5 --n 4194304 --chunk-factor 6 > $out/sim8_n22_grid24.json 2> $out/sim8_n22.err
python bench.py --no-cpu --no-side-runs --steps 10 --warmup 3 --n 4194304 > $out/one_n22_grid16.json 2> /dev/null
python bench.py --sim-world 8 --all-pairs --steps 5 --warmup 2 > $out/sim8_allpairs.json 2> $out/sim8_allpairs.err
say "all-pairs at N=2^20 and N=2^22 (BASELINE configs[3]: 8 ranks, all-gather of the snapshot)"
python bench.py --all-pairs --n 1048576 --no-cpu --no-side-runs --steps 3 --warmup 1 > $out/ap_n20_one.json 2> /dev/null
python bench.py --sim-world 8 --all-pairs --n 1048576 --steps 3 --warmup 1 > $out/ap_n20_sim8.json 2> $out/ap_n20_sim8.err
python bench.py --all-pairs --n 4194304 --no-cpu --no-side-runs --steps 2 --warmup 1 > $out/ap_n22_one.json 2> /dev/null
python bench.py --sim-world 8 --all-pairs --n 4194304 --steps 2 --warmup 1 > $out/ap_n22_sim8.json 2> $out/ap_n22_sim8.err
say "rocprof of the 8-rank projection"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3art_sim8 -o sim -- python3 bench.py --sim-world 8 --steps 10 --warmup 3 > $out/sim8_prof.json 2> $out/sim8_prof.err
say "done"
