#!/bin/bash
# Copy the judged summaries of the last scripts/r5_artifacts.sh run from gpurun_out/ into profiles/ (run here, after gpurun merged them).
set -e
g=gpurun_out; a=$g/r5art; p=profiles
last() { tail -1 "$1"; }
last $a/bench_line.json > $p/r5_bench_line.json
last $a/bench_steps20.json > $p/r5_bench_steps20.json
last $a/bench_python_host.json > $p/r5_bench_python_host.json
last $a/allpairs_line.json > $p/r5_allpairs_line.json
last $a/one_n20_s30.json > $p/r5_bench_n20_steps30.json
last $a/one_n22_grid24_s30.json > $p/r5_bench_n22_grid24_steps30.json
last $a/one_n24_grid40.json > $p/r5_bench_n24_grid40.json
for w in 2 4 8; do last $a/sim${w}_n20.json > $p/r5_sim_world$w.json; done
last $a/sim8_n22_grid24.json > $p/r5_sim_world8_n22_grid24.json
last $a/sim8_n24_grid40.json > $p/r5_sim_world8_n24_grid40.json
last $a/ap_n20_one.json > $p/r5_allpairs_n20_line.json
last $a/ap_n20_sim8.json > $p/r5_sim_world8_allpairs_n20.json
for f in s0 s1 s2 allpairs; do last $a/ring_w8_$f.json > $p/r5_ring_loopback_w8_$f.json; done
for f in s0 s2; do last $a/ring_w1_$f.json > $p/r5_ring_one_rank_$f.json; done
cp $g/prof_r5art_ring/bench_kernel_stats.csv $p/r5_bench_kernel_stats.csv
grep psamd_ring $a/prof_ring_stdout.json | tail -1 > $p/r5_bench_under_rocprof.json
cp $g/prof_r5art_exact/bench_kernel_stats.csv $p/r5_bench_kernel_stats_python_host.csv
cp $g/prof_r5art_fast/bench_kernel_stats.csv $p/r5_fast_kernel_stats.csv
last $g/prof_r5art_fast/bench_stdout.json > $p/r5_fast_under_rocprof.json
cp $g/pmc_r5art_fetch/pmc_counter_collection.csv $p/r5_pmc_fetch_size.csv
cp $g/pmc_r5art_write/pmc_counter_collection.csv $p/r5_pmc_write_size.csv
python scripts/make_traffic_json.py $p/r5_pmc_fetch_size.csv $p/r5_pmc_write_size.csv $p/r5_traffic.json
{
  echo "== exact arithmetic: SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE (means over launches) =="
  python scripts/pmc_summary.py $g/pmc_r5art_sq/pmc_counter_collection.csv k_
  echo "== exact arithmetic: SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA =="
  python scripts/pmc_summary.py $g/pmc_r5art_wc/pmc_counter_collection.csv k_pairs
} > $p/r5_pmc_sq_summary.txt
ls -la $p | grep r5_ | wc -l
