#!/bin/bash
# Copy the judged summaries of the last scripts/r4_artifacts.sh run from gpurun_out/ into profiles/ (run here, after gpurun merged them).
set -e
g=gpurun_out; a=$g/r4art; p=profiles
last() { tail -1 "$1"; }
last $a/bench_line.json > $p/r4_bench_line.json
last $a/bench_steps20.json > $p/r4_bench_steps20.json
last $a/allpairs_line.json > $p/r4_allpairs_line.json
last $a/one_n20_s30.json > $p/r4_bench_n20_steps30.json
last $a/one_n22_grid24_s30.json > $p/r4_bench_n22_grid24_steps30.json
last $a/one_n24_grid40.json > $p/r4_bench_n24_grid40.json
for w in 2 4 8; do last $a/sim${w}_n20.json > $p/r4_sim_world$w.json; done
last $a/sim8_n22_grid24.json > $p/r4_sim_world8_n22_grid24.json
last $a/sim8_n24_grid40.json > $p/r4_sim_world8_n24_grid40.json
last $a/ap_n20_one.json > $p/r4_allpairs_n20_line.json
last $a/ap_n20_sim8.json > $p/r4_sim_world8_allpairs_n20.json
for f in g0_s1 g1_s1 g0_s0 allpairs; do last $a/ring_w8_$f.json > $p/r4_ring_loopback_w8_$f.json; done
cp $g/prof_r4art_exact/bench_kernel_stats.csv $p/r4_bench_kernel_stats.csv
last $g/prof_r4art_exact/bench_stdout.json > $p/r4_bench_under_rocprof.json
cp $g/prof_r4art_fast/bench_kernel_stats.csv $p/r4_fast_kernel_stats.csv
last $g/prof_r4art_fast/bench_stdout.json > $p/r4_fast_under_rocprof.json
cp $g/prof_r4art_sim8/sim_kernel_stats.csv $p/r4_sim_world8_kernel_stats.csv
cp $g/prof_r4art_allpairs/bench_kernel_stats.csv $p/r4_allpairs_kernel_stats.csv
cp $g/pmc_r4art_fetch/pmc_counter_collection.csv $p/r4_pmc_fetch_size.csv
cp $g/pmc_r4art_write/pmc_counter_collection.csv $p/r4_pmc_write_size.csv
python scripts/make_traffic_json.py $p/r4_pmc_fetch_size.csv $p/r4_pmc_write_size.csv $p/r4_traffic.json
{
  echo "== exact arithmetic: SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE (means over launches) =="
  python scripts/pmc_summary.py $g/pmc_r4art_sq/pmc_counter_collection.csv k_
  echo "== exact arithmetic: SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA =="
  python scripts/pmc_summary.py $g/pmc_r4art_wc/pmc_counter_collection.csv k_pairs
  echo "== tolerance mode: SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE =="
  python scripts/pmc_summary.py $g/pmc_r4art_fast_sq/pmc_counter_collection.csv k_pairs
} > $p/r4_pmc_sq_summary.txt
ls -la $p | grep r4_ | wc -l
