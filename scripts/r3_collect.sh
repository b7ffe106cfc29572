#!/bin/bash
# Copy the judged summaries of the last scripts/r3_artifacts.sh run from gpurun_out/ into profiles/ (run here, after gpurun merged them).
set -e
g=gpurun_out; a=$g/r3art; p=profiles
last() { tail -1 "$1"; }
last $a/bench_line.json > $p/r3_bench_line.json
last $a/allpairs_line.json > $p/r3_allpairs_line.json
last $a/one_n20_s30.json > $p/r3_bench_n20_steps30.json
last $a/one_n22_grid16.json > $p/r3_bench_n22_grid16.json
last $a/one_n22_grid24_s30.json > $p/r3_bench_n22_grid24_steps30.json
last $a/sim2_n20.json > $p/r3_sim_world2.json
last $a/sim4_n20.json > $p/r3_sim_world4.json
last $a/sim8_n20.json > $p/r3_sim_world8.json
last $a/sim8_n22_grid24.json > $p/r3_sim_world8_n22_grid24.json
last $a/sim8_allpairs.json > $p/r3_sim_world8_allpairs.json
for f in ap_n20_one:r3_allpairs_n20_line ap_n20_sim8:r3_sim_world8_allpairs_n20 ap_n22_one:r3_allpairs_n22_line ap_n22_sim8:r3_sim_world8_allpairs_n22; do
    [ -f $a/${f%%:*}.json ] && last $a/${f%%:*}.json > $p/${f##*:}.json
done
cp $g/prof_r3art_exact/bench_kernel_stats.csv $p/r3_bench_kernel_stats.csv
last $g/prof_r3art_exact/bench_stdout.json > $p/r3_bench_under_rocprof.json
cp $g/prof_r3art_fast/bench_kernel_stats.csv $p/r3_fast_kernel_stats.csv
last $g/prof_r3art_fast/bench_stdout.json > $p/r3_fast_under_rocprof.json
cp $g/prof_r3art_allpairs/bench_kernel_stats.csv $p/r3_allpairs_kernel_stats.csv
cp $g/prof_r3art_sim8/sim_kernel_stats.csv $p/r3_sim_world8_kernel_stats.csv
cp $g/pmc_r3art_fetch/pmc_counter_collection.csv $p/r3_pmc_fetch_size.csv
cp $g/pmc_r3art_write/pmc_counter_collection.csv $p/r3_pmc_write_size.csv
python scripts/make_traffic_json.py $p/r3_pmc_fetch_size.csv $p/r3_pmc_write_size.csv $p/r3_traffic.json
{
  echo "== exact arithmetic: SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE (means over launches) =="
  python scripts/pmc_summary.py $g/pmc_r3art_sq/pmc_counter_collection.csv k_
  echo "== exact arithmetic: SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA =="
  python scripts/pmc_summary.py $g/pmc_r3art_wc/pmc_counter_collection.csv k_pairs
  echo "== tolerance mode: SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE =="
  python scripts/pmc_summary.py $g/pmc_r3art_fast_sq/pmc_counter_collection.csv k_pairs
  echo "== tolerance mode: SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA =="
  python scripts/pmc_summary.py $g/pmc_r3art_fast_wc/pmc_counter_collection.csv k_pairs
} > $p/r3_pmc_sq_summary.txt
ls -la $p | grep r3_ | wc -l
