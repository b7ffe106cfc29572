#!/bin/bash
# A/B of the force pass's wave count / walk on a 1/8 share (sim-world 8).  usage (GPU box): bash scripts/ab_slab_waves.sh
out=gpurun_out/ab_slab_waves.log
: > $out
run() {
    label=$1; shift
    line=$(env "$@" python bench.py --sim-world 8 --steps 20 --warmup 3 2>/dev/null | tail -1)
    echo "$label $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); p=d["stage_ms_per_rank"]["pairs"]; print("step %.4f opt %.4f pairs interior-rank %.4f max %.4f" % (d["modelled_step_ms"], d["modelled_step_ms_optimistic"], p[3], max(p)))')" >> $out
}
run "default (1024 waves, tile)" X=1
run "2048 tile                 " PSAMD_WAVES=2048 PSAMD_TILE=1
run "2048 scalar               " PSAMD_WAVES=2048 PSAMD_TILE=0
run "3072 scalar               " PSAMD_WAVES=3072 PSAMD_TILE=0
run "4096 scalar               " PSAMD_WAVES=4096 PSAMD_TILE=0
run "7168 scalar               " PSAMD_WAVES=7168 PSAMD_TILE=0
run "1024 tile + packs         " PSAMD_TILE_PACKS=1
run "2048 tile + packs         " PSAMD_WAVES=2048 PSAMD_TILE=1 PSAMD_TILE_PACKS=1
cat $out
