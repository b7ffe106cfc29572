#!/bin/bash
# small shares (a slab of four / eight ranks at N = 2^20): how many waves the pair stage's balanced pass is cut into, tile walk or scalar walk
out=gpurun_out/r5_small_waves_ab.txt; : > $out
run() { w=$1; label=$2; shift 2
  line=$(env "$@" timeout -k 10 200 python bench.py --sim-world $w --steps 30 --warmup 5 --no-cpu 2>/dev/null | tail -1)
  echo "world $w $label: $(echo "$line" | python3 -c '
import sys, json
d = json.loads(sys.stdin.read()); p = sorted(d["stage_ms_per_rank"]["pairs"])
print("pair stage per rank (median ms): min %.4f median %.4f max %.4f; modelled step %.4f ms" % (p[0], p[len(p)//2], p[-1], d["modelled_step_ms"]))')" >> $out; }
run 8 "default" X=1
run 8 "tile walk, 2048 waves" PSAMD_TILE=1 PSAMD_WAVES=2048
run 8 "tile walk, 1536 waves" PSAMD_TILE=1 PSAMD_WAVES=1536
run 4 "default" X=1
run 4 "tile walk, 2048 waves" PSAMD_TILE=1 PSAMD_WAVES=2048
run 4 "tile walk, 3072 waves" PSAMD_TILE=1 PSAMD_WAVES=3072
run 8 "default again" X=1
cat $out
