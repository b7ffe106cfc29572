#!/bin/bash
# scripts/simworld_try.sh W "ENV=.. ENV=.." ...   -- the --sim-world W projection under several environment settings
W=$1; shift
pick='import json,sys; d=json.loads(sys.stdin.read()); s=d["stage_ms_per_rank"]; print(sys.argv[1].ljust(40), "step", round(d["modelled_step_ms"],4), "pairs max", max(s["pairs"]), "mean", round(sum(s["pairs"])/len(s["pairs"]),4))'
for e in "$@"; do
  env $e timeout -k 10 200 python bench.py --sim-world $W --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "$pick" "$e" || exit 1
done
