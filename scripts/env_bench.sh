#!/bin/bash
# scripts/env_bench.sh "ENV=.. ENV=.." ... [-- bench args]: the default bench under several environment settings, twice each
pick='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1].ljust(36), round(d["ms_per_step"],4), {k: round(v,1) for k,v in d["kernel_us_per_step"].items()})'
envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done; [ "$1" == "--" ] && shift
for i in 1 2; do for e in "${envs[@]}"; do
  env $e timeout -k 10 200 python bench.py --no-cpu "$@" 2>/dev/null | python -c "$pick" "$e" || exit 1
done; done
