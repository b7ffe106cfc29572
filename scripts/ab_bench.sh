#!/bin/bash
# A/B of two builds of libpsamd.so on the GPU box: scripts/ab_bench.sh <alt.so> [bench args]
alt=$1; shift
pick='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["ms_per_step"],4), {k: round(v,1) for k,v in d["kernel_us_per_step"].items()})'
for i in 1 2; do
  timeout -k 10 200 python bench.py --no-cpu "$@" 2>/dev/null | python -c "$pick" main || exit 1
  PSAMD_LIB=$alt timeout -k 10 200 python bench.py --no-cpu "$@" 2>/dev/null | python -c "$pick" alt || exit 1
done
