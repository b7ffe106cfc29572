#!/bin/bash
# rocprofv3 PMC pass over the bench command (counters in their own run, with
# --kernel-trace only).  usage: bash scripts/pmc_bench.sh <tag> "<counters>" [bench args]
set -e
tag=$1; shift
ctrs=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_$tag
mkdir -p "$out"
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out" -o pmc -- python3 bench.py --no-cpu "$@" > "$out/stdout.json" 2> "$out/stderr.log"
ls "$out"
