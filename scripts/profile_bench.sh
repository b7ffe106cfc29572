#!/bin/bash
# rocprofv3 capture of the bench command; summaries land in gpurun_out/prof_<tag>/
# usage (on the GPU box): bash scripts/profile_bench.sh <tag> [bench args...]
set -e
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o bench -- python3 bench.py --no-cpu "$@" > "$out/bench_stdout.json" 2> "$out/bench_stderr.log"
ls -R "$out" | head -30
