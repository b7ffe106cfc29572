#!/bin/bash
# A/B of the force pass's wave count / group size (round 3).  usage (on the GPU box): bash scripts/ab_r3.sh
out=gpurun_out/ab_r3.log
: > $out
run() {  # label, env..., -- bench args
    label=$1; shift
    envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    line=$(env "${envs[@]}" python bench.py --no-cpu --no-side-runs --steps 100 --warmup 5 "$@" 2>/dev/null | tail -1)
    echo "$label $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_per_step %.4f pairs_us %.1f frac %.4f collide %.1f apply %.1f lifecycle %.1f" % (d["ms_per_step"], d["kernel_us_per_step"]["pairs"], d["roofline"]["frac"], d["kernel_us_per_step"].get("collide",0), d["kernel_us_per_step"]["apply"], d["kernel_us_per_step"]["lifecycle"]))')" >> $out
}
W7=$PWD/particlesystem_amd/libpsamd_w7.so
run "exact  w6        " X=1 --
run "exact  w7        " PSAMD_LIB=$W7 PSAMD_WAVES_PER_SIMD=7 --
run "exact  w6 again  " X=1 --
run "fast   w6 nq8    " X=1 -- --fast-math
run "fast   w7 nq8    " PSAMD_LIB=$W7 PSAMD_WAVES_PER_SIMD=7 -- --fast-math
run "fast   w8 nq4    " PSAMD_LIB=$W7 PSAMD_WAVES_PER_SIMD=8 PSAMD_FAST_NQ=4 -- --fast-math
run "fast   w6 nq4    " PSAMD_FAST_NQ=4 -- --fast-math
cat $out
