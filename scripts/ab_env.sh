#!/bin/bash
# Generic A/B of bench.py under environment switches.  usage (GPU box): bash scripts/ab_env.sh <out-tag> "<label>|ENV=.. ENV2=.." ...
tag=$1; shift
out=gpurun_out/ab_$tag.log
: > $out
for spec in "$@"; do
    label=${spec%%|*}; envs=${spec#*|}
    line=$(env $envs python bench.py --no-cpu --no-side-runs --steps 100 --warmup 5 2>/dev/null | tail -1)
    echo "$label $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); k=d["kernel_us_per_step"]; print("ms_per_step %.4f pairs %.1f flags+plan %.1f apply %.1f lifecycle %.1f" % (d["ms_per_step"], k["pairs"], k.get("collide",0), k["apply"], k["lifecycle"]))')" >> $out
done
cat $out
