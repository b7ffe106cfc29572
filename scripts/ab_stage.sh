#!/bin/bash
# A/B of bench.py with per-stage HIP-event times (--kernel-times).  usage (GPU box): bash scripts/ab_stage.sh <out-tag> "<label>|ENV=.." ...
tag=$1; shift
out=gpurun_out/ab_$tag.log
: > $out
for spec in "$@"; do
    label=${spec%%|*}; envs=${spec#*|}
    line=$(env $envs python bench.py --no-cpu --no-side-runs --kernel-times --steps 100 --warmup 5 2>/dev/null | tail -1)
    echo "$label $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); k=d["kernel_us_per_step"]; print("ms_per_step %.4f " % d["ms_per_step"] + " ".join("%s %.1f" % (a, b) for a, b in k.items()))')" >> $out
done
cat $out
