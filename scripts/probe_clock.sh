ls /sys/class/drm/ | head; for c in /sys/class/drm/card*/device; do echo $c; cat $c/pp_dpm_sclk 2>/dev/null | head -5; ls $c | grep -i -E "freq|clk|gpu_metrics|power" | head; done
(python bench.py --no-cpu --no-side-runs --steps 600 > /dev/null 2>&1 &) ; sleep 6
for i in 1 2 3; do cat /sys/class/drm/card*/device/pp_dpm_sclk | tr '\n' ' '; echo; sleep 0.3; done
rocm-smi --showclocks 2>&1 | head -20
rocm-smi --showpower 2>&1 | grep -i -E "power|W" | head -5
amd-smi metric -c 2>&1 | head -30
sleep 3
