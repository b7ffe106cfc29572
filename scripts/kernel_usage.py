#!/usr/bin/env python3
"""Register / LDS / occupancy table of every kernel in kernels.hip (compiles to /tmp, no GPU needed)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "particlesystem_amd", "csrc", "kernels.hip")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
       "-fno-fast-math", "-fno-slp-vectorize", "-c", src, "-o", "/tmp/psamd_usage.o", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for l in err.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
    for k, pat in (("v", r" VGPRs: (\d+)"), ("s", r"TotalSGPRs: (\d+)"), ("scr", r"ScratchSize \[bytes/lane\]: (\d+)"),
                   ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, l)
        if m and cur is not None:
            cur[k] = m.group(1)
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("void psamd::", "").replace("psamd::", "")
    print("%-40s vgpr=%-4s sgpr=%-4s scratch=%-4s occ=%-2s lds=%s" % (n[:40], r.get("v"), r.get("s"), r.get("scr"), r.get("occ"), r.get("lds")))
if "error" in err:
    print(err[-3000:])
