#!/bin/bash
# the end-trace build beside the product's (same flags + -DPSAMD_END_TRACE on pairs.hip only), then the trace
set -e
cd particlesystem_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize -DPSAMD_END_TRACE -c csrc/pairs.hip -o /tmp/pairs_endtrace.o
objs="build/grid.hip.o build/apply.hip.o build/lifecycle.hip.o build/slab.hip.o build/capi.hip.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/pairs_endtrace.o -o libpsamd_endtrace.so
cd ..
