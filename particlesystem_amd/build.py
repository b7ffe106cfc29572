"""Build libpsamd.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpsamd.so")
# the step's kernels by stage (kernels_common.hpp holds the map), the host context + C ABI
SOURCES = ["grid.hip", "pairs.hip", "apply.hip", "lifecycle.hip", "slab.hip", "capi.hip"]
DEPS = SOURCES + ["kernels_common.hpp", "kernels.h", "device_types.h", "geometry.hpp", "partition.hpp", os.path.join("..", "..", "include", "psamd.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# -ffp-contract=off + correctly rounded fp32 divide/sqrt: the exact kernels must
# perform the reference's operations one rounding at a time.
# SLP vectorisation is off because the pair kernel packs its fp32 work by hand; the
# vectoriser's own packing cost it register shuffles (3.75 v_mov per pair).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-fno-fast-math", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]


STAMP = os.path.join(HERE, "build", "flags.stamp")


def _flags_stamp(extra=()):
    return " ".join(FLAGS + list(extra) + os.environ.get("PSAMD_EXTRA_FLAGS", "").split())


def needs_build(extra=()):
    """the library is older than a source, or was built with other flags (PSAMD_EXTRA_FLAGS: diagnostic builds)"""
    if not os.path.exists(LIB):
        return True
    try:
        with open(STAMP) as f:
            if f.read() != _flags_stamp(extra):
                return True
    except OSError:
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False, extra=()):
    stamp = _flags_stamp(extra)
    extra = list(extra) + os.environ.get("PSAMD_EXTRA_FLAGS", "").split()
    if not force and not needs_build(extra[:len(extra) - len(os.environ.get("PSAMD_EXTRA_FLAGS", "").split())]):
        return LIB
    tmp = LIB + ".tmp"
    # a failed build must not leave a stale library behind to be tested by mistake
    if os.path.exists(LIB):
        os.remove(LIB)
    # the translation units side by side (no device code calls across them), then one link
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src + ".o")
        cmd = [HIPCC] + FLAGS + list(extra) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        jobs.append((cmd, obj, subprocess.Popen(cmd)))
    failed = [cmd for cmd, _, p in jobs if p.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + [obj for _, obj, _ in jobs] + ["-o", tmp])
    os.replace(tmp, LIB)
    with open(STAMP, "w") as f:
        f.write(stamp)
    return LIB


DRIVER_SRC = os.path.join(os.path.dirname(HERE), "host", "ps_driver.cpp")
DRIVER = os.path.join(os.path.dirname(HERE), "host", "ps_driver")


def build_driver(force=False):
    """host/ps_driver: the reference's driver loop in C++ on top of the C ABI (plain g++,
    links libpsamd.so by relative rpath so the pair travels together)."""
    inc = os.path.join(os.path.dirname(HERE), "include")
    if not force and os.path.exists(DRIVER) and os.path.getmtime(DRIVER) > max(
            os.path.getmtime(DRIVER_SRC), os.path.getmtime(os.path.join(inc, "psamd.h")), os.path.getmtime(LIB)):
        return DRIVER
    tmp = DRIVER + ".tmp.%d" % os.getpid()
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-I" + inc, DRIVER_SRC, "-L" + HERE, "-lpsamd",
                           "-Wl,-rpath,$ORIGIN/../particlesystem_amd", "-Wl,-rpath-link,/opt/rocm/lib", "-o", tmp])
    os.replace(tmp, DRIVER)
    return DRIVER


RING_SRC = os.path.join(os.path.dirname(HERE), "host", "ps_ring_rccl.cpp")
RING = os.path.join(os.path.dirname(HERE), "host", "ps_ring_rccl")


def build_ring(force=False):
    """host/ps_ring_rccl: the multi-GPU step in C++ with RCCL moving the messages (hipcc: it uses
    the HIP runtime and rccl.h; links libpsamd.so by relative rpath and the image's librccl)."""
    inc = os.path.join(os.path.dirname(HERE), "include")
    if not force and os.path.exists(RING) and os.path.getmtime(RING) > max(
            os.path.getmtime(RING_SRC), os.path.getmtime(os.path.join(inc, "psamd.h")), os.path.getmtime(LIB)):
        return RING
    # linked beside its final name and renamed: another rank that sees the file sees a whole program
    tmp = RING + ".tmp.%d" % os.getpid()
    subprocess.check_call([HIPCC, "-std=c++17", "-O2", "-Wall", "-I" + inc, RING_SRC, "-L" + HERE, "-lpsamd", "-lrccl",
                           "-Wl,-rpath,$ORIGIN/../particlesystem_amd", "-o", tmp])
    os.replace(tmp, RING)
    return RING


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True,
          extra=["-Rpass-analysis=kernel-resource-usage"] if "--usage" in sys.argv else ())
    print(LIB)
    print(build_driver(force=True))
    print(build_ring(force=True))
