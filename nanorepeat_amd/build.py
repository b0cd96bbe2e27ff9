"""Build nanorepeat_amd/libnanorepeat_amd.so (HIP kernels + C ABI) in-tree for gfx950.

hipcc cross-compiles without a GPU.  The kernel file is split into parts
(-DNRA_PART=1..27) that compile in parallel; the shared library carries only gfx950 code.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libnanorepeat_amd.so")
OBJ = os.path.join(HERE, "csrc", "build")

SOURCES = ["nra_kernels.hip", "nra_sweep.hip", "nra_joint.hip", "nra_trace.hip", "nra_host.cpp", "nra_internal.h", "nra_device.h", "nra_pk16.h"]
ARCH = "gfx950"


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(INCLUDE, "nanorepeat_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, jobs=None, verbose=False):
    """Compile and link the shared library; returns its path."""
    if not force and not _stale():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    common = [hipcc, f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-I", INCLUDE, "-I", CSRC,
              "-Wno-unused-command-line-argument"]
    cmds = []
    objs = []
    for part in (1, 2, 3, 4):
        o = os.path.join(OBJ, f"nra_kernels_p{part}.o")
        objs.append(o)
        cmds.append(common + [f"-DNRA_PART={part}", "-c", os.path.join(CSRC, "nra_kernels.hip"), "-o", o])
    for part in (5, 6, 11, 12, 13, 14, 15, 16, 18, 19, 25, 26):
        o = os.path.join(OBJ, f"nra_sweep_p{part}.o")
        objs.append(o)
        cmds.append(common + [f"-DNRA_PART={part}", "-c", os.path.join(CSRC, "nra_sweep.hip"), "-o", o])
    for part in (7, 8, 10, 17, 20, 21, 22, 23):
        o = os.path.join(OBJ, f"nra_joint_p{part}.o")
        objs.append(o)
        cmds.append(common + [f"-DNRA_PART={part}", "-c", os.path.join(CSRC, "nra_joint.hip"), "-o", o])
    o = os.path.join(OBJ, "nra_trace_p9.o")
    objs.append(o)
    cmds.append(common + ["-DNRA_PART=9", "-c", os.path.join(CSRC, "nra_trace.hip"), "-o", o])
    o = os.path.join(OBJ, "nra_host.o")
    objs.append(o)
    cmds.append(common + ["-pthread", "-c", os.path.join(CSRC, "nra_host.cpp"), "-o", o])

    # an object is rebuilt when its source, a header next to it or the public header is newer
    headers = [os.path.join(CSRC, h) for h in SOURCES if h.endswith(".h")] + [os.path.join(INCLUDE, "nanorepeat_amd.h")]

    def fresh(cmd):
        obj, src = cmd[-1], cmd[-3]
        if force or not os.path.exists(obj):
            return False
        t = os.path.getmtime(obj)
        return all(os.path.getmtime(d) <= t for d in [src] + headers)

    cmds = [c for c in cmds if not fresh(c)]

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)

    jobs = jobs or max(1, min(len(cmds), max(1, (os.cpu_count() or 2) - 1)))
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(run, cmds))
    run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-pthread", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
