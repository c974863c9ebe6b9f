#!/usr/bin/env python3
"""Build a TIMING VARIANT of the library: a copy of comfyui-seedvr2_amd/csrc with one of tools/timing_variants/*.patch
applied and the variant's macro set, compiled into build/timing/<name>/libseedvr2_hip.so.

These libraries return WRONG results by construction (a K loop without its loads, a conv without its epilogue ...): they
exist to measure what a piece of a kernel costs.  The product sources contain none of this code, `__graft_entry__.build()`
refuses sources that contain the macros, and the product only ever loads the library next to hip.py unless the
A/B switch SVR2_LIB names another file (hip.lib() then says so on stderr).

    python tools/build_timing_variant.py gemm4 SVR2_G4_DBG=1      # no DMA in the K loop
    python tools/build_timing_variant.py conv_halo SVR2_HALO_DBG=1 # conv_halo_wr without epilogue
    python tools/build_timing_variant.py conv_halo_gnfuse SVR2_HALO_GNFUSE=1   # + the LDS/VALU work of a fused GroupNorm+SiLU
    SVR2_LIB=build/timing/gemm4_SVR2_G4_DBG_1/libseedvr2_hip.so python tools/kernel_bench.py gemm
"""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(variant: str, macro: str) -> str:
    import __graft_entry__ as ge
    tag = f"{variant}_{macro.replace('=', '_')}"
    work = os.path.join(ROOT, "build", "timing", tag)
    shutil.rmtree(work, ignore_errors=True)
    shutil.copytree(os.path.join(ge.PKG_DIR, "csrc"), os.path.join(work, "csrc"))
    shutil.rmtree(os.path.join(work, "..", "include"), ignore_errors=True)       # common.h includes ../../include/seedvr2_hip.h
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(work, "..", "include"))
    patch = os.path.join(ROOT, "tools", "timing_variants", f"{variant}_dbg.patch")
    with open(patch) as fh:
        target = fh.readline().split()[1]          # "--- conv_halo.hip <date>": the source the patch applies to
    subprocess.check_call(["patch", "-s", os.path.join(work, "csrc", os.path.basename(target)), patch])
    objs = []
    for s in sorted(glob.glob(os.path.join(work, "csrc", "*.hip"))):
        o = os.path.join(work, os.path.basename(s) + ".o")
        cmd = ge.compile_command(s, o) + [f"-D{macro}"]
        ge.run_compile(cmd)
        objs.append(o)
    out = os.path.join(work, "libseedvr2_hip.so")
    subprocess.check_call([ge.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    print(out)
    return out


if __name__ == "__main__":
    build(sys.argv[1], sys.argv[2])
