"""Builds quisk_amd/lib/libquiskhip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libquiskhip.so")
SOURCES = ["qh_engine.hip", "qh_fir.hip", "qh_pan.hip", "qh_qrx.hip", "qh_hbcascade.hip", "qh_polyphase.hip", "qh_ingest.hip", "qh_qagc.hip", "qh_nb.hip", "qh_analyzer.hip", "qh_wdsp_compat.cpp", "qh_quisk_compat.cpp", "qh_quisk_rx_compat.cpp", "qh_design.cpp"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))) + [os.path.join("..", "..", "include", "quiskhip.h")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False, defines=(), out=None):
    """defines/out: experiment builds (tools/ab_bench.py) with -D overrides into another file; QH_HIPCC_FLAGS in the environment adds
    compiler flags to such builds (scheduler strategies and the like)."""
    target = out or LIB
    if not force and not out and not needs_build():
        return LIB
    os.makedirs(os.path.dirname(target), exist_ok=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
           "-x", "hip", "-o", target] + ["-D" + d for d in defines] + os.environ.get("QH_HIPCC_FLAGS", "").split() + \
          [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return target


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
