"""Builds quisk_amd/lib/libquiskhip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libquiskhip.so")
SOURCES = ["qh_engine.hip", "qh_fir.hip", "qh_pan.hip", "qh_qrx.hip", "qh_hbcascade.hip", "qh_polyphase.hip", "qh_ingest.hip", "qh_qagc.hip", "qh_nb.hip", "qh_analyzer.hip", "qh_wdsp_compat.cpp", "qh_quisk_compat.cpp", "qh_quisk_rx_compat.cpp", "qh_qps.hip", "qh_design.cpp"]
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


def _export_map(path, extra=()):
    """Linker version script: the C ABI of include/quiskhip.h and nothing else.  The library is loaded into other programs' processes
    (Quisk's Python, anything that links filter.o's names): kernels' host stubs, C++ helpers and file-scope state stay local."""
    import re
    src = open(os.path.join(HERE, "..", "include", "quiskhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(n for n in re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", src) if n != "defined"))
    text = "{\n  global:\n" + "".join("    %s;\n" % n for n in names + list(extra)) + "  local: *;\n};\n"
    try:
        if open(path).read() == text:
            return path
    except OSError:
        pass
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(text)
    return path


def _deps(src):
    """Headers a source includes (transitively, by name), so that a header edit rebuilds only the units that see it."""
    seen, todo = set(), [os.path.join(CSRC, src)]
    while todo:
        f = todo.pop()
        try:
            text = open(f).read()
        except OSError:
            continue
        for line in text.splitlines():
            line = line.strip()
            if line.startswith('#include "'):
                h = os.path.normpath(os.path.join(os.path.dirname(f), line.split('"')[1]))
                if h not in seen:
                    seen.add(h)
                    todo.append(h)
    return seen


def build(force=False, verbose=False, defines=(), out=None, jobs=None):
    """One object per source under quisk_amd/lib/obj (rebuilt when the source or a header it includes is newer), then the
    link.  defines/out: experiment builds (tools/ab_bench.py) with -D overrides into another file, compiled in one go;
    QH_HIPCC_FLAGS in the environment adds compiler flags to such builds (scheduler strategies and the like)."""
    target = out or LIB
    base = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-x", "hip"]
    extra = ["-D" + d for d in defines] + os.environ.get("QH_HIPCC_FLAGS", "").split()
    if extra and not out:
        # (the shipped library is kept when it is newer than its sources, whatever it was compiled with: an experiment build that
        # overwrote or silently reused it would measure the wrong thing)
        raise ValueError("quisk_amd.build: -D overrides / QH_HIPCC_FLAGS make an experiment build: pass out=<another file> (tools/ab_bench.py)")
    os.makedirs(os.path.dirname(target), exist_ok=True)
    if out:
        vmap = _export_map(os.path.join(os.path.dirname(target), "quiskhip.map"), extra=["qh_dbg_*"])      # (experiment builds may add probes)
        cmd = base + ["-shared", "-Wl,--version-script=" + vmap, "-o", target] + extra + [os.path.join(CSRC, f) for f in SOURCES]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        return target
    objdir = os.path.join(LIBDIR, "obj")
    stamp_file = os.path.join(objdir, "build.stamp")
    if not force and not needs_build():          # the library that travelled with the snapshot is current: nothing to do, nothing started
        return LIB
    # Something will be compiled: what were the objects that are kept compiled with?  Another compiler, ROCm release or flag set makes
    # all of them stale.  (Asked only here -- a process that merely loads a current library must not start a compiler driver.)
    import hashlib
    try:
        ver = subprocess.run([base[0], "--version"], capture_output=True, text=True).stdout
    except OSError:
        ver = ""
    stamp = hashlib.sha256(("\n".join(base[1:] + extra) + "\n" + ver).encode()).hexdigest()
    try:
        if open(stamp_file).read().strip() != stamp:
            force = True
    except OSError:
        if os.path.isdir(objdir):
            force = True
    os.makedirs(objdir, exist_ok=True)
    base = base + extra
    todo, objs = [], []
    for src in SOURCES:
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        stale = force or not os.path.exists(obj)
        if not stale:
            t = os.path.getmtime(obj)
            stale = any(os.path.getmtime(f) > t for f in [os.path.join(CSRC, src)] + [d for d in _deps(src) if os.path.exists(d)])
        if stale:
            todo.append((src, obj))
    hdr = os.path.join(HERE, "..", "include", "quiskhip.h")
    if not todo and os.path.exists(LIB) and all(os.path.getmtime(o) <= os.path.getmtime(LIB) for o in objs + [hdr]):
        return LIB
    jobs = jobs or min(4, os.cpu_count() or 1)
    running = []

    def reap(block):
        for pr, src in list(running):
            if block or pr.poll() is not None:
                if pr.wait() != 0:
                    for other, _ in running:
                        if other is not pr:
                            other.kill()
                    raise subprocess.CalledProcessError(pr.returncode, src)
                running.remove((pr, src))
                if block:
                    return

    for src, obj in todo:
        while len(running) >= jobs:
            reap(True)
        cmd = base + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        running.append((subprocess.Popen(cmd), src))
    while running:
        reap(True)
    vmap = _export_map(os.path.join(objdir, "quiskhip.map"))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + vmap, "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    with open(stamp_file, "w") as fh:
        fh.write(stamp + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
