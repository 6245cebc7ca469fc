"""The C-ABI library loads without a GPU, exports every symbol include/quiskhip.h declares, and refuses to
compute (loudly) when no HIP device is present.  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "quiskhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", src)
    return sorted(set(n for n in names if n not in ("defined",)))


def test_header_symbols_are_exported(qh):
    lib = qh.load()
    names = declared_functions()
    assert len(names) >= 45
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_nothing_but_the_header_is_exported(qh):
    """The library is loaded into other programs' processes (Quisk's Python process binds it as wdsp/libwdsp.so, C programs link
    filter.o's names): what `nm -D` shows must be include/quiskhip.h and nothing else -- no kernel stubs, no C++ helpers, no
    file-scope state (round 4 leaked g_shim, g_disp, shim_side, a kernel stub ...).  quisk_amd/build.py links with an export
    map made from the header."""
    import shutil
    import subprocess
    nm = shutil.which("nm") or shutil.which("llvm-nm") or ("/opt/rocm/lib/llvm/bin/llvm-nm" if os.path.exists("/opt/rocm/lib/llvm/bin/llvm-nm") else None)
    if not nm:
        pytest.skip("no nm")
    out = subprocess.run([nm, "-D", "--defined-only", qh.load()._name], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(line.split()[-1].split("@")[0] for line in out.splitlines() if line.strip()))
    extra = [n for n in exported if n not in set(declared_functions())]
    assert not extra, extra[:20]


def test_no_device_is_an_error_not_a_fallback(qh):
    lib = qh.load()
    if lib.qh_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(qh.QuiskHipError) as e:
        qh.RxaEngine(1)
    assert "no HIP device" in str(e.value) or "CPU fallback" in str(e.value)
    # the WDSP-named layer reports the same condition through qh_wdsp_status()
    lib.OpenChannel.argtypes = [ctypes.c_int] * 8 + [ctypes.c_double] * 4 + [ctypes.c_int]
    lib.OpenChannel(1, 256, 256, 48000, 48000, 48000, 0, 1, 0.010, 0.025, 0.0, 0.010, 1)
    assert lib.qh_wdsp_status() != 0


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "quisk_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.replace("the CPU oracle", ""), os.path.join(dirpath, f)


def test_header_is_plain_c_and_links_from_c(qh, tmp_path):
    """include/quiskhip.h is C99 (no C++ or torch types in any signature) and a C program links against the library."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    lib = qh.load()._name
    src = tmp_path / "t.c"
    src.write_text('#include "quiskhip.h"\n'
                   'int main(void) { qh_iq_format f; double t[43]; qh_iq_format_le24(&f, 1.0); qh_hb45_taps(t);\n'
                   '  return (GetWDSPVersion() == 125 && f.sample_bytes == 3 && t[21] == 0.5) ? 0 : 1; }\n')
    inc = os.path.join(ROOT, "include")
    exe = tmp_path / "t"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, str(src), "-o", str(exe),
                    "-L", os.path.dirname(lib), "-lquiskhip", "-Wl,-rpath," + os.path.dirname(lib)], check=True)
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-x", "c++", "-fsyntax-only", str(src)], check=True)
    assert subprocess.run([str(exe)]).returncode == 0
