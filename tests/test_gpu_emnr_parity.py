"""xemnr, WDSP's spectral noise reduction (wdsp/emnr.c), in the RXA engine against the restatement (oracle/emnr_oracle.c):
every gain method, both noise estimators, the post-filter on and off, both chain positions.  The input is speech-like
(syllable-rate bursts of a few carriers over steady noise), long enough for the minimum-statistics window (1.5 s) to
turn over.  fp64 gate 1e-6 relative RMS.  -m gpu."""
import numpy as np
import pytest

from conftest import rel_rms
from quisk_amd import synth

pytestmark = pytest.mark.gpu


def speechy(c, n, fs=192000.0):
    rng = np.random.default_rng(500 + c)
    t = np.arange(n) / fs
    env = np.clip(np.sin(2 * np.pi * (3.1 + 0.4 * c) * t), 0.0, 1.0) ** 2 * (0.6 + 0.4 * np.sin(2 * np.pi * 0.7 * t))
    f0 = -synth.shift_freq(c)
    x = sum(a * np.exp(2j * np.pi * (f0 - f) * t * 1.0) for a, f in ((0.06, 700.0), (0.04, 1210.0), (0.03, 1900.0), (0.02, 2500.0)))
    return x * env + 0.004 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))


@pytest.mark.parametrize("gain_method,npe,ae,position", [(2, 0, 1, 0), (3, 0, 1, 0), (0, 0, 0, 0), (1, 1, 1, 0), (2, 0, 1, 1), (3, 1, 0, 1),
                                                         (2, 2, 1, 0), (3, 2, 1, 1)])
def test_emnr_matches_oracle(qh, oracle, gain_method, npe, ae, position):
    nch, nblk = 2, 700                       # 3.7 s at 192 k
    x = np.stack([speechy(c, nblk * 1024) for c in range(nch)])
    e = qh.RxaEngine(nch)
    e.load_emnr_tables()
    refs = []
    for ch in range(nch):
        o = oracle.WdspChannel(1024, 256, 192000, 48000, 48000)
        for t, a in ((e, (ch,)), (o, ())):
            t.SetRXAShiftRun(*a, 1); t.SetRXAShiftFreq(*a, synth.shift_freq(ch)); t.RXANBPSetRun(*a, 1)
            t.SetRXAMode(*a, 1); t.RXASetPassband(*a, 300.0, 3000.0)
            t.SetRXAAGCMode(*a, 0 if ch == 0 else 3); t.SetRXAAGCFixed(*a, 6.0)
            t.SetRXAEMNRgainMethod(*a, gain_method); t.SetRXAEMNRnpeMethod(*a, npe); t.SetRXAEMNRaeRun(*a, ae)
            t.SetRXAEMNRPosition(*a, position); t.SetRXAEMNRRun(*a, 1)
            if ch == 1:     # the four scalar knobs of emnr.c:1145-1174
                t.SetRXAEMNRaeZetaThresh(*a, 0.6); t.SetRXAEMNRaePsi(*a, 12.0); t.SetRXAEMNRtrainZetaThresh(*a, -1.0); t.SetRXAEMNRtrainT2(*a, 0.3)
        refs.append(o)
    ys, rs = [], [[] for _ in range(nch)]
    for a, b in ((0, 3), (3, 4), (4, 301), (301, nblk)):        # ragged calls: frames straddle them
        ys.append(e.process_host(x[:, a * 1024:b * 1024]))
        for ch in range(nch):
            rs[ch].append(refs[ch].xrxa(x[ch, a * 1024:b * 1024]))
    y = np.concatenate(ys, axis=1)
    for ch in range(nch):
        ref = np.concatenate(rs[ch])
        assert np.all(np.isfinite(ref)) and np.abs(ref[-48000:]).max() > 1e-3
        assert rel_rms(y[ch], ref) < 1e-6, (ch, rel_rms(y[ch], ref))


def test_emnr_needs_its_tables_and_switches_mid_stream(qh, oracle):
    e = qh.RxaEngine(1)
    with pytest.raises(qh.QuiskHipError):
        e.SetRXAEMNRRun(0, 1)                                   # no tables yet: refused, nothing silently skipped
    e.load_emnr_tables()
    o = oracle.WdspChannel(1024, 256, 192000, 48000, 48000)
    x = speechy(0, 400 * 1024)
    for t, a in ((e, (0,)), (o, ())):
        t.SetRXAShiftRun(*a, 1); t.SetRXAShiftFreq(*a, synth.shift_freq(0)); t.RXANBPSetRun(*a, 1); t.SetRXAMode(*a, 1)
        t.RXASetPassband(*a, 300.0, 3000.0); t.SetRXAAGCMode(*a, 0); t.SetRXAAGCFixed(*a, 0.0)
    ys, rs = [], []
    for (a, b), run, method in (((0, 60), 0, 2), ((60, 200), 1, 2), ((200, 290), 1, 3), ((290, 330), 0, 3), ((330, 400), 1, 2)):
        for t, lead in ((e, (0,)), (o, ())):
            t.SetRXAEMNRgainMethod(*lead, method); t.SetRXAEMNRRun(*lead, run)
        ys.append(e.process_host(x[None, a * 1024:b * 1024])[0]); rs.append(o.xrxa(x[a * 1024:b * 1024]))
    y, r = np.concatenate(ys), np.concatenate(rs)
    assert rel_rms(y, r) < 1e-6
    # and the noise reduction is really in the path: without it the same chain gives something else
    p = oracle.WdspChannel(1024, 256, 192000, 48000, 48000)
    p.SetRXAShiftRun(1); p.SetRXAShiftFreq(synth.shift_freq(0)); p.RXANBPSetRun(1); p.SetRXAMode(1); p.RXASetPassband(300.0, 3000.0)
    p.SetRXAAGCMode(0); p.SetRXAAGCFixed(0.0)
    plain = p.xrxa(x)
    assert rel_rms(r[100 * 256:190 * 256], plain[100 * 256:190 * 256]) > 0.1
    assert rel_rms(r[:60 * 256], plain[:60 * 256]) < 1e-12


def _emnr_names_run(lib, ch, x, nb, method=2):
    import ctypes as C
    D = C.c_double
    lib.OpenChannel(ch, 1024, 256, 192000, 48000, 48000, 0, 1, D(0.010), D(0.025), D(0.0), D(0.010), 1)
    lib.SetRXAShiftRun(ch, 1); lib.SetRXAShiftFreq(ch, D(synth.shift_freq(0))); lib.RXANBPSetRun(ch, 1); lib.SetRXAMode(ch, 1)
    lib.RXASetPassband(ch, D(300.0), D(3000.0)); lib.SetRXAAGCMode(ch, 0); lib.SetRXAAGCFixed(ch, D(0.0))
    lib.SetRXAEMNRgainMethod(ch, method)
    lib.SetRXAEMNRRun(ch, 1)
    assert lib.qh_wdsp_status() == 0, lib.qh_last_error()
    out = np.zeros(nb * 256, dtype=np.complex128)
    err = C.c_int(0)
    for b in range(nb):
        blk = np.ascontiguousarray(x[b * 1024:(b + 1) * 1024])
        lib.fexchange0(ch, blk.ctypes.data_as(C.c_void_p), out[b * 256:].ctypes.data_as(C.c_void_p), C.byref(err))
        assert err.value == 0
    lib.CloseChannel(ch)
    return out


def _emnr_oracle_run(oracle, x, method=2, tables=None):
    o = oracle.WdspChannel(1024, 256, 192000, 48000, 48000)
    if tables is not None:
        o.set_emnr_tables(tables)
    o.SetRXAShiftRun(1); o.SetRXAShiftFreq(synth.shift_freq(0)); o.RXANBPSetRun(1); o.SetRXAMode(1); o.RXASetPassband(300.0, 3000.0)
    o.SetRXAAGCMode(0); o.SetRXAAGCFixed(0.0); o.SetRXAEMNRgainMethod(method); o.SetRXAEMNRRun(1)
    ref, errs = o.fexchange0(x)
    assert errs == 0
    return ref


def test_emnr_through_the_wdsp_names_from_an_empty_directory(qh, oracle, tmp_path, monkeypatch):
    """Quisk's case (quisk.py:6017-6027 from quisk/, the data files in quisk/wdsp/): no `calculus`, no `zetaHat.bin`, no environment --
    WDSP takes the tables compiled into it (emnr.c:207-225, 322-326) and so does the library (qh_emnr_tables.hpp)."""
    lib = qh.load()
    monkeypatch.delenv("QH_WDSP_DATA", raising=False)
    monkeypatch.chdir(tmp_path)
    assert not list(tmp_path.iterdir())
    nb = 300
    x = speechy(0, nb * 1024)
    for method, ch in ((2, 14), (3, 15)):                       # the two gain methods Quisk selects
        out = _emnr_names_run(lib, ch, x, nb, method)
        assert rel_rms(out, _emnr_oracle_run(oracle, x, method)) < 1e-6


def test_emnr_through_the_wdsp_names_reads_wdsps_data_files(qh, oracle, tmp_path, monkeypatch):
    """... and a file that IS there wins, each on its own like in the reference: a `calculus` with other numbers in the working directory
    (GG scaled: the gain of method 2 follows), zetaHat still the built-in one; then the same file through $QH_WDSP_DATA."""
    t = dict(oracle.emnr_tables())
    t["GG"] = t["GG"] * 0.5
    (tmp_path / "calculus").write_bytes(t["GG"].tobytes() + t["GGS"].tobytes())
    lib = qh.load()
    nb = 200
    x = speechy(0, nb * 1024)
    ref_default = _emnr_oracle_run(oracle, x, 2)
    ref_scaled = _emnr_oracle_run(oracle, x, 2, tables=t)
    assert rel_rms(ref_scaled, ref_default) > 1e-2
    monkeypatch.delenv("QH_WDSP_DATA", raising=False)
    monkeypatch.chdir(tmp_path)
    assert rel_rms(_emnr_names_run(lib, 14, x, nb), ref_scaled) < 1e-6
    monkeypatch.chdir(tmp_path.parent)
    assert rel_rms(_emnr_names_run(lib, 14, x, nb), ref_default) < 1e-6
    monkeypatch.setenv("QH_WDSP_DATA", str(tmp_path))
    assert rel_rms(_emnr_names_run(lib, 14, x, nb), ref_scaled) < 1e-6
