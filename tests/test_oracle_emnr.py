"""oracle/emnr_oracle.c (WDSP's EMNR, wdsp/emnr.c) against what the algorithm must do by construction.  PARITY UNPINNED by
reference execution (wdsp needs <fftw3.h>); these pin the restatement's framing, tables and estimators."""
import ctypes as C

import numpy as np


def _emnr(oracle, bsize=256, rate=48000, tables=None):
    L = oracle.lib()
    L.wo_emnr_create.restype = C.c_void_p
    L.wo_emnr_create.argtypes = [C.c_int, C.c_int]
    L.wo_emnr_free.argtypes = [C.c_void_p]
    L.wo_emnr_set_tables.argtypes = [C.c_void_p] * 5 + [C.c_double] * 4
    L.wo_emnr_run.restype = C.POINTER(C.c_int)
    L.wo_emnr_run.argtypes = [C.c_void_p]
    L.wo_emnr_exec.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.wo_emnr_set_gain_method.argtypes = [C.c_void_p, C.c_int]
    L.wo_emnr_set_npe_method.argtypes = [C.c_void_p, C.c_int]
    L.wo_emnr_set_ae_run.argtypes = [C.c_void_p, C.c_int]
    h = L.wo_emnr_create(bsize, rate)
    t = tables or oracle.emnr_tables()
    L.wo_emnr_set_tables(h, t["GG"].ctypes.data, t["GGS"].ctypes.data, t["zeta_hat"].ctypes.data, t["zeta_valid"].ctypes.data,
                         *[float(v) for v in t["zeta_range"]])
    L.wo_emnr_run(h)[0] = 1
    return L, h, t


def _run(L, h, x, bsize=256):
    y = np.zeros_like(x)
    for b in range(x.size // bsize):
        buf = np.zeros(2 * bsize)
        buf[0::2] = x[b * bsize:(b + 1) * bsize]
        buf[1::2] = 7.0                                    # the imaginary part is ignored and comes out zero
        L.wo_emnr_exec(h, 0, buf.ctypes.data)
        assert not np.any(buf[1::2])
        y[b * bsize:(b + 1) * bsize] = buf[0::2]
    return y


def test_unit_gain_tables_make_the_stft_an_identity_with_its_delay(oracle):
    # gain method 2 multiplies two table look-ups: tables of ones give mask == 1 in every bin; the sqrt-Hamming analysis and
    # synthesis windows multiply to a Hamming window whose four overlapped copies add up to 2.16, so the chain is a pure delay of
    # 4096 - 256 samples (calc_emnr's init_oainidx) times c = 2.16 / 4 * (N / sum(sqrt-Hamming))^2 (calc_window's normalisation)
    t = dict(oracle.emnr_tables())
    t["GG"] = np.ones(241 * 241); t["GGS"] = np.ones(241 * 241)
    L, h, keep = _emnr(oracle, tables=t)
    L.wo_emnr_set_gain_method(h, 2); L.wo_emnr_set_ae_run(h, 0)
    x = np.random.default_rng(0).standard_normal(256 * 80)
    y = _run(L, h, x)
    d = 4096 - 256
    w = np.sqrt(0.54 - 0.46 * np.cos(2 * np.pi * np.arange(4096) / 4096))
    c = 2.16 / 4 * (4096 / w.sum()) ** 2
    assert not np.any(y[:d])
    assert np.abs(y[d + 4096:] - c * x[4096:-d]).max() < 1e-12          # (the first frames see an input ring that is still filling)
    L.wo_emnr_free(h)


def test_tables_are_wdsps(oracle):
    t = oracle.emnr_tables()
    assert t["GG"].size == 241 * 241 and t["GGS"].size == 241 * 241 and t["zeta_hat"].size == 3600
    assert abs(t["GG"][0] - 7.25654181154076983e-01) < 1e-15 and abs(t["GG"][1] - 7.05038822098223439e-01) < 1e-15     # calculus.c:2
    assert list(t["zeta_dims"]) == [60, 60]


def test_noise_is_reduced_and_a_burst_comes_through(oracle):
    rng = np.random.default_rng(1)
    n = 256 * 1200                                         # 6.4 s at 48 k
    noise = 0.01 * rng.standard_normal(n)
    burst = np.zeros(n)
    k = np.arange(n)
    on = (k % 24000) < 6000                                # 125 ms of tone every half second
    burst[on] = 0.2 * np.cos(2 * np.pi * 1000 / 48000 * k[on])
    for method, npe in ((2, 0), (0, 0), (1, 1), (3, 2)):
        L, h, keep = _emnr(oracle)
        L.wo_emnr_set_gain_method(h, method); L.wo_emnr_set_npe_method(h, npe)
        y = _run(L, h, noise + burst)
        tail = slice(n - 48000 * 2, n)
        d = 4096 - 256
        quiet = ~np.roll(on, d) & ~np.roll(on, d + 2048) & ~np.roll(on, d - 2048)
        loud = np.roll(on, d)
        nz = np.sqrt(np.mean(y[tail][quiet[tail]] ** 2))
        sg = np.sqrt(np.mean(y[tail][loud[tail]] ** 2))
        assert nz < 0.5 * 0.01, (method, nz)               # the noise floor between bursts is down by more than 6 dB
        assert sg > 0.05, (method, sg)                      # the bursts are still there
        L.wo_emnr_free(h)
