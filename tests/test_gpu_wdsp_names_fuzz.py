"""Seeded walks through the WDSP NAMES -- OpenChannel / fexchange0 / the SetRXA* exports bound the way quisk_wdsp.py binds libwdsp
(ctypes: ints as int, floats as c_double) -- with the channel's geometry drawn too: in_size, dsp_size and the three rates that
OpenChannel takes (wdsp/channel.c:75-103), so the double rings of iobuffs.c:384-516 run at every ratio of in_size to dsp_insize, the
input and output resamplers come and go, and every setter lands between fexchange0 calls on a channel whose DSP blocks are replayed
from hipGraphs.  Against the oracle's fexchange0 (rings, latency, up-slew) with the same setters.  -m gpu."""
import ctypes as C

import numpy as np
import pytest

from conftest import rel_rms
from quisk_amd import synth
from test_gpu_rxa_fuzz import _Without, _apply, _apply2

pytestmark = pytest.mark.gpu


class _Names:
    """lib.<WDSP name>(channel, ...) with Python numbers turned into what the C prototypes take"""
    def __init__(self, lib, channel, dsp_size):
        self._lib, self._ch, self._dsp = lib, channel, dsp_size

    def __getattr__(self, name):
        if name.startswith("SetRXAEMNR"):                # (EMNR reads WDSP's data files when it is switched on through these names: INTEGRATION.md section 12)
            return lambda *args: None
        f = getattr(self._lib, name)

        def call(*args):
            if name == "RXASetNC":
                args = (max(args[0], self._dsp),)        # nc is a multiple of the block size (fircore's nfor = nc / size, firmin.c:296)
            f(self._ch, *[C.c_double(a) if isinstance(a, float) else C.c_int(int(a)) for a in args])
            assert self._lib.qh_wdsp_status() == 0, (name, args, self._lib.qh_last_error())
        return call


class _NoEmnr:
    def __init__(self, o, dsp_size):
        self._o, self._dsp = o, dsp_size

    def __getattr__(self, name):
        if name.startswith("SetRXAEMNR"):
            return lambda *args: None
        if name == "RXASetNC":
            return lambda nc: self._o.RXASetNC(max(nc, self._dsp))
        return getattr(self._o, name)


GEOMETRY = [  # in_size, dsp_size, in_rate, dsp_rate, out_rate
    (1024, 256, 192000, 48000, 48000), (256, 256, 48000, 48000, 48000), (64, 256, 192000, 48000, 48000), (4096, 256, 192000, 48000, 48000),
    (512, 128, 96000, 48000, 48000), (2048, 512, 192000, 48000, 96000), (256, 64, 48000, 48000, 24000), (1024, 1024, 96000, 48000, 48000),
    (192, 64, 144000, 48000, 48000), (128, 256, 48000, 48000, 96000),
]


@pytest.mark.parametrize("seed", list(range(1, 21)) + list(range(5001, 5013)) + list(range(6001, 6009)))
def test_random_walk_through_the_wdsp_names(qh, oracle, seed):
    """(seeds above 5000: every other draw from the second menu -- the notch database's edits, its filter's window / auto-increase /
    edges / shift, SetRXAANFVals / ANRVals, SetRXABandpassRun, SetRXAPanelGain2 -- through the names' own wrappers)"""
    lib = qh.load()
    rng = np.random.default_rng(31000 + seed)
    notches = [0]
    in_size, dsp_size, in_rate, dsp_rate, out_rate = GEOMETRY[(seed - 1) % len(GEOMETRY)]
    ch = 16 + seed % 8
    D = C.c_double
    lib.OpenChannel(ch, in_size, dsp_size, in_rate, dsp_rate, out_rate, 0, 1, D(0.010), D(0.025), D(0.0), D(0.010), 1)
    assert lib.qh_wdsp_status() == 0, lib.qh_last_error()
    o = _NoEmnr(oracle.WdspChannel(in_size, dsp_size, in_rate, dsp_rate, out_rate), dsp_size)
    names = _Names(lib, ch, dsp_size)
    launches0 = lib.qh_wdsp_graph_launches()
    try:
        for t in (names, o):
            t.SetRXAShiftRun(1); t.SetRXAShiftFreq(float(synth.shift_freq(seed % 4))); t.RXANBPSetRun(1); t.SetRXAMode(1)
            t.RXASetPassband(300.0, 3000.0); t.SetRXAAGCMode(0)
        out_size = o.out_size
        nblk = 60 * max(1, 1024 // in_size)
        x = synth.make_input_numpy(4, nblk * in_size * 192000 // in_rate)[seed % 4][::192000 // in_rate][:nblk * in_size].copy()
        err = C.c_int(0)
        got, want, lms, log, per = [], [], False, [], []
        b = 0
        while b < nblk:
            if b:
                for _ in range(int(rng.integers(0, 2))):
                    tg = [(names, ()), (o, ())]
                    if seed > 6000:                       # (from 6000 up the FM detector too; the AM detector forced on beside it is refused)
                        tg = [(_Without(t, "SetRXAAMDRun"), lead) for t, lead in tg]
                    if seed > 5000 and rng.integers(0, 2):
                        done = _apply2(rng, tg, notches, fm=seed > 6000)
                    else:
                        done = _apply(rng, tg)
                        notches[0] += sum(1 for d in done if d[0] == "RXANBPAddNotch")
                    lms = lms or any(d[0] in ("SetRXAANFRun", "SetRXAANRRun") and d[1] for d in done)
                    log.append((b, done))
            n = min(nblk - b, int(rng.integers(1, 5)) * max(1, 1024 // in_size))
            seg = np.ascontiguousarray(x[b * in_size:(b + n) * in_size])
            y = np.zeros(n * out_size, dtype=np.complex128)
            for k in range(n):
                blk = np.ascontiguousarray(seg[k * in_size:(k + 1) * in_size])
                lib.fexchange0(ch, blk.ctypes.data_as(C.c_void_p), y[k * out_size:].ctypes.data_as(C.c_void_p), C.byref(err))
                assert err.value == 0 and lib.qh_wdsp_status() == 0, (seed, b + k, err.value, lib.qh_last_error())
            r, nerr = o.fexchange0(seg)
            assert nerr == 0
            got.append(y); want.append(r)
            per.append("%d:%.1e/%.1e" % (b, float(np.abs(y - r).max()), float(np.abs(r).max())))
            b += n
        y, r = np.concatenate(got), np.concatenate(want)
        assert np.all(np.isfinite(r))
        if np.abs(r).max() > 1e-9:                       # (a squelch or a panel setting may keep the channel quiet for the whole walk)
            assert rel_rms(y, r) < (1e-4 if lms else 1e-6), (seed, (in_size, dsp_size, in_rate, dsp_rate, out_rate), rel_rms(y, r), log,
                                                              "first block of the call: largest error / largest reference sample", per)
        if in_rate // dsp_rate in (1, 2, 4, 8, 16) and out_rate == dsp_rate:
            assert lib.qh_wdsp_graph_launches() > launches0      # (a resampler at either end keeps host-side state: plain launches)
    finally:
        lib.CloseChannel(ch)
