"""Seeded random walks over the setters of the batched quisk_process_samples (qh_qps_*, include/quiskhip.h group 9b) between ragged
calls, the same calls made on one block-level restatement per receiver (oracle qo_ps_*, quisk.c:2289-2742): what is carried from
call to call -- filter histories through changes of the Rx filter's length, cFracDecim's and the interpolator's phases, the
NoiseBlanker's and the auto-notch's state, process_agc's machine, the squelches' averages, the tune vectors' phase through changes
of frequency -- is where a batched form goes wrong first (tests/test_gpu_rxa_fuzz.py found such a bug in the RXA engine).
Mode and rates are fixed per walk (a change of mode rebuilds the bank: a stated deviation).  -m gpu."""
import os

import numpy as np
import pytest

from conftest import rel_rms
from quisk_amd import rxfilter
from test_gpu_quisk_process_bank import BW, NAMES, _filters, _refs, _signal

pytestmark = pytest.mark.gpu
NCH = 4


def _draw(rng, mode, fs, bank, refs):
    k = int(rng.integers(0, 11))
    done = None
    if k == 0:
        c, f = int(rng.integers(0, NCH)), int(rng.integers(-30000, 30000))
        bank.set_tune(c, f); refs[c].set_tune(f); done = ("set_tune", c, f)
    elif k == 1:                                       # another Rx filter for one receiver: bandwidth, and with it sometimes the length
        c = int(rng.integers(0, NCH))
        bw = int(rng.choice({3: [1800, 2400, 2700, 3000], 2: [1800, 2400, 2700, 3000], 4: [4000, 6000, 8000], 5: [10000, 12000, 16000], 13: [10000, 12000, 16000],
                             1: [200, 500, 1000], 0: [200, 500, 1000]}.get(mode, [BW[mode]])))
        frate = rxfilter.get_filter_rate(fs, mode, BW[mode])
        fI, fQ = rxfilter.make_filter_coef(frate, int(rng.choice([0, 0, 193, 325, 1025])) or None, bw, rxfilter.get_filter_center(NAMES[mode], bw))
        bank.set_filters(c, fI, fQ); refs[c].set_filters(fI, fQ, BW[mode]); done = ("set_filters", c, bw, len(fI))
    elif k == 2:
        lvl = float(rng.choice([5.0, 20.0, 60.0, 150.0]))       # below the limiter: an overload ramp's end is chaotic (tests/test_gpu_bench_shapes.py)
        bank.set_agc(lvl); [r.set_agc(lvl) for r in refs]; done = ("set_agc", lvl)
    elif k == 3:
        # (FM walks leave the blanker off: behind a blanked stretch the discriminator takes arg() of rounding-level numbers, as at start-up)
        lvl = int(rng.integers(0, 4)) if mode not in (5, 13) else 0
        bank.set_noise_blanker(lvl); [r.set_noise_blanker(lvl) for r in refs]; done = ("set_noise_blanker", lvl)
    elif k == 4:
        on = int(rng.integers(0, 2))
        bank.set_auto_notch(on); [r.set_auto_notch(on) for r in refs]; done = ("set_auto_notch", on)
    elif k == 5:
        inv = int(rng.integers(0, 2))
        bank.invert_spectrum(inv); [r.invert_spectrum(inv) for r in refs]; done = ("invert_spectrum", inv)
    elif k == 6:
        kill = int(rng.integers(0, 4) == 0)
        bank.set_kill_audio(kill); [r.set_kill_audio(kill) for r in refs]; done = ("set_kill_audio", kill)
    elif k == 7:
        # (the test tone is -40 dB of full scale whatever the signal, quisk.c:1263: in a passband, times the AGC's starting gain of 100, it
        # is an overload -- FM walks only, where the audio's level does not follow the input's)
        f = int(rng.choice([0, 0, 7000, -12000, 21000])) if mode in (5, 13) else 0
        bank.add_tone(f); [r.add_tone(f) for r in refs]; done = ("add_tone", f)
    elif k == 8:
        c, lvl = int(rng.integers(0, NCH)), float(rng.uniform(-90.0, -30.0))
        bank.set_squelch(c, lvl); refs[c].set_squelch(lvl); done = ("set_squelch", c, lvl)
    elif k == 9:
        en, lvl = int(rng.integers(0, 2)), int(rng.integers(1, 10))
        bank.set_ssb_squelch(en, lvl); [r.set_ssb_squelch(en, lvl) for r in refs]; done = ("set_ssb_squelch", en, lvl)
    else:
        p = int(rng.choice([0, 1, 2, 3, 8]))
        bank.set_pieces(p); done = ("set_pieces", p)
    return done


class _Both:
    """A restated receiver and its twin as one: a setter goes to both."""
    def __init__(self, a, b): self._a, self._b = a, b
    def __getattr__(self, name):
        fa, fb = getattr(self._a, name), getattr(self._b, name)
        def call(*args):
            fb(*args)
            return fa(*args)
        return call


@pytest.mark.parametrize("seed,mode,fs,play", [(1, 3, 192000, 48000), (2, 3, 111111, 96000), (3, 4, 96000, 48000), (4, 5, 192000, 48000),
                                               (5, 3, 48000, 48000), (6, 1, 133333, 48000), (7, 4, 185185, 96000), (8, 5, 96000, 192000),
                                               (9, 3, 192000, 192000), (10, 3, 370370, 48000),
                                               # found by tools/dbg/bank_fuzz_sweep.py (a one-off script, in git history): a squelch switched on in a call that is cut into pieces
                                               (146, 4, 185185, 96000), (101, 5, 192000, 48000), (106, 3, 192000, 192000),
                                               # the FM squelch's measuring windows while the bank cuts its calls into pieces (threshold set later)
                                               (7122, 5, 192000, 48000),
                                               # CWL, LSB, DGT-U, DGT-L, DGT-IQ (stereo: process_agc on the complex magnitude), DGT-FM, IMD
                                               (21, 0, 96000, 48000), (22, 2, 192000, 48000), (23, 7, 192000, 96000), (24, 8, 111111, 48000), (417, 9, 192000, 48000),
                                               (26, 13, 96000, 48000), (27, 10, 48000, 48000),
                                               # a sweep of round 6: two set_filters of different lengths ahead of ONE block (the second found indexFilter
                                               # beyond the first one's size and took it for a write position: 3e-2 off on that receiver)
                                               (920306, 5, 192000, 48000)])
def test_random_setter_walk_over_the_bank(qh, oracle, seed, mode, fs, play):
    rng = np.random.default_rng(9000 + seed)
    tunes = [7000 + 1300 * c for c in range(NCH)]
    filt = [_filters(mode, fs)] * NCH
    bank = qh.QuiskProcessBank(NCH, fs, mode, BW[mode], playback_rate=play, fft_size=2048, data_width=512)
    refs = _refs(oracle, NCH, fs, play, mode, tunes, filt)
    graphs = []
    for c in range(NCH):
        bank.set_tune(c, tunes[c]); bank.set_filters(c, *filt[c])
        graphs.append(oracle.OracleGraph(2048, 512, float(fs)))      # the panadapter's feed: the samples behind tone, inversion and blanker (quisk.c:2454-2475)
        refs[c].set_graph(graphs[c])
    bank.set_agc(20.0); [r.set_agc(20.0) for r in refs]
    ratio = max(1, fs // 48000)
    sizes = [int(rng.choice([1, 2, 3, 5, 8])) * int(rng.integers(300, 1700)) * ratio // 1 for _ in range(22)]
    sizes = [min(s, 52000, 50000 * fs // play) for s in sizes]          # the reference's interpolators stop at 52 800 outputs per call
    n = sum(sizes)
    # (levels that keep process_agc under its limiter even at the gain of 100 it starts from, quisk.c:2182: the end of an overload ramp is
    # a chaotic function of the input, tests/test_gpu_bench_shapes.py -- the machine itself is held bit for bit elsewhere)
    x = np.stack([_signal(mode, c, n, fs, float(tunes[c]), amp=2.0 ** 18) for c in range(NCH)])
    x[:, 5000::9973] += 2.0 ** 21                                                 # impulses for the blanker
    x[:, n // 2:n // 2 + n // 6] *= 0.01                                          # a fade (squelches, AGC release)
    # QH_TWIN=1 (diagnostics): every restated receiver once more, fed the input with 1e-13 of relative noise -- how far the restatement is
    # from itself on this walk (a setter goes to both: _Both)
    twins = _refs(oracle, NCH, fs, play, mode, tunes, filt) if os.environ.get("QH_TWIN") else []
    for t in twins:
        t.set_agc(20.0)
    pert = np.random.default_rng(7)
    both = [_Both(refs[c], twins[c]) for c in range(NCH)] if twins else refs
    log, pos, outs = [], 0, 0
    for k, s in enumerate(sizes):
        if k:
            for _ in range(int(rng.integers(1, 3))):
                log.append((k, _draw(rng, mode, fs, bank, both)))
        seg = x[:, pos:pos + s]
        pos += s
        y = bank.process_host(seg)
        for c in range(NCH):
            want = refs[c].process(seg[c])
            assert y[c].size == want.size, (seed, k, c, y[c].size, want.size, log)
            tw = twins[c].process(seg[c] * (1.0 + 1e-13 * pert.standard_normal(seg[c].size))) if twins else None
            if want.size == 0:
                continue
            settle = 6 * 1024 * (play // 48000) if mode in (5, 13) else 0                    # FM: arg() of rounding-level numbers while the filters fill
            lo = min(want.size, max(0, settle - outs))
            scale = max(np.abs(want).max(), 1.0)
            err = np.abs(y[c][lo:] - want[lo:]).max() / scale if want.size > lo else 0.0
            terr = "" if tw is None or tw.size != want.size else "; the restatement's twin: %.2e" % (np.abs(tw[lo:] - want[lo:]).max() / scale if want.size > lo else 0.0)
            assert err < 1e-6, "seed %d call %d (%d samples) receiver %d: max error %.2e of %.3e%s; setters %r" % (seed, k, s, c, err, scale, terr, log)
        outs += y.shape[1]
        if rng.integers(0, 4) == 0:                  # get_graph (quisk.c:5142) now and then: the average starts over on both sides
            zoom, deltaf = float(rng.choice([1.0, 1.0, 2.0, 4.0])), float(rng.choice([0.0, 0.0, 5000.0, -12000.0]))
            got = bank.get_graph(zoom, deltaf)
            for c in range(NCH):
                want_g = graphs[c].get(zoom, deltaf)
                assert (got is None) == (want_g is None), (seed, k, c)        # no whole block of 2048 samples yet: None on both sides
                if got is None:
                    continue
                rp, rs, rc = want_g
                pix, sm, cnt = got
                assert cnt == rc, (seed, k, c, cnt, rc)
                assert np.abs(pix[c] - rp).max() < 1e-6 and abs(sm[c] - rs) < 1e-6, (seed, k, c, np.abs(pix[c] - rp).max(), sm[c], rs)
    bank.close()
