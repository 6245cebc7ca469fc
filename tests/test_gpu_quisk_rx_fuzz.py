"""Seeded random walks over the Quisk receiver bank's setters with ragged block lengths, against one oracle receiver per
channel: tune, Rx filter taps (a size change re-reads the reference's sample ring as a ring of the new size), AGC,
squelches, noise blanker and auto-notch all carry state across calls.  fp64 gate: 1e-6 relative RMS over the run; 1e-4
once process_agc has been on -- its overload ramp ends on a comparison that is exact in real arithmetic (quisk.c:2219,
2257), so the 1e-13 by which the FFT-based filters ahead of it differ from the reference's direct sums can move the end
of a ramp by one step, after which the two gains relax to the same target exponentially (seen: 2.8e-6 for three blocks, gone
the moment the AGC is switched off; 1.8e-5 decaying over a second in 1 of 140 walks).  -m gpu."""
import numpy as np
import pytest

from conftest import rel_rms
from quisk_amd import rxfilter, synth

pytestmark = pytest.mark.gpu

NAMES = {rxfilter.CWL: "CWL", rxfilter.CWU: "CWU", rxfilter.LSB: "LSB", rxfilter.USB: "USB", rxfilter.AM: "AM", rxfilter.FM: "FM"}


@pytest.mark.parametrize("fs,mode,seed", [(96000, rxfilter.USB, 1), (192000, rxfilter.LSB, 2), (48000, rxfilter.CWU, 3), (96000, rxfilter.AM, 4),
                                          (96000, rxfilter.FM, 5), (240000, rxfilter.USB, 6), (192000, rxfilter.AM, 7), (48000, rxfilter.USB, 8),
                                          (96000, rxfilter.USB, 9), (96000, rxfilter.LSB, 10), (192000, rxfilter.CWL, 11), (48000, rxfilter.AM, 12),
                                          (192000, rxfilter.FM, 13), (250000, rxfilter.LSB, 14), (96000, rxfilter.USB, 15), (96000, rxfilter.AM, 16)])
def test_random_walk(qh, oracle, fs, mode, seed):
    rng = np.random.default_rng(seed)
    nch = 2
    tabs = rxfilter.coefficient_tables()
    total = fs * 3 // 2
    t = np.arange(total)
    xs = []
    for c in range(nch):
        x = synth.impulsive_input(1, total, seed=seed * 10 + c, scale=2.0 ** 18)[0]
        # two carriers of clearly different strength (the auto-notch ranks spectral peaks: equal ones -- the two sidebands of
        # an AM signal, say -- are a tie that rounding decides); AM mode gets its modulation, FM a deviation
        car = 2.0 ** 24 * np.exp(2j * np.pi * ((6000.0 + 700.0 * c) / fs * t % 1.0))
        if mode == rxfilter.AM:
            car = car * (1.0 + 0.5 * np.cos(2 * np.pi * 431.0 * t / fs) + 0.2 * np.cos(2 * np.pi * 1013.0 * t / fs))
        elif mode == rxfilter.FM:
            car = car * np.exp(1j * (1.2 * np.sin(2 * np.pi * 431.0 * t / fs) + 0.4 * np.sin(2 * np.pi * 1013.0 * t / fs)))
        x += car
        if mode not in (rxfilter.AM, rxfilter.FM):
            x += 2.0 ** 22 * np.exp(2j * np.pi * ((7300.0 + 700.0 * c) / fs * t % 1.0))
        xs.append(x)
    x = np.stack(xs)
    bank = qh.QuiskRxBank(nch, fs, mode)
    refs = [oracle.OracleQuiskRx(fs, tabs) for _ in range(nch)]
    import os
    # QH_TWIN=1 (diagnostics): every restated receiver once more, fed 1e-13 of noise per sample -- how far the restatement is from itself
    twins = [oracle.OracleQuiskRx(fs, tabs) for _ in range(nch)] if os.environ.get("QH_TWIN") else []
    tws, pert = [[] for _ in range(nch)], np.random.default_rng(int(os.environ.get("QH_TWIN_SEED", "7")))

    def filt(bw):
        frate = rxfilter.get_filter_rate(fs, mode, bw)
        return rxfilter.make_filter_coef(frate, None, bw, rxfilter.get_filter_center(NAMES[mode], bw))

    skip = set(filter(None, os.environ.get("QH_SKIP", "").split(",")))         # diagnostics: these setters are left out on every side

    class _Skipping:
        def __init__(self, t): self._t = t
        def __getattr__(self, name):
            return (lambda *a: None) if name in skip else getattr(self._t, name)
    if skip:
        bank, refs, twins = _Skipping(bank), [_Skipping(r) for r in refs], [_Skipping(r) for r in twins]
    bw0 = {rxfilter.AM: 6000, rxfilter.FM: 12000, rxfilter.CWU: 500}.get(mode, 2700)
    fI, fQ = filt(bw0)
    for c, r in enumerate(refs):
        r.set_mode(mode); r.set_tune(5800 + 700 * c); r.set_filters(fI, fQ); r.set_bandwidth(bw0)
        bank.set_tune(c, 5800 + 700 * c)
    for c, r in enumerate(twins):
        r.set_mode(mode); r.set_tune(5800 + 700 * c); r.set_filters(fI, fQ); r.set_bandwidth(bw0)
    bank.set_filters(-1, fI, fQ)
    pos, ys, rs, log, blk, agc_used = 0, [], [[] for _ in range(nch)], [], 0, False
    while pos < total:
        n = int(min(total - pos, rng.integers(1, fs // 8)))
        if pos and rng.random() < 0.7:
            k = int(rng.integers(0, 7))
            if k == 0:
                c = int(rng.integers(0, nch)); f = int(rng.integers(4000, 9000))
                bank.set_tune(c, f); refs[c].set_tune(f); log.append((blk, "tune", c, f))
                if twins: twins[c].set_tune(f)
            elif k == 1:
                bw = int(rng.choice([1800, 2400, 2700, 3000])) if mode in (rxfilter.USB, rxfilter.LSB) else bw0
                a, b = filt(bw)
                bank.set_filters(-1, a, b)
                for r in refs + twins:
                    r.set_filters(a, b); r.set_bandwidth(bw)
                log.append((blk, "filters", bw))
            elif k == 2:
                on, g = bool(rng.integers(0, 2)), float(rng.uniform(5, 200))
                bank.set_agc(on, g)
                for r in refs + twins:
                    r.set_agc(on, g)
                log.append((blk, "agc", on, g))
                agc_used = agc_used or on
            elif k == 3:
                lvl = int(rng.integers(0, 4))
                bank.set_noise_blanker(lvl)
                for r in refs + twins:
                    r.set_noise_blanker(lvl)
                log.append((blk, "nb", lvl))
            elif k == 4:
                on = int(rng.integers(0, 2)); rit = 600 if mode == rxfilter.CWU else 0
                bank.set_auto_notch(on, rit)
                for r in refs + twins:
                    r.set_auto_notch(on, rit)
                log.append((blk, "notch", on))
            elif k == 5 and mode == rxfilter.FM:
                lvl = float(rng.uniform(-80, -5))
                bank.set_squelch(-1, lvl)
                for r in refs + twins:
                    r.set_squelch(lvl)
                log.append((blk, "squelch", lvl))
            elif k == 6 and mode != rxfilter.FM:
                on, lvl = int(rng.integers(0, 2)), int(rng.integers(1, 10))
                bank.set_ssb_squelch(on, lvl)
                for r in refs + twins:
                    r.set_ssb_squelch(on, lvl)
                log.append((blk, "ssb_squelch", on, lvl))
        ys.append(bank.process_host(x[:, pos:pos + n]))
        for c in range(nch):
            rs[c].append(refs[c].process(x[c, pos:pos + n]))
            if twins: tws[c].append(twins[c].process(x[c, pos:pos + n] * (1.0 + float(os.environ.get("QH_TWIN_EPS", "1e-13")) * pert.standard_normal(n))))
        pos += n
        blk += 1
    y = np.concatenate(ys, axis=1)
    for c in range(nch):
        ref = np.concatenate(rs[c])
        assert y[c].shape == ref.shape, (y[c].shape, ref.shape)
        assert np.abs(ref).max() > 0
        if mode == rxfilter.FM:
            # while the decimators fill, the discriminator takes the argument of numbers at rounding level (1e-13 of full
            # scale): both sides produce noise there, not the same noise
            y[c, :2500] = ref[:2500]
            for b in range(len(ys)):
                if sum(v.shape[1] for v in ys[:b]) < 2500:
                    k = max(0, min(ys[b].shape[1], 2500 - sum(v.shape[1] for v in ys[:b])))
                    ys[b][c, :k] = rs[c][b][:k]
        err = rel_rms(y[c], ref)
        if twins:
            print("seed %d channel %d: bank %.3e, the restatement against its twin %.3e" % (seed, c, err, rel_rms(np.concatenate(tws[c]), ref)), flush=True)
        tol = 1e-4 if agc_used else 1e-6
        if err >= tol:
            o = 0
            for b in range(len(ys)):
                m = ys[b].shape[1]
                if m and np.abs(ys[b][c] - rs[c][b]).max() > tol * np.abs(ref).max():
                    trace = ["%d:%.1e" % (bb, np.abs(ys[bb][c] - rs[c][bb]).max() / np.abs(ref).max()) for bb in range(b, min(b + 10, len(ys))) if ys[bb].shape[1]]
                    d = np.abs(ys[b][c] - rs[c][b]); k0 = int(np.argmax(d > tol * np.abs(ref).max()))
                    raise AssertionError("channel %d rel rms %.3e: first bad block %d (out %d..%d, first bad sample %d of %d); setters before it: %r; per-block max err %r" %
                                         (c, err, b, o, o + m, k0, m, [l for l in log if l[0] <= b][-8:], trace))
                o += m
        assert err < tol, (c, err)
