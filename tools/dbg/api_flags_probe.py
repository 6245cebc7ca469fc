"""the one-receiver walk of tests/test_gpu_quisk_api_fuzz.py with the squelch flags of both sides printed call by call: <seed> <mode> <rate> <playback>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import quisk_amd as qh
import pyoracle as oracle
from test_gpu_quisk_api_fuzz import *
from test_gpu_quisk_api_fuzz import _draw, _filters, _signal
def walk(qh, oracle, seed, mode, fs, play):
    rng = np.random.default_rng(7000 + seed)
    api = qh.quiskapi
    api.open(fs, fft_size=2048, data_width=512, playback_rate=play)
    ref = oracle.OracleQuiskBlock(fs, play, rxfilter.coefficient_tables())
    graph = oracle.OracleGraph(2048, 512, float(fs))     # the panadapter's feed behind tone, inversion and blanker (quisk.c:2454-2475)
    ref.set_graph(graph)
    obs = np.random.default_rng(70000 + seed)            # (the observers draw from a generator of their own: the walks stay the walks they were)
    try:
        st = {"rx": 8300, "tx": 9100}
        fI, fQ = _filters(mode, fs)
        for o in (api, ref):
            o.set_rx_mode(mode); o.set_filters(fI, fQ, BW[mode]); o.set_agc(20.0)
        api.set_tune2(st["rx"], st["tx"]); ref.set_tune(st["rx"], st["tx"])
        api.set_sidetone(0.3, 600, play, 20); ref.set_sidetone(0.3, 600, 20)
        if mode == 3:
            gI, gQ = _filters(mode, fs, 2400)
            for o in (api, ref):
                o.set_filters(gI, gQ, 2400, 1); o.set_filters(fI, fQ, BW[mode])        # filter set 1, then the one global sizeFilter back (quisk.c:4591)
                o.set_multirx_mode(1, 3); o.set_multirx_freq(1, -15000); o.set_multirx_play_method(1)
        ratio = max(1, fs // 48000)
        sizes = [int(rng.choice([1, 2, 3, 5, 8])) * int(rng.integers(300, 1700)) * ratio for _ in range(24)]
        sizes = [min(s, 52000, 50000 * fs // play, 11000 * (fs // 48000 or 1)) for s in sizes]          # (the reference's interpolators stop at 52 800 outputs per call, its Buffer2Chan at 12 000 audio samples)
        n = sum(sizes)
        x = _signal(mode, 0, n, fs, float(st["rx"]), amp=2.0 ** 18)
        x[5000::9973] += 2.0 ** 21
        x[n // 2:n // 2 + n // 6] *= 0.01
        xs = _signal(mode, 1, n, fs, -15000.0, amp=2.0 ** 18) if mode == 3 else None
        log, pos, outs, loose_left = [], 0, 0, 0
        for k, s in enumerate(sizes):
            if k:
                for _ in range(int(rng.integers(1, 3))):
                    log.append((k, _draw(rng, mode, fs, play, api, ref, st)))
                    if mode in (5, 13) and log[-1][1][0] == "set_split_rxtx":
                        # the second FM receiver starts on an empty delay line: arg() of rounding-level numbers again, for as long as at the walk's
                        # own start (in output samples, not calls: a short call ends inside it)
                        loose_left = 6 * 1024 * (play // 48000) + 1024
            seg = x[pos:pos + s]
            if xs is not None:
                api.multirx_samples(1, xs[pos:pos + s]); ref.multirx_samples(1, xs[pos:pos + s])
            pos += s
            y, want = api.process(seg), ref.process(seg)
            assert y.size == want.size, (seed, k, y.size, want.size, log)
            print('call %d (%d samples): flags %d / %d; out %d; %r' % (k, s, api.squelch_flags(), ref.squelch_flags(), want.size, [l[1] for l in log if l[0] == k]), flush=True)
            if obs.integers(0, 4) == 0:                # get_graph (quisk.c:5142) now and then: the average starts over on both sides
                zoom, deltaf = float(obs.choice([1.0, 1.0, 2.0, 4.0])), float(obs.choice([0.0, 0.0, 5000.0, -12000.0]))
                got_g, want_g = api.get_graph(zoom, deltaf), graph.get(zoom, deltaf)
                assert (got_g is None) == (want_g is None), (seed, k)
                if got_g is not None:
                    assert got_g[2] == want_g[2], (seed, k, got_g[2], want_g[2])
                    assert np.abs(got_g[0] - want_g[0]).max() < 1e-6 and abs(got_g[1] - want_g[1]) < 1e-6, (seed, k, np.abs(got_g[0] - want_g[0]).max())
            if want.size == 0:
                continue
            settle = 6 * 1024 * (play // 48000) if mode in (5, 13) else 0                    # FM: arg() of rounding-level numbers while the filters fill
            lo = min(want.size, max(0, settle - outs))
            outs += want.size
            scale = max(np.abs(want).max(), 1.0)
            err = np.abs(y[lo:] - want[lo:]).max() / scale if want.size > lo else 0.0
            loose = loose_left > 0
            if np.abs(want).max() > 0.0:                 # (a key held down gives silence and stops the receivers: their run-in goes on afterwards)
                loose_left -= want.size
            assert err < (1e-4 if loose else 1e-6), "seed %d call %d (%d samples): max error %.2e of %.3e; setters %r" % (seed, k, s, err, scale, log)
    finally:
        api.close()



walk(qh, oracle, *[int(v) for v in sys.argv[1:5]])
