"""The one-receiver Quisk block API (C ABI group 9: qh_quisk_process_samples and the _quisk-named setters) against the
staged restatement of quisk_process_samples with process_agc on, and its get_graph against the graph restatement.
-m gpu."""
import numpy as np
import pytest

from conftest import rel_rms
from quisk_amd import rxfilter

pytestmark = pytest.mark.gpu


def test_process_samples_in_place_with_mode_and_tune_changes(qh, oracle):
    fs, blk = 192000, 4096
    api = qh.quiskapi
    api.open(fs, fft_size=2048, data_width=512)
    tabs = rxfilter.coefficient_tables()
    ref = oracle.OracleQuiskBlock(fs, 48000, tabs)          # quisk_process_samples restated as a whole: tune vector, Agc1 and
    g = oracle.OracleGraph(2048, 512, float(fs))            # the FFT ring carry on across mode changes
    ref.set_graph(g)
    t = np.arange(fs)
    x = 2.0 ** 22 * np.exp(2j * np.pi * ((10900.0 / fs) * t % 1.0)) + 2.0 ** 14 * (np.random.default_rng(0).standard_normal(fs) + 0j)
    plan = [(0, 3, 10000, "USB", 2700), (60000, 2, 11800, "LSB", 2400), (120000, 4, 10900, "AM", 6000)]
    outs, refs = [], []
    assert api.process_samples(np.zeros(4, complex), 0) == 0 and api.process_samples(np.zeros(4, complex), -3) == -3   # quisk.c:2336
    for k in range(0, fs - blk + 1, blk):
        for start, mode, tune, name, bw in plan:
            if k <= start < k + blk:
                api.set_rx_mode(mode); api.set_tune(tune)
                frate = rxfilter.get_filter_rate(fs, mode, bw)
                fI, fQ = rxfilter.make_filter_coef(frate, None, bw, rxfilter.get_filter_center(name, bw))
                api.set_filters(fI, fQ, bw)
                assert api.get_filter_rate() == frate
                # the reference keeps its static filter histories; here a mode change starts fresh filters (everything else carries on)
                ref.set_rx_mode(mode); ref.set_tune(tune); ref.set_filters(fI, fQ, bw)
                if start:
                    ref.restart_bank(0)
        buf = np.zeros(blk, dtype=np.complex128)
        buf[:] = x[k:k + blk]
        n = api.process_samples(buf, blk)
        outs.append(buf[:n].copy())
        refs.append(ref.process(x[k:k + blk]))
    y, want = np.concatenate(outs), np.concatenate(refs)
    assert y.size == want.size == (fs // blk) * blk // 4
    assert np.abs(want).max() > 2.0 ** 24
    assert rel_rms(y, want) < 1e-8
    pix, sm, cnt = api.get_graph(1.0, 0.0)
    rp, rs, rc = g.get(1.0, 0.0)
    assert cnt == rc and np.abs(pix - rp).max() < 1e-8 and abs(sm - rs) < 1e-8
    assert api.get_graph() is None
    api.close()


def _filters(mode_name, mode, bw, fs=192000):
    frate = rxfilter.get_filter_rate(fs, mode, bw)
    return rxfilter.make_filter_coef(frate, None, bw, rxfilter.get_filter_center(mode_name, bw))


def _two_tone(fs, n, f1, f2, seed):
    t = np.arange(n)
    rng = np.random.default_rng(seed)
    return (2.0 ** 22 * np.exp(2j * np.pi * ((f1 / fs) * t % 1.0)) + 2.0 ** 21 * np.exp(2j * np.pi * ((f2 / fs) * t % 1.0))
            + 2.0 ** 13 * (rng.standard_normal(n) + 1j * rng.standard_normal(n)))


@pytest.mark.parametrize("split", [1, 2, 3, 4])
def test_split_rxtx_second_bank_and_two_agcs(qh, oracle, split):
    """split Rx/Tx (quisk.c:2539-2590): the same samples demodulated a second time at quisk_tx_tune_freq + rit_freq on bank 1,
    the two audio streams on the real / imaginary output by split mode, one AGC each (quisk.c:2690-2698).  The oracle side is
    two staged receivers and two process_agc states, joined here the way the reference joins them."""
    fs, blk, nblk = 192000, 4096, 24
    api = qh.quiskapi
    tabs = rxfilter.coefficient_tables()
    x = _two_tone(fs, blk * nblk, 10900.0, 21900.0, 4)
    rx_tune, tx_tune = 10000, 21000
    fI, fQ = _filters("USB", 3, 2700)
    api.open(fs)
    api.set_rx_mode(3); api.set_tune2(rx_tune, tx_tune); api.set_filters(fI, fQ, 2700); api.set_split_rxtx(split)
    rs = []
    for tune in (rx_tune, tx_tune):
        r = oracle.OracleQuiskRx(fs, tabs)
        r.set_mode(3); r.set_tune(tune); r.set_filters(fI, fQ); r.set_bandwidth(2700)
        rs.append(r)
    agc = [oracle.OracleQuiskAgc(48000, 0.7, 1.0), oracle.OracleQuiskAgc(48000, 0.7, 1.0)]
    outs, refs = [], []
    for k in range(nblk):
        seg = x[k * blk:(k + 1) * blk]
        buf = seg.copy()
        n = api.process_samples(buf, blk)
        outs.append(buf[:n].copy())
        d, d2 = rs[0].process(seg).real, rs[1].process(seg).real
        if split == 1:
            re, im = (d, d2) if tx_tune < rx_tune else (d2, d)
        elif split == 2:
            re, im = (d, d2) if tx_tune >= rx_tune else (d2, d)
        elif split == 3:
            re, im = d, d
        else:
            re, im = d2, d2
        a = agc[0].process(re + 0j, False, 80.0).real
        b = agc[1].process(im + 0j, False, 80.0).real
        refs.append(a + 1j * b)
    api.close()
    y, want = np.concatenate(outs), np.concatenate(refs)
    assert y.size == want.size and np.abs(want.real).max() > 2.0 ** 24 and np.abs(want.imag).max() > 2.0 ** 24
    assert rel_rms(y, want) < 1e-8
    if split in (1, 2):
        assert rel_rms(y.real, y.imag) > 0.1                # two different receivers on the two channels


@pytest.mark.parametrize("method", [0, 1, 2])
def test_played_sub_receiver(qh, oracle, method):
    """multirx_play_channel (quisk.c:2592-2621): bank 1 demodulates ANOTHER receiver's samples with its own frequency, mode and
    nFilter-1 filter; play methods 0 (both), 1 (left = main), 2 (right = main)."""
    fs, blk, nblk = 192000, 4096, 20
    api = qh.quiskapi
    tabs = rxfilter.coefficient_tables()
    x = _two_tone(fs, blk * nblk, 10900.0, 40000.0, 5)
    xs = _two_tone(fs, blk * nblk, -15600.0, 30000.0, 6)
    fI, fQ = _filters("USB", 3, 2700)
    gI, gQ = _filters("LSB", 2, 2700)                       # as long as fI: set_filters keeps ONE sizeFilter for every filter set (quisk.c:4591)
    assert len(gI) == len(fI)
    api.open(fs)
    api.set_rx_mode(3); api.set_tune(10000); api.set_filters(fI, fQ, 2700)
    api.set_multirx_mode(1, 2); api.set_multirx_freq(1, -15000); api.set_filters2(gI, gQ, 2700)
    api.set_multirx_play_channel(1); api.set_multirx_play_method(method)
    r0 = oracle.OracleQuiskRx(fs, tabs); r0.set_mode(3); r0.set_tune(10000); r0.set_filters(fI, fQ); r0.set_bandwidth(2700)
    r1 = oracle.OracleQuiskRx(fs, tabs); r1.set_mode(2); r1.set_tune(-15000); r1.set_filters(gI, gQ); r1.set_bandwidth(2700)
    agc = [oracle.OracleQuiskAgc(48000, 0.7, 1.0), oracle.OracleQuiskAgc(48000, 0.7, 1.0)]
    outs, refs = [], []
    for k in range(nblk):
        seg, sub = x[k * blk:(k + 1) * blk], xs[k * blk:(k + 1) * blk]
        api.multirx_samples(0, sub)                         # not the played one: ignored
        api.multirx_samples(1, sub)
        buf = seg.copy()
        n = api.process_samples(buf, blk)
        outs.append(buf[:n].copy())
        d, d2 = r0.process(seg).real, r1.process(sub).real
        re, im = ((d2, d2), (d, d2), (d2, d))[method]
        refs.append(agc[0].process(re + 0j, False, 80.0).real + 1j * agc[1].process(im + 0j, False, 80.0).real)
    api.close()
    y, want = np.concatenate(outs), np.concatenate(refs)
    assert y.size == want.size and rel_rms(y, want) < 1e-8


def test_key_down_sidetone_silence_and_keyup_ramp(qh, oracle):
    """quisk.c:2368-2433,2729-2738: while the key is down the block is replaced (CW sidetone with a 5 ms rise, else silence) and
    nothing is demodulated; after key-up the sidetone falls in 5 ms, TxRxSilenceMsec of silence follows, then radio sound comes
    back under a 5 ms ramp; kill_audio zeroes the output."""
    fs, blk = 192000, 4096
    api = qh.quiskapi
    tabs = rxfilter.coefficient_tables()
    x = _two_tone(fs, blk * 40, 10900.0, 50000.0, 7)
    fI, fQ = _filters("CWU", 1, 1000)
    api.open(fs)
    api.set_rx_mode(1); api.set_tune(10000); api.set_filters(fI, fQ, 1000)
    api.set_sidetone(0.5, 600, 48000, 50)
    r = oracle.OracleQuiskRx(fs, tabs); r.set_mode(1); r.set_tune(10000); r.set_filters(fI, fQ); r.set_bandwidth(1000)
    agc = oracle.OracleQuiskAgc(48000, 0.7, 1.0)
    step = 1.0 / (48000 * 5e-3)
    phase = np.exp(2j * np.pi * 600 / 48000)
    # the reference's statics, restated
    st = dict(vec=2.2e9 + 0j, env=0.0, on=0, silence=0, keyup=1.0, cnt=0.0)
    outs, refs = [], []
    for k in range(40):
        down = 8 <= k < 14
        api.set_key_state(int(down), int(down), 2, 0)
        seg = x[k * blk:(k + 1) * blk]
        buf = seg.copy()
        n = api.process_samples(buf, blk)
        outs.append(buf[:n].copy())
        if down or st["on"] or st["silence"] > 0:
            st["cnt"] += blk * 48000 / fs
            nout = int(st["cnt"]); st["cnt"] -= nout
            o = np.zeros(nout, dtype=np.complex128)
            if down:
                st["silence"] = int(48000 * 1e-3 * 50); st["keyup"] = 0.0
                if not st["on"]:
                    st["on"], st["env"], st["vec"] = 1, 0.0, 2.2e9 + 0j
                for i in range(nout):
                    if st["env"] < 1.0:
                        st["env"] = min(1.0, st["env"] + step)
                    d = st["vec"].real * 0.5 * st["env"]
                    o[i] = d + 1j * d
                    st["vec"] *= phase
            elif st["on"]:
                i = 0
                while i < nout:
                    st["env"] -= step
                    if st["env"] < 0:
                        st["on"], st["env"] = 0, 0.0
                        break
                    d = st["vec"].real * 0.5 * st["env"]
                    o[i] = d + 1j * d
                    st["vec"] *= phase
                    i += 1
                st["silence"] -= nout - i
            else:
                st["silence"] -= nout
            refs.append(o)
            continue
        y = agc.process(r.process(seg), False, 80.0)
        if st["keyup"] < 1.0:
            for i in range(y.size):
                st["keyup"] += step
                if st["keyup"] > 1.0:
                    st["keyup"] = 1.0
                    break
                y[i] *= st["keyup"]
        refs.append(y)
    y, want = np.concatenate(outs), np.concatenate(refs)
    assert y.size == want.size
    assert np.abs(want[8 * 1024:14 * 1024]).max() > 1e9                 # the sidetone
    assert rel_rms(y, want) < 1e-8
    api.set_kill_audio(1)
    buf = x[:blk].copy()
    n = api.process_samples(buf, blk)
    assert n == 1024 and np.all(buf[:n] == 0)
    api.set_kill_audio(0)
    api.close()
