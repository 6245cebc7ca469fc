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
    r = oracle.OracleQuiskRx(fs, tabs)
    agc = oracle.OracleQuiskAgc(48000, 0.7, 1.0)            # Agc1 is static: one AGC for every mode (quisk.c:2321)
    g = oracle.OracleGraph(2048, 512, float(fs))
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
                # the reference keeps its static filter histories; here a mode change starts fresh filters (the AGC carries on)
                r = oracle.OracleQuiskRx(fs, tabs)
                r.set_mode(mode); r.set_tune(tune); r.set_filters(fI, fQ); r.set_bandwidth(bw)
        buf = np.zeros(blk, dtype=np.complex128)
        buf[:] = x[k:k + blk]
        n = api.process_samples(buf, blk)
        outs.append(buf[:n].copy())
        refs.append(agc.process(r.process(x[k:k + blk]), False, 80.0))
        g.feed(x[k:k + blk])
    y, want = np.concatenate(outs), np.concatenate(refs)
    assert y.size == want.size == (fs // blk) * blk // 4
    assert np.abs(want).max() > 2.0 ** 24
    assert rel_rms(y, want) < 1e-8
    pix, sm, cnt = api.get_graph(1.0, 0.0)
    rp, rs, rc = g.get(1.0, 0.0)
    assert cnt == rc and np.abs(pix - rp).max() < 1e-8 and abs(sm - rs) < 1e-8
    assert api.get_graph() is None
    api.close()
