"""The decimating FIR that shares the panadapter's read (qh_pan_attach_fir / qh_pan_feed_decimate, BASELINE config 3's shape:
1023-tap / 32 beside the 16384-point panadapter): the FIR out of the panadapter's own unwindowed transform (fold + 512-point
inverse + wrap repair), the Hanning window applied in the frequency domain.  Against the two separate engines, and against the
oracle's quisk_cDecimate / get_graph restatements.  -m gpu."""
import numpy as np
import pytest

from conftest import rel_rms

pytestmark = pytest.mark.gpu


def _taps(ntaps=1023):
    k = np.arange(ntaps) - (ntaps - 1) / 2.0
    return np.sinc(k / 32.0) / 32.0 * np.blackman(ntaps)


@pytest.mark.parametrize("ntaps", [1023, 200, 1024])
def test_fused_fir_and_panadapter_match_the_separate_engines_and_the_oracle(qh, oracle, ntaps):
    import torch
    dev = torch.device("cuda", 0)
    nch, N, fs = 3, 16384, 1536000.0
    calls = [2, 1, 4]                                    # blocks per call: the FIR state carries over
    n = sum(calls) * N
    rng = np.random.default_rng(5)
    t = np.arange(n)
    x = np.stack([2.0 ** 24 * np.exp(2j * np.pi * (((3000.0 + 7000.0 * c) / fs) * t % 1.0)) +
                  2.0 ** 18 * (rng.standard_normal(n) + 1j * rng.standard_normal(n)) for c in range(nch)])
    taps = _taps(ntaps)
    xd = torch.from_numpy(x).to(dev)
    s = torch.cuda.current_stream(dev).cuda_stream
    fused = qh.Panadapter(nch, N, 1024, fs, stream=s)
    fused.attach_fir(taps, 32)
    sep_pan = qh.Panadapter(nch, N, 1024, fs, stream=s)
    bank = qh.FirBank(nch, taps, 32, stream=s)
    y = torch.zeros((nch, n // 32), dtype=torch.complex128, device=dev)
    yb = torch.zeros_like(y)
    pos = 0
    for nb in calls:
        m = nb * N
        got = fused.feed_decimate_ptr(xd.data_ptr() + 16 * pos, n, m, y.data_ptr() + 16 * (pos // 32), n // 32)
        assert got == m // 32
        bank.process_ptr(xd.data_ptr() + 16 * pos, n, m, yb.data_ptr() + 16 * (pos // 32), n // 32)
        sep_pan.feed_ptr(xd.data_ptr() + 16 * pos, n, m)
        pos += m
    torch.cuda.synchronize()
    yh, ybh = y.cpu().numpy(), yb.cpu().numpy()
    for c in range(nch):
        assert rel_rms(yh[c], ybh[c]) < 1e-12, (c, rel_rms(yh[c], ybh[c]))
    o = oracle.OracleFir(taps)
    want = o.cDecimate(x[0], 32)
    assert want.size == n // 32 and rel_rms(yh[0], want) < 1e-12
    pf, sf, cf = fused.get_graph(1.0, 0.0)
    ps, ss, cs = sep_pan.get_graph(1.0, 0.0)
    assert cf == cs == sum(calls)
    assert np.abs(pf - ps).max() < 1e-8 and np.abs(sf - ss).max() < 1e-8
    g = oracle.OracleGraph(N, 1024, fs)
    g.feed(x[0])
    rp, rs, rc = g.get(1.0, 0.0)
    assert rc == sum(calls) and np.abs(pf[0] - rp).max() < 1e-8


def test_shapes_outside_the_fused_form_are_refused(qh):
    p = qh.Panadapter(2, 4096, 512, 192000.0)
    with pytest.raises(qh.QuiskHipError):
        p.attach_fir(_taps(255), 32)                     # fft_size 4096
    p = qh.Panadapter(2, 16384, 512, 192000.0)
    with pytest.raises(qh.QuiskHipError):
        p.attach_fir(_taps(255), 16)
    with pytest.raises(qh.QuiskHipError):
        p.attach_fir(_taps(1500), 32)
