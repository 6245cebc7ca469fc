"""process_agc's two kernels (qh_qagc.hip): the regimes as instruction chains (the one that runs) against the whole state machine
stepped sample by sample -- bit for bit, state included -- and both against the restatement.  -m gpu."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def overloads(seed, nch, n, cpx):
    """low-passed noise with level steps: an overload ramp every few FIFO cycles, new maxima inside ramps, quiet stretches"""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((nch, n + 8))
    x = (x[:, :-8] + x[:, 1:-7] + x[:, 2:-6] + x[:, 3:-5] + x[:, 4:-4]) * 2.0 ** 20
    t = np.arange(n)
    x = x * (1.0 + 3.0 * ((t // 7000) % 3 == 1)) * np.where((t // 20000) % 4 == 3, 1e-3, 1.0)
    return x + 1j * (np.roll(x, 5, axis=1) if cpx else x)


def run(qh, form, x, rate, cpx, cuts, gains, two_buffers=False):
    import torch
    nch = x.shape[0]
    a = qh.QuiskAgc(nch, rate, is_cpx=cpx)
    a.debug_form(form)
    for c, g in enumerate(gains):
        a.set_agc(c, g)
    if not two_buffers:
        return np.concatenate([a.process_host(x[:, i:j]) for i, j in zip(cuts, cuts[1:])], axis=1)
    dev = torch.device("cuda:0")
    out = []
    for i, j in zip(cuts, cuts[1:]):
        src = torch.from_numpy(np.ascontiguousarray(x[:, i:j])).to(dev)
        dst = torch.full_like(src, 7.0)
        torch.cuda.synchronize()
        a.process2_ptr(src.data_ptr(), j - i, dst.data_ptr(), j - i, j - i)
        torch.cuda.synchronize()
        out.append(dst.cpu().numpy())
    return np.concatenate(out, axis=1)


@pytest.mark.parametrize("rate", [8000, 12000, 24000, 48000, 96000])
@pytest.mark.parametrize("cpx", [False, True])
def test_chain_kernel_is_the_state_machine_bit_for_bit(qh, rate, cpx):
    nch, n = 6, 60000
    x = overloads(rate + cpx, nch, n, cpx)
    cuts = [0, 100, 101, 164, 165, 4000, 4064, 4065, 30000, n]
    gains = [80.0, 300.0, 2000.0, 5000.0, 1e5, 1.0]
    y1 = run(qh, 1, x, rate, cpx, cuts, gains)
    y0 = run(qh, 0, x, rate, cpx, cuts, gains)
    assert np.abs(y1[:, 200:]).max() > 1e8                          # the AGC is at work
    assert np.array_equal(y0.view(np.float64), y1.view(np.float64))


def test_two_buffers_and_one(qh):
    nch, n = 4, 40000
    x = overloads(5, nch, n, False)
    cuts = [0, 64, 1000, 1001, 20000, n]
    gains = [80.0, 1000.0, 5000.0, 1e5]
    y_in = run(qh, 0, x, 48000, False, cuts, gains)
    y_two = run(qh, 0, x, 48000, False, cuts, gains, two_buffers=True)
    assert np.array_equal(y_two[:, :64], x[:, :64])                 # the first call only initialises: the samples pass
    assert np.array_equal(y_in.view(np.float64), y_two.view(np.float64))


@pytest.mark.parametrize("cpx", [False, True])
def test_chain_kernel_against_the_restatement(qh, oracle, cpx):
    nch, n = 3, 30000
    x = overloads(11, nch, n, cpx)
    cuts = [0, 128, 129, 5000, n]
    gains = [300.0, 5000.0, 1e5]
    y = run(qh, 0, x, 48000, cpx, cuts, gains)
    for c in range(nch):
        o = oracle.OracleQuiskAgc(48000)
        want = np.concatenate([o.process(x[c, i:j], cpx, gains[c]) for i, j in zip(cuts, cuts[1:])])
        if cpx:     # |z| is the device's hypot here, libm's cabs there
            assert np.abs(y[c] - want).max() <= 1e-12 * np.abs(want).max(), c
        else:
            assert np.array_equal(y[c].view(np.float64), want.view(np.float64)), c
