"""The three xrxa meters (adc, S, agc: wdsp/RXA.c:566,569,589; xmeter wdsp/meter.c:75-108) fused into the nbp0 launch of
the linear fast path (qh_osfir.hpp METER, meter_finish_kernel) against the oracle's sample-by-sample xmeter, against the
stand-alone meter kernel of the per-mode path, and the audio against an engine that runs without meters.  -m gpu."""
import numpy as np
import pytest

from conftest import rel_rms
from quisk_amd import synth

pytestmark = pytest.mark.gpu

# mlog10 truncates the mantissa to 11 bits (wdsp/meterlog10.c): one step is 10*log10(2)/2048 dB
STEP_DB = 10.0 * np.log10(2.0) / 2048.0


def _setup(e, nch, agc_db, nc=None):
    for ch in range(nch):
        e.SetRXAShiftRun(ch, 1)
        e.SetRXAShiftFreq(ch, synth.shift_freq(ch))
        e.RXANBPSetRun(ch, 1)
        e.SetRXAMode(ch, 1)
        e.RXASetPassband(ch, 300.0, 3000.0)
        if nc:
            e.RXASetNC(ch, nc)
        e.SetRXAAGCMode(ch, 0)
        e.SetRXAAGCFixed(ch, agc_db + ch)


def _oracle(po, ch, in_size, dsp_size, agc_db, nc=None):
    o = po.WdspChannel(in_size, dsp_size, 192000, 48000, 48000)
    o.SetRXAShiftRun(1)
    o.SetRXAShiftFreq(synth.shift_freq(ch))
    o.RXANBPSetRun(1)
    o.SetRXAMode(1)
    o.RXASetPassband(300.0, 3000.0)
    if nc:
        o.RXASetNC(nc)
    o.SetRXAAGCMode(0)
    o.SetRXAAGCFixed(agc_db + ch)
    return o


@pytest.mark.parametrize("dsp_size,nc", [(256, None), (256, 256), (64, 512), (128, 1024), (512, None), (1024, 1024)])
def test_fused_meters_match_the_oracle(qh, oracle, dsp_size, nc):
    nch = 3
    in_size = 4 * dsp_size
    calls = [8, 3, 21, 1, 40] if dsp_size <= 256 else [4, 1, 7]       # ragged against the 2048-sample tile
    total = sum(calls)
    x = synth.make_input_numpy(nch, total * in_size)
    e = qh.RxaEngine(nch, dsp_size=dsp_size)
    p = qh.RxaEngine(nch, dsp_size=dsp_size)          # no meters: plain fast path
    _setup(e, nch, 6.0, nc)
    _setup(p, nch, 6.0, nc)
    e.enable_meters(True)
    refs = [_oracle(oracle, ch, in_size, dsp_size, 6.0, nc) for ch in range(nch)]
    pos = 0
    for k, nb in enumerate(calls):
        seg = np.ascontiguousarray(x[:, pos:pos + nb * in_size])
        pos += nb * in_size
        y = e.process_host(seg)
        yp = p.process_host(seg)
        # the meter launch starts its tiles one sample earlier than the plain one: same audio up to FFT rounding
        assert rel_rms(y, yp) < 1e-12
        for ch in range(nch):
            want = refs[ch].xrxa(seg[ch])
            assert rel_rms(y[ch], want) < 1e-9
            for mt in (0, 1, 2, 3, 5, 6):
                got, ref = e.GetRXAMeter(ch, mt), refs[ch].GetRXAMeter(mt)
                assert abs(got - ref) < 1.01 * STEP_DB, (k, ch, mt, got, ref)
    e.close(); p.close()


def test_fused_and_standalone_meter_paths_agree(qh):
    """One channel with a running AGC forces the whole engine onto the per-mode path (meter_kernel); the other channels'
    readings must agree with the fused launch's."""
    nch = 4
    x = synth.make_input_numpy(nch, 37 * 1024)
    vals = []
    for force_mixed in (False, True):
        e = qh.RxaEngine(nch + 1 if force_mixed else nch)
        _setup(e, nch, 3.0)
        if force_mixed:
            e.SetRXAMode(nch, 1); e.SetRXAAGCMode(nch, 3)
        e.enable_meters(True)
        xin = x if not force_mixed else np.concatenate([x, x[:1]], axis=0)
        e.process_host(np.ascontiguousarray(xin[:, :16 * 1024]))
        e.process_host(np.ascontiguousarray(xin[:, 16 * 1024:]))
        vals.append([[e.GetRXAMeter(ch, mt) for mt in range(7)] for ch in range(nch)])
        e.close()
    a, b = np.array(vals[0]), np.array(vals[1])
    assert np.all(np.abs(a - b) < 1.01 * STEP_DB), (a, b)
    assert np.all(a[:, [0, 1, 2, 3, 5, 6]] > -200.0)


def test_meters_start_at_minus_400_and_follow_the_agc_gain(qh):
    e = qh.RxaEngine(2)
    _setup(e, 2, 0.0)
    e.enable_meters(True)
    assert e.GetRXAMeter(0, 1) == -400.0
    x = synth.make_input_numpy(2, 64 * 1024)
    e.process_host(x)
    s_av, agc_av = e.GetRXAMeter(0, 1), e.GetRXAMeter(0, 6)
    assert abs(agc_av - s_av) < 2 * STEP_DB                 # channel 0: 0 dB
    s1, a1 = e.GetRXAMeter(1, 1), e.GetRXAMeter(1, 6)
    assert abs((a1 - s1) - 1.0) < 2 * STEP_DB               # channel 1: fixed gain 1 dB
    e.close()
