"""The WDSP-named exports, bound the way quisk_wdsp.py binds libwdsp (ctypes, ints as int, floats as
c_double: quisk_wdsp.py:79-91,128-139), against the oracle's fexchange0.  -m gpu."""
import ctypes as C

import numpy as np
import pytest

from conftest import rel_rms
from quisk_amd import synth

pytestmark = pytest.mark.gpu
D = C.c_double


def _open(lib, channel, in_size, dsp_size, in_rate, nbp, shift_freq=None, nc=None):
    # the call sequence of quisk_wdsp.Cwdsp.open (quisk_wdsp.py:69-99) with the bench's shift/NBP options
    lib.OpenChannel(channel, in_size, dsp_size, in_rate, 48000, 48000, 0, 1, D(0.010), D(0.025), D(0.0), D(0.010), 1)
    assert lib.qh_wdsp_status() == 0, lib.qh_last_error()
    lib.SetRXAShiftRun(channel, 1 if shift_freq is not None else 0)
    if shift_freq is not None:
        lib.SetRXAShiftFreq(channel, D(shift_freq))
    lib.RXANBPSetRun(channel, 1 if nbp else 0)
    lib.SetRXAAMSQRun(channel, 0)
    lib.SetRXAMode(channel, 1)
    lib.RXASetPassband(channel, D(300.0), D(3000.0))
    if nc:
        lib.RXASetNC(channel, nc)
    lib.RXASetMP(channel, 0)
    lib.SetRXAAGCMode(channel, 0)
    lib.SetRXAAGCFixed(channel, D(0.0))
    lib.SetRXAPanelRun(channel, 0)
    lib.SetRXAEMNRRun(channel, 0)
    assert lib.qh_wdsp_status() == 0, lib.qh_last_error()


def _run(lib, channel, x, in_size, out_size):
    nb = x.size // in_size
    out = np.zeros(nb * out_size, dtype=np.complex128)
    err = C.c_int(0)
    for b in range(nb):
        blk = np.ascontiguousarray(x[b * in_size:(b + 1) * in_size])
        lib.fexchange0(channel, blk.ctypes.data_as(C.c_void_p), out[b * out_size:].ctypes.data_as(C.c_void_p), C.byref(err))
        assert err.value == 0
    return out


def _oracle(po, in_size, dsp_size, in_rate, nbp, shift_freq=None, nc=None):
    ch = po.WdspChannel(in_size, dsp_size, in_rate, 48000, 48000)
    ch.SetRXAShiftRun(1 if shift_freq is not None else 0)
    if shift_freq is not None:
        ch.SetRXAShiftFreq(shift_freq)
    ch.RXANBPSetRun(1 if nbp else 0)
    ch.SetRXAMode(1)
    ch.RXASetPassband(300.0, 3000.0)
    if nc:
        ch.RXASetNC(nc)
    ch.SetRXAAGCMode(0)
    ch.SetRXAAGCFixed(0.0)
    return ch


@pytest.mark.parametrize("in_size,in_rate,nc", [(1024, 192000, None), (64, 192000, None), (256, 48000, 256), (4096, 192000, None)])
def test_fexchange0_matches_oracle_including_latency_and_slew(qh, oracle, in_size, in_rate, nc):
    lib = qh.load()
    assert lib.GetWDSPVersion() == 125
    ch = 3
    d = in_rate // 48000
    n = max(in_size, 256 * d) * 24
    x = synth.make_input_numpy(1, n, fs=float(in_rate))[0]
    x[:37] = 0.0                                   # leading zeros: the upslew triggers on the first non-zero sample
    _open(lib, ch, in_size, 256, in_rate, nbp=True, shift_freq=10000.0, nc=nc)
    try:
        y = _run(lib, ch, x, in_size, in_size // d)
    finally:
        lib.CloseChannel(ch)
    ref, errs = _oracle(oracle, in_size, 256, in_rate, True, 10000.0, nc).fexchange0(x)
    assert errs == 0
    assert rel_rms(y, ref) < 1e-9
    # the output is silent for the two-block latency plus the 10 ms delay, like the reference
    assert np.abs(y[:512]).max() < 1e-12


def test_two_channels_are_independent(qh, oracle):
    lib = qh.load()
    x0 = synth.make_input_numpy(1, 1024 * 10)[0]
    x1 = synth.make_input_numpy(1, 1024 * 10, first_channel=1)[0]
    _open(lib, 0, 1024, 256, 192000, nbp=True, shift_freq=10000.0)
    _open(lib, 1, 1024, 256, 192000, nbp=False)
    try:
        y0, y1 = np.zeros(2560, dtype=np.complex128), np.zeros(2560, dtype=np.complex128)
        err = C.c_int(0)
        for b in range(10):                         # interleaved calls, like two receivers in one loop
            for chan, xs, ys in ((0, x0, y0), (1, x1, y1)):
                blk = np.ascontiguousarray(xs[b * 1024:(b + 1) * 1024])
                lib.fexchange0(chan, blk.ctypes.data_as(C.c_void_p), ys[b * 256:].ctypes.data_as(C.c_void_p), C.byref(err))
    finally:
        lib.CloseChannel(0)
        lib.CloseChannel(1)
    r0, _ = _oracle(oracle, 1024, 256, 192000, True, 10000.0).fexchange0(x0)
    r1, _ = _oracle(oracle, 1024, 256, 192000, False).fexchange0(x1)
    assert rel_rms(y0, r0) < 1e-9 and rel_rms(y1, r1) < 1e-9


def test_closed_channel_and_unsupported_requests_are_reported(qh):
    lib = qh.load()
    err = C.c_int(0)
    buf = np.zeros(256, dtype=np.complex128)
    lib.fexchange0(7, buf.ctypes.data_as(C.c_void_p), buf.ctypes.data_as(C.c_void_p), C.byref(err))
    assert err.value != 0 and lib.qh_wdsp_status() != 0
    lib.SetRXAEMNRRun(7, 1)
    assert lib.qh_wdsp_status() != 0            # NR2 is outside the GPU hot path: reported, not silently ignored


def test_meters_and_quisk_shim(qh, oracle):
    """GetRXAMeter after a run, and quisk_wdsp.c's re-blocking shim wdspFexchange0 (CLIP32 scaling, ragged n)."""
    lib = qh.load()
    lib.GetRXAMeter.restype = C.c_double
    x = synth.make_input_numpy(1, 1024 * 30)[0]
    _open(lib, 5, 1024, 256, 192000, nbp=True, shift_freq=10000.0)
    try:
        lib.qh_wdsp_set_parameter(5, 1024, 1)
        buf = (x * 2147483647.0).copy()
        outs, pos = [], 0
        for k in (100, 1024, 3000, 5, 2048 * 4 + 17, x.size - (100 + 1024 + 3000 + 5 + 2048 * 4 + 17)):
            seg = np.ascontiguousarray(buf[pos:pos + k]).copy()
            work = np.zeros(max(k, 1) + 2048, dtype=np.complex128)
            work[:k] = seg
            n = lib.wdspFexchange0(5, work.ctypes.data_as(C.c_void_p), k)
            outs.append(work[:n].copy())
            pos += k
        y = np.concatenate(outs) / 2147483647.0
        meters = [lib.GetRXAMeter(5, mt) for mt in range(7)]
    finally:
        lib.qh_wdsp_set_parameter(5, 0, 0)
        lib.CloseChannel(5)
    o = _oracle(oracle, 1024, 256, 192000, True, 10000.0)
    ref, _ = o.fexchange0(x)
    # the shim returns whole in_size blocks worth of input: 29 of the 30 blocks have been consumed in order
    n = y.size
    assert n % 1024 == 0 and n >= 1024 * 29
    # the shim hands back in_size samples per block although WDSP produced out_size = in_size/4 (quisk_wdsp.c:61-64
    # advances by in_size): the first out_size entries of every in_size chunk are the data
    got = y.reshape(-1, 1024)[:, :256].reshape(-1)
    assert rel_rms(got, ref[:got.size]) < 1e-9
    want = [o.GetRXAMeter(mt) for mt in range(7)]
    for mt in (0, 1, 2, 3, 5, 6):
        assert abs(meters[mt] - want[mt]) < 1e-3, (mt, meters[mt], want[mt])


def test_fexchange2_float_split_buffers(qh, oracle):
    """fexchange2 (wdsp/iobuffs.c:518-582): separate float I / Q buffers; same exchange as fexchange0 on the
    float-rounded input, output rounded to float."""
    lib = qh.load()
    ch, in_size, out_size, nb = 6, 1024, 256, 24
    _open(lib, ch, in_size, 256, 192000, True, shift_freq=synth.shift_freq(0))
    lib.SetRXAEMNRgainMethod(ch, 2)                    # accepted (quisk.py:6017), the block never runs
    assert lib.qh_wdsp_status() == 0
    x = synth.make_input_numpy(1, nb * in_size)[0]
    xi, xq = x.real.astype(np.float32), x.imag.astype(np.float32)
    oi, oq = np.zeros(nb * out_size, np.float32), np.zeros(nb * out_size, np.float32)
    err = C.c_int(0)
    for b in range(nb):
        lib.fexchange2(ch, xi[b * in_size:].ctypes.data_as(C.c_void_p), xq[b * in_size:].ctypes.data_as(C.c_void_p),
                       oi[b * out_size:].ctypes.data_as(C.c_void_p), oq[b * out_size:].ctypes.data_as(C.c_void_p), C.byref(err))
        assert err.value == 0
    lib.CloseChannel(ch)
    ref, errs = _oracle(oracle, in_size, 256, 192000, True, shift_freq=synth.shift_freq(0)).fexchange0(xi.astype(np.float64) + 1j * xq.astype(np.float64))
    assert errs == 0
    assert rel_rms(oi + 1j * oq, ref) < 2e-7            # float32 rounding of the output


def test_setrxaamdrun_switches_the_am_detector(qh, oracle):
    """SetRXAAMDRun (wdsp/amd.c:264-277) on its own: USB mode with the AM detector forced on == the oracle's."""
    lib = qh.load()
    ch, in_size, out_size, nb = 7, 1024, 256, 40
    _open(lib, ch, in_size, 256, 192000, True, shift_freq=synth.shift_freq(0))
    lib.SetRXAAMDRun(ch, 1)
    assert lib.qh_wdsp_status() == 0, lib.qh_last_error()
    x = synth.make_input_numpy(1, nb * in_size)[0]
    y = _run(lib, ch, x, in_size, out_size)
    lib.CloseChannel(ch)
    o = _oracle(oracle, in_size, 256, 192000, True, shift_freq=synth.shift_freq(0))
    o.SetRXAAMDRun(1)
    ref, errs = o.fexchange0(x)
    assert errs == 0 and np.abs(ref).max() > 1e-3
    assert rel_rms(y, ref) < 1e-9


@pytest.mark.parametrize("out_rate", [96000, 24000])
def test_fexchange0_with_output_resampler(qh, oracle, out_rate):
    """OpenChannel with out_rate != dsp_rate: out_size = in_size * out_rate / in_rate, latency and slew at the output rate."""
    lib = qh.load()
    ch, in_size, nb = 9, 1024, 24
    out_size = in_size * out_rate // 192000
    lib.OpenChannel(ch, in_size, 256, 192000, 48000, out_rate, 0, 1, D(0.010), D(0.025), D(0.0), D(0.010), 1)
    assert lib.qh_wdsp_status() == 0, lib.qh_last_error()
    lib.SetRXAShiftRun(ch, 1); lib.SetRXAShiftFreq(ch, D(synth.shift_freq(0))); lib.RXANBPSetRun(ch, 1)
    lib.SetRXAMode(ch, 1); lib.RXASetPassband(ch, D(300.0), D(3000.0)); lib.SetRXAAGCMode(ch, 0); lib.SetRXAAGCFixed(ch, D(0.0))
    x = synth.make_input_numpy(1, nb * in_size)[0]
    y = _run(lib, ch, x, in_size, out_size)
    lib.CloseChannel(ch)
    o = oracle.WdspChannel(in_size, 256, 192000, 48000, out_rate)
    o.SetRXAShiftRun(1); o.SetRXAShiftFreq(synth.shift_freq(0)); o.RXANBPSetRun(1); o.SetRXAMode(1)
    o.RXASetPassband(300.0, 3000.0); o.SetRXAAGCMode(0); o.SetRXAAGCFixed(0.0)
    ref, errs = o.fexchange0(x)
    assert errs == 0 and ref.size == y.size and np.abs(ref).max() > 1e-3
    assert rel_rms(y, ref) < 1e-9


def test_steady_state_blocks_are_replayed_from_hipgraphs_and_setters_invalidate_them(qh, oracle):
    """fexchange0's per-block launch sequence is captured once the parameters stand still; a setter in mid-stream
    drops the graphs and the output still follows the oracle."""
    lib = qh.load()
    lib.qh_wdsp_graph_launches.restype = C.c_longlong
    ch, in_size, out_size, nb = 11, 1024, 256, 40
    _open(lib, ch, in_size, 256, 192000, True, shift_freq=synth.shift_freq(0))
    o = _oracle(oracle, in_size, 256, 192000, True, shift_freq=synth.shift_freq(0))
    x = synth.make_input_numpy(1, nb * in_size)[0]
    before = lib.qh_wdsp_graph_launches()
    y1 = _run(lib, ch, x[:20 * in_size], in_size, out_size)
    r1, errs1 = o.fexchange0(x[:20 * in_size])
    mid = lib.qh_wdsp_graph_launches()
    assert mid - before >= 15                                   # all but the first few blocks
    lib.RXASetPassband(ch, D(200.0), D(2500.0)); o.RXASetPassband(200.0, 2500.0)
    lib.SetRXAShiftFreq(ch, D(9000.0)); o.SetRXAShiftFreq(9000.0)
    y2 = _run(lib, ch, x[20 * in_size:], in_size, out_size)
    assert lib.qh_wdsp_graph_launches() - mid >= 15
    lib.CloseChannel(ch)
    r2, errs2 = o.fexchange0(x[20 * in_size:])
    assert errs1 == 0 and errs2 == 0
    assert rel_rms(np.concatenate([y1, y2]), np.concatenate([r1, r2])) < 1e-9


def test_anf_and_anr_through_the_wdsp_names(qh, oracle):
    lib = qh.load()
    for n in ("SetRXAANFRun", "SetRXAANRRun", "SetRXAANFPosition"):
        getattr(lib, n).argtypes = [C.c_int, C.c_int]
    lib.SetRXAANRVals.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
    ch, in_size, out_size, nb = 12, 1024, 256, 50
    _open(lib, ch, in_size, 256, 192000, True, shift_freq=synth.shift_freq(0))
    o = _oracle(oracle, in_size, 256, 192000, True, shift_freq=synth.shift_freq(0))
    x = synth.make_input_numpy(1, nb * in_size)[0]
    lib.SetRXAANFRun(ch, 1); o.SetRXAANFRun(1)
    y1 = _run(lib, ch, x[:25 * in_size], in_size, out_size); r1, _ = o.fexchange0(x[:25 * in_size])
    lib.SetRXAANFRun(ch, 0); o.SetRXAANFRun(0)
    lib.SetRXAANRVals(ch, 32, 8, 1e-4, 0.1); o.SetRXAANRVals(32, 8, 1e-4, 0.1)
    lib.SetRXAANRRun(ch, 1); o.SetRXAANRRun(1)
    y2 = _run(lib, ch, x[25 * in_size:], in_size, out_size); r2, _ = o.fexchange0(x[25 * in_size:])
    assert lib.qh_wdsp_status() == 0
    lib.CloseChannel(ch)
    assert rel_rms(np.concatenate([y1, y2]), np.concatenate([r1, r2])) < 1e-6


def test_snba_through_the_wdsp_names_as_quisk_switches_it(qh, oracle):
    """quisk.py:6041-6047: SetRXASNBARun(ch, 1) on a running channel, off again later; block-at-a-time fexchange0 (graph replay)."""
    lib = qh.load()
    lib.SetRXASNBARun.argtypes = [C.c_int, C.c_int]
    ch, in_size, out_size, nb = 14, 1024, 256, 90
    _open(lib, ch, in_size, 256, 192000, True, shift_freq=synth.shift_freq(0))
    o = _oracle(oracle, in_size, 256, 192000, True, shift_freq=synth.shift_freq(0))
    x = synth.impulsive_input(1, nb * in_size, seed=4, scale=0.01)[0] + synth.make_input_numpy(1, nb * in_size)[0]
    ys, rs = [], []
    for (a, b), run in (((0, 20), 0), ((20, 70), 1), ((70, 90), 0)):
        lib.SetRXASNBARun(ch, run); o.SetRXASNBARun(run)
        assert lib.qh_wdsp_status() == 0
        ys.append(_run(lib, ch, x[a * in_size:b * in_size], in_size, out_size)); rs.append(o.fexchange0(x[a * in_size:b * in_size])[0])
    lib.CloseChannel(ch)
    y, r = np.concatenate(ys), np.concatenate(rs)
    assert rel_rms(y, r) < 1e-6, rel_rms(y, r)
    assert rel_rms(r[40 * out_size:70 * out_size], r[40 * out_size:70 * out_size].real) > 1e-3     # bp1 made it analytic again


def test_set_channel_state_slews_down_flushes_and_restarts_like_a_fresh_channel(qh, oracle):
    """SetChannelState(ch, 0, 0) (wdsp/channel.c:261-296): the following fexchange0 calls run the down-slew state machine on the
    output (iobuffs.c:226-300: the first sample as it is, then tdelaydown, then a raised cosine over tslewdown, then zeros up to the
    end of a block), after which the channel is flushed and stops exchanging; SetChannelState(ch, 1, 0) starts it again with the
    up-slew, and from there on it must behave exactly like a freshly opened channel."""
    lib = qh.load()
    lib.SetChannelState.argtypes = [C.c_int, C.c_int, C.c_int]
    ch, in_size, out_size = 13, 1024, 256
    _open(lib, ch, in_size, 256, 192000, True, shift_freq=synth.shift_freq(0))
    x = synth.make_input_numpy(1, 120 * in_size)[0]
    y1 = _run(lib, ch, x[:40 * in_size], in_size, out_size)
    assert lib.SetChannelState(ch, 0, 0) == 1                   # prior state
    # the blocks of the down-slew: compare with the same channel left running
    o = _oracle(oracle, in_size, 256, 192000, True, shift_freq=synth.shift_freq(0))
    ref, _ = o.fexchange0(x[:44 * in_size])
    assert rel_rms(y1, ref[:40 * out_size]) < 1e-9
    err = C.c_int(0)
    tail = np.full(4 * out_size, 7.0 + 7.0j)
    for b in range(4):
        blk = np.ascontiguousarray(x[(40 + b) * in_size:(41 + b) * in_size])
        lib.fexchange0(ch, blk.ctypes.data_as(C.c_void_p), tail[b * out_size:].ctypes.data_as(C.c_void_p), C.byref(err))
    ntdown = int(0.010 * 48000)
    ramp = 0.5 * (1.0 + np.cos(np.pi * np.arange(ntdown + 1) / ntdown))
    want = ref[40 * out_size:41 * out_size].copy()
    # sample 0 passes (BEGIN), tdelaydown = 0 -> DOWNSLEW: samples 1 .. 256 take cdown[0 .. 255] ...
    want[1:] *= ramp[:out_size - 1]
    assert np.abs(tail[:out_size] - want).max() < 1e-9 * np.abs(ref).max()
    # ... the second block continues the ramp to zero, then zeros to the block's end; after that the channel does not exchange
    want2 = ref[41 * out_size:42 * out_size].copy()
    k = np.arange(out_size) + out_size - 1
    want2 = np.where(k <= ntdown, want2 * ramp[np.minimum(k, ntdown)], 0.0)
    assert np.abs(tail[out_size:2 * out_size] - want2).max() < 1e-9 * np.abs(ref).max()
    assert lib.SetChannelState(ch, 1, 0) == 0
    y2 = _run(lib, ch, x[60 * in_size:], in_size, out_size)
    fresh = _oracle(oracle, in_size, 256, 192000, True, shift_freq=synth.shift_freq(0))
    ref2, _ = fresh.fexchange0(x[60 * in_size:])
    lib.CloseChannel(ch)
    assert rel_rms(y2, ref2) < 1e-9


def test_reference_quisk_wdsp_c_drives_our_fexchange0(qh, oracle):
    """Row b1 with the reference's own code in the loop: quisk_wdsp.c compiled from /root/reference (oracle/_ref, built in
    the build container, shipped as a binary) is handed the ADDRESS of libquiskhip's fexchange0 exactly as Quisk hands it
    libwdsp's (quisk_wdsp.py:57-67 -> QS.wdsp_set_parameter(0, fexchange0=fpt)), and re-blocks ragged sample counts into it.
    libquiskhip's own wdspFexchange0 on a second, identically configured channel must return the same bits."""
    ref = oracle.ref_wdsp_shim_lib()
    if ref is None:
        pytest.skip("oracle/_ref/libquisk_wdsp_ref.so not shipped")
    lib = qh.load()
    cha, chb, in_size = 11, 12, 1024
    for ch in (cha, chb):
        _open(lib, ch, in_size, 256, 192000, nbp=True, shift_freq=10000.0)
    try:
        fpt = C.cast(lib.fexchange0, C.c_void_p).value
        oracle.ref_wdsp_set_parameter(cha, in_size=in_size, fexchange0=fpt, in_use=1)
        lib.qh_wdsp_set_parameter(chb, in_size, 1)
        x = synth.make_input_numpy(1, 1024 * 40)[0] * 2147483647.0
        pos = 0
        for k in (100, 1024, 3000, 5, 0, 2048 * 4 + 17, 1, 1023, 2, 7000, 11111):
            a = np.zeros(k + 4 * in_size, dtype=np.complex128); a[:k] = x[pos:pos + k]
            b = a.copy()
            na = ref.wdspFexchange0(cha, a.ctypes.data, k)
            nb = lib.wdspFexchange0(chb, b.ctypes.data_as(C.c_void_p), k)
            pos += k
            assert na == nb, (k, na, nb)
            assert np.array_equal(a.view(np.float64), b.view(np.float64)), k
        assert np.abs(a[:na]).max() > 1e6               # real audio came back (CLIP32 scale), not zeros
    finally:
        oracle.ref_wdsp_set_parameter(cha, in_use=0)
        lib.qh_wdsp_set_parameter(chb, 0, 0)
        lib.CloseChannel(cha); lib.CloseChannel(chb)
