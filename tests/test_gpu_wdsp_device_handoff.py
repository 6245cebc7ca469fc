"""qh_wdsp_fexchange0_device -- Quisk's hand-off to WDSP (quisk.c:2660-2661 -> quisk_wdsp.c:24-69 -> fexchange0, wdsp/iobuffs.c:464-516)
for samples that are already on the GPU: the shim's ring, fexchange0's double rings, the up- and down-slews and the DSP blocks in
device memory, nothing copied to the host and nothing waited for.  It must leave the samples the host-pointer call leaves, bit for
bit: ragged call lengths, the up-slew's data-dependent trigger (leading zeros), a channel switched off and on in mid-stream, in_use
dropped and raised, and a caller that changes sides in mid-stream.  -m gpu."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import rel_rms
from quisk_amd import synth
from test_gpu_wdsp_dropin import _open

pytestmark = pytest.mark.gpu
CLIP32 = 2147483647.0


def _host_call(lib, ch, seg, in_size):
    work = np.zeros(seg.size + 2 * in_size, dtype=np.complex128)
    work[:seg.size] = seg
    n = lib.wdspFexchange0(ch, work.ctypes.data_as(C.c_void_p), seg.size)
    return work[:n].copy()


def _dev_call(lib, ch, seg, in_size, dev, stream):
    with torch.cuda.stream(stream):     # (the fill too: torch's streams do not wait for the default stream)
        work = torch.zeros(seg.size + 2 * in_size, dtype=torch.complex128, device=dev)
        work[:seg.size] = torch.from_numpy(seg).to(dev)
        n = lib.qh_wdsp_fexchange0_device(ch, C.c_void_p(work.data_ptr()), seg.size, C.c_void_p(stream.cuda_stream))
        assert lib.qh_wdsp_status() == 0, lib.qh_last_error()
        out = work[:n].cpu()            # (on `stream`: ordered behind the channel's work by the call itself)
    stream.synchronize()
    return out.numpy()


SIZES = [100, 1024, 3000, 5, 2048 * 4 + 17, 1, 1023, 4096, 777, 6000, 2048, 333]


@pytest.mark.parametrize("in_size,in_rate,lead", [(1024, 192000, 0), (256, 48000, 700), (64, 192000, 131)], ids=["1024-192k", "256-48k-zeros", "64-192k-zeros"])
def test_device_hand_off_is_the_host_hand_off_bit_for_bit(qh, in_size, in_rate, lead):
    lib = qh.load()
    lib.SetChannelState.argtypes = [C.c_int, C.c_int, C.c_int]
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    total = sum(SIZES) * 3
    x = synth.make_input_numpy(1, total)[0] * CLIP32
    x[:lead] = 0.0                                       # the up-slew waits for the first non-zero sample (iobuffs.c:104-113)
    a, b = 6, 7
    for ch in (a, b):
        _open(lib, ch, in_size, 256, in_rate, nbp=True, shift_freq=None if in_rate == 48000 else 10000.0, nc=256 if in_rate == 48000 else None)
        lib.qh_wdsp_set_parameter(ch, in_size, 1)
    try:
        pos, step = 0, 0
        got_h, got_d = [], []
        for rep in range(3):
            for k in SIZES:
                seg = np.ascontiguousarray(x[pos:pos + k])
                pos += k
                step += 1
                if step == 14:                           # down-slew, flush, stop (channel.c:261-296) ...
                    assert lib.SetChannelState(a, 0, 0) == 1 and lib.SetChannelState(b, 0, 0) == 1
                if step == 20:                           # ... and on again: the up-slew from a flushed channel
                    assert lib.SetChannelState(a, 1, 0) == 0 and lib.SetChannelState(b, 1, 0) == 0
                if step == 27:                           # in_use dropped: the shim rewinds its ring (quisk_wdsp.c:32-37)
                    lib.qh_wdsp_set_parameter(a, -1, 0); lib.qh_wdsp_set_parameter(b, -1, 0)
                if step == 29:
                    lib.qh_wdsp_set_parameter(a, -1, 1); lib.qh_wdsp_set_parameter(b, -1, 1)
                h = _host_call(lib, a, seg, in_size)
                d = _dev_call(lib, b, seg, in_size, dev, stream)
                assert h.size == d.size, (step, k, h.size, d.size)
                assert np.array_equal(h.view(np.float64), d.view(np.float64)), (step, k, np.abs(h - d).max())
                got_h.append(h); got_d.append(d)
        y = np.concatenate(got_h)
        assert np.abs(y).max() > 1e6                     # there is audio in it
    finally:
        for ch in (a, b):
            lib.qh_wdsp_set_parameter(ch, 0, 0)
            lib.wdspFexchange0(ch, None, 0)              # not in use: the shim rewinds its ring, nothing is left for the next test
            lib.CloseChannel(ch)


def test_a_caller_may_change_sides_in_mid_stream(qh):
    """Host-pointer calls and device calls on ONE channel in turn (the rings, the pending DSP block, the shim's ring and the up-slew's
    state move with it) against a channel that only ever sees host-pointer calls: the same bits.  (The host-pointer path against the
    oracle: tests/test_gpu_wdsp_dropin.py; the device path inside quisk_process_samples against the oracle:
    tests/test_gpu_quisk_process_samples.py::test_wdsp_hand_off_inside_the_block.)"""
    lib = qh.load()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    in_size, in_rate, a, b = 1024, 192000, 9, 10
    x = synth.make_input_numpy(1, sum(SIZES) * 2)[0] * CLIP32
    x[:50] = 0.0
    for ch in (a, b):
        _open(lib, ch, in_size, 256, in_rate, nbp=True, shift_freq=10000.0)
        lib.qh_wdsp_set_parameter(ch, in_size, 1)
    try:
        pos, sides = 0, set()
        for j, k in enumerate(SIZES * 2):
            seg = np.ascontiguousarray(x[pos:pos + k])
            pos += k
            on_dev = (j // 3) % 2 == 0
            sides.add(on_dev)
            got = _dev_call(lib, a, seg, in_size, dev, stream) if on_dev else _host_call(lib, a, seg, in_size)
            want = _host_call(lib, b, seg, in_size)
            assert got.size == want.size, (j, k)
            assert np.array_equal(got.view(np.float64), want.view(np.float64)), (j, k, on_dev, np.abs(got - want).max())
        assert sides == {True, False}
    finally:
        for ch in (a, b):
            lib.qh_wdsp_set_parameter(ch, 0, 0)
            lib.wdspFexchange0(ch, None, 0)              # not in use: the shim rewinds its ring, nothing is left for the next test
            lib.CloseChannel(ch)
