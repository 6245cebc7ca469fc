"""Audio egress (include/quiskhip.h 7b): the narrowing of Quisk's sound back ends -- (short)(int)(volume * x / 65536),
the Int24 / Int32 forms and (float)(volume * x / CLIP32): sound_alsa.c:344-390, sound_pulseaudio.c:684-695 -- done in the
store of the receive chain's last kernel.  Bit-exact against that C expression evaluated (numpy, same operation order)
on the complex-double output the same engine produces without egress.  -m gpu."""
import numpy as np
import pytest
import torch

import quisk_amd
from quisk_amd import synth

pytestmark = pytest.mark.gpu
CLIP32 = 2147483647.0


def c_rule(kind, x, volume, prescale):
    """x: float64 array of real or imaginary parts"""
    t = x * prescale if prescale != 1.0 else x.copy()
    t = volume * t
    if kind == "i16":
        return np.trunc(t / 65536).astype(np.int64).astype(np.int32).astype(np.int16)        # (short)(int): wraps
    if kind == "i24":
        return np.trunc(t / 256).astype(np.int64).astype(np.int32)
    if kind == "i32":
        return np.trunc(t).astype(np.int64).astype(np.int32)
    return (t / CLIP32).astype(np.float32)


def _engine(qh, nch, mode, agc_mode=0, meters=False, out_rate=48000, tile=0):
    e = qh.RxaEngine(nch, out_rate=out_rate)
    e.set_band_tile(tile)
    for c in range(nch):
        e.SetRXAShiftRun(c, 1); e.SetRXAShiftFreq(c, synth.shift_freq(c)); e.RXANBPSetRun(c, 1)
        e.SetRXAMode(c, mode); e.SetRXAAGCMode(c, agc_mode); e.SetRXAAGCFixed(c, 0.0)
        e.RXASetPassband(c, *((300.0, 3000.0) if mode == 1 else (-4000.0, 4000.0)))
    e.enable_meters(meters)
    return e


def _frames(fmt, raw, nch, n):
    if fmt.kind == 2:       # Int24: three bytes per slot, little endian, sign in the top byte
        b = raw.reshape(nch, n, fmt.num_channels, 3).astype(np.int32)
        v = b[..., 0] | (b[..., 1] << 8) | (b[..., 2] << 16)
        return np.where(v & 0x800000, v - (1 << 24), v)
    return raw.view(quisk_amd.AudioFormat.NP[fmt.kind]).reshape(nch, n, fmt.num_channels)


@pytest.mark.parametrize("kind", ["i16", "i24", "i32", "f32"])
@pytest.mark.parametrize("path", ["fast", "fast_meters", "modes", "resampled", "fast_tile8k", "fast_meters_tile8k"])
def test_egress_is_the_reference_expression_bit_for_bit(qh, kind, path):
    nch, nblk = 3, 23
    dev = torch.device("cuda:0")
    x = torch.from_numpy(synth.make_input_numpy(nch, nblk * 1024)).to(dev)
    kw = {"fast": dict(mode=1), "fast_meters": dict(mode=1, meters=True), "modes": dict(mode=6, agc_mode=3),
          "resampled": dict(mode=1, out_rate=96000), "fast_tile8k": dict(mode=1, tile=8192),
          "fast_meters_tile8k": dict(mode=1, meters=True, tile=8192)}[path]
    ea, eb = _engine(qh, nch, **kw), _engine(qh, nch, **kw)
    n_out = nblk * ea.dsp_outsize
    y = torch.zeros((nch, n_out), dtype=torch.complex128, device=dev)
    fmt = quisk_amd.AudioFormat(kind, volume=0.37, prescale=CLIP32, num_channels=3, channel_I=2, channel_Q=0)
    rowb = n_out * fmt.frame_bytes + 16
    out = torch.full((nch, rowb), 0x5A, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    # two calls, to carry state across an egress call as well
    h = (nblk // 2) * 1024
    ho = (nblk // 2) * ea.dsp_outsize
    for eng_call in range(2):
        i0, o0, nb = (0, 0, nblk // 2) if eng_call == 0 else (h, ho, nblk - nblk // 2)
        ea.process_ptr(x.data_ptr() + 16 * i0, nblk * 1024, y.data_ptr() + 16 * o0, n_out, nb)
        eb.process_audio_ptr(x.data_ptr() + 16 * i0, nblk * 1024, out.data_ptr() + o0 * fmt.frame_bytes, rowb, nb, fmt)
    ea.synchronize(); eb.synchronize()
    yh = y.cpu().numpy()
    raw = out.cpu().numpy()
    got = _frames(fmt, np.ascontiguousarray(raw[:, :n_out * fmt.frame_bytes]), nch, n_out)
    assert np.abs(yh).max() * CLIP32 * 0.37 > 1e6                 # real audio, well inside the 16-bit range after / 65536
    for part, slot in ((yh.real, 2), (yh.imag, 0)):
        want = c_rule(kind, np.ascontiguousarray(part), 0.37, CLIP32)
        assert np.array_equal(got[..., slot], want), (kind, path, slot)
    if kind != "i24":
        assert np.all(got[..., 1] == np.frombuffer(bytes([0x5A] * 4), dtype=got.dtype)[0])       # the unused slot is not written
    assert np.all(raw[:, n_out * fmt.frame_bytes:] == 0x5A)        # nothing beyond the frames


def test_audio_pack_standalone_and_int16_wrap(qh):
    """qh_audio_pack on arbitrary doubles, including values whose (int) does not fit a short: (short) wraps like the C cast."""
    import ctypes as C
    L = quisk_amd.load()
    rng = np.random.default_rng(3)
    n = 5000
    z = (rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n))) * 2.0 ** 30
    z[0, :4] = [65536 * 40000.7, -65536 * 40000.7, 65535.9999, -65535.9999]
    dev = torch.device("cuda:0")
    src = torch.from_numpy(z).to(dev)
    fmt = quisk_amd.AudioFormat("i16", volume=1.0)
    dst = torch.zeros((2, n * 4), dtype=torch.uint8, device=dev)
    rc = L.qh_audio_pack(0, None, src.data_ptr(), n, 2, n, C.byref(fmt), dst.data_ptr(), n * 4)
    assert rc == 0
    torch.cuda.synchronize()
    got = dst.cpu().numpy().view(np.int16).reshape(2, n, 2)
    assert np.array_equal(got[..., 0], c_rule("i16", z.real, 1.0, 1.0)) and np.array_equal(got[..., 1], c_rule("i16", z.imag, 1.0, 1.0))
    assert got[0, 0, 0] == np.int16(40000 - 65536) and got[0, 2, 0] == 0 and got[0, 3, 0] == 0
