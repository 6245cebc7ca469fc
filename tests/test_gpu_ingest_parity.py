"""Wire-format ingest on the GPU (C ABI group 7): bit-exact against the restatement of the reference's unpacking
loops, stand-alone and fused into the RXA front kernel.  -m gpu."""
import numpy as np
import pytest

from conftest import rel_rms
from oracle import ingest_oracle as io

pytestmark = pytest.mark.gpu


def rand_bytes(seed, n):
    return np.random.default_rng(seed).integers(0, 256, n, dtype=np.uint8).tobytes()


@pytest.mark.parametrize("n", [1, 2, 3, 1000, 4097])
def test_le24_bit_exact(qh, n):
    buf = rand_bytes(n, 6 * n)
    want = io.read_rx_udp_le(buf, 3, 0.75)
    got = qh.ingest.unpack_host(buf, qh.IqFormat.le24(0.75), 1, 0, n)
    assert np.array_equal(got[0], want)         # integer -> double -> one multiply: no tolerance


@pytest.mark.parametrize("sb,be", [(1, 0), (2, 0), (2, 1), (3, 1), (4, 0), (4, 1)])
def test_add_rx_samples_formats_bit_exact(qh, sb, be):
    n = 777
    buf = rand_bytes(10 * sb + be, 2 * sb * n)
    want = io.add_rx_samples(buf, sb, be)
    got = qh.ingest.unpack_host(buf, qh.IqFormat.plain(sb, be), 1, 0, n)
    assert np.array_equal(got[0], want)


@pytest.mark.parametrize("nrx", [1, 2, 4])
def test_hermes_frames_multirx_bit_exact(qh, nrx):
    nframes = 9
    buf = rand_bytes(nrx, 512 * nframes)
    want = io.hermes_frames(buf, nrx)
    got = qh.ingest.unpack_host(buf, qh.IqFormat.hermes(nrx), nrx, 6, want.shape[1])       # receiver r: 6 r bytes in
    assert np.array_equal(got, want)


def test_two_channels_in_separate_buffers_and_fp32(qh):
    n = 500
    a, b = rand_bytes(1, 6 * n), rand_bytes(2, 6 * n)
    got = qh.ingest.unpack_host(a + b, qh.IqFormat.le24(), 2, 6 * n, n, dtype=1)
    assert np.array_equal(got[0], io.read_rx_udp_le(a).astype(np.complex64))
    assert np.array_equal(got[1], io.read_rx_udp_le(b).astype(np.complex64))


def test_buffer_too_short_is_an_error(qh):
    with pytest.raises(qh.QuiskHipError):
        qh.ingest.unpack_host(bytes(6 * 10), qh.IqFormat.le24(), 1, 0, 11)


def test_rxa_chain_fed_with_24bit_samples(qh, oracle):
    """qh_rxa_process_packed == unpack on the CPU + the WDSP restatement; and == the GPU chain fed with doubles."""
    nch, nblk = 3, 12
    n = nblk * 1024
    rng = np.random.default_rng(5)
    t = np.arange(n)
    raws, xs = [], []
    for c in range(nch):
        x = 0.3 * np.exp(2j * np.pi * ((10000.0 + 37 * c - 1000.0) / 192000 * t % 1.0)) + 0.01 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        q = np.round(np.stack([x.real, x.imag], axis=1) * 2 ** 23).astype("<i4")            # 24-bit ADC codes
        raw = q.view(np.uint8).reshape(n, 2, 4)[:, :, :3].tobytes()
        raws.append(raw)
        xs.append(io.read_rx_udp_le(raw, 3, 1.0 / 2 ** 31))
    eng = qh.RxaEngine(nch, dsp_size=256, in_rate=192000, dsp_rate=48000, out_rate=48000)
    eng2 = qh.RxaEngine(nch, dsp_size=256, in_rate=192000, dsp_rate=48000, out_rate=48000)
    for e in (eng, eng2):
        e.SetRXAShiftRun(-1, 1); e.RXANBPSetRun(-1, 1)
        e.SetRXAMode(-1, 1); e.RXASetPassband(-1, 300.0, 3000.0)
        e.SetRXAAGCMode(-1, 0); e.SetRXAAGCFixed(-1, 0.0)
        for c in range(nch):
            e.SetRXAShiftFreq(c, 10000.0 + 37 * c)
    fmt = qh.IqFormat.le24(1.0 / 2 ** 31)
    half = nblk // 2 * 1024 * 6
    parts = []
    for lo, hi, blocks in ((0, half, nblk // 2), (half, 6 * n, nblk - nblk // 2)):          # two calls: history carries over
        buf = b"".join(r[lo:hi] for r in raws)
        parts.append(eng.process_packed_host(buf, fmt, hi - lo, blocks))
    y = np.concatenate(parts, axis=1)
    y2 = eng2.process_host(np.stack(xs))
    assert rel_rms(y, y2) < 1e-13               # same arithmetic on bit-identical inputs; only the tiling of the two calls differs
    for c in range(nch):
        ch = oracle.WdspChannel(1024, 256, 192000, 48000, 48000)
        ch.SetRXAShiftRun(1); ch.SetRXAShiftFreq(10000.0 + 37 * c); ch.RXANBPSetRun(1)
        ch.SetRXAMode(1); ch.RXASetPassband(300.0, 3000.0); ch.SetRXAAGCMode(0); ch.SetRXAAGCFixed(0.0)
        want = ch.xrxa(xs[c])
        assert rel_rms(y[c], want) < 1e-9


@pytest.mark.parametrize("npk,invert", [(1, False), (5, True), (40, False)])
def test_udp17_two_stream_demultiplexer_bit_exact(qh, npk, invert):
    """read_rx_udp17 (quisk.c:3821-3999): random packets, so the two streams and the block marks interleave at random; 40 packets
    = 9600 records = several 1024-record tiles with carried offsets."""
    buf = bytearray(rand_bytes(170 + npk, 1442 * npk))
    want = io.read_rx_udp17(bytes(buf), 1442, 1.053497942, invert, 1234.5 - 987.25j)
    got = qh.ingest.unpack_udp17_host(bytes(buf), 1442, 1.053497942, invert, 1234.5 - 987.25j)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
    assert got[3] == want[3]
    assert len(got[0]) + len(got[1]) == 240 * npk and len(got[2]) > 0
    # the sum is added in another order on the GPU (lanes, wavefronts): last-bit agreement, not identity
    assert abs(got[4] - want[4]) <= 1e-12 * abs(want[4]) + 1e-3
    # one stream only: every I with the LSB clear
    b = np.frombuffer(bytes(buf), dtype=np.uint8).copy().reshape(npk, 1442)
    b[:, 2::6] &= 0xfe
    ch0, ch1, marks, over, dcs = qh.ingest.unpack_udp17_host(b.tobytes(), 1442, 1.0)
    assert len(ch0) == 240 * npk and len(ch1) == 0 and len(marks) == 0 and dcs == 0
    assert np.array_equal(ch0, io.read_rx_udp17(b.tobytes(), 1442, 1.0)[0])
