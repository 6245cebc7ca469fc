"""Known answers for the sample-unpacking restatement, worked out by hand from quisk.c:3378-3392, 2923-2952, 3745-3760."""
import numpy as np

from oracle import ingest_oracle as io


def test_le24_known_values():
    # I = 0x000001 -> int 0x00000100 = 256; Q = 0xFFFFFF (-1) -> 0xFFFFFF00 = -256
    assert io.read_rx_udp_le(bytes([1, 0, 0, 0xff, 0xff, 0xff]))[0] == 256 - 256j
    # I = 0x7FFFFF -> 0x7FFFFF00; Q = 0x800000 -> 0x80000000 = -2^31; gain applies to both
    v = io.read_rx_udp_le(bytes([0xff, 0xff, 0x7f, 0, 0, 0x80]), gain=0.5)[0]
    assert v == 0.5 * 0x7fffff00 - 0.5j * 2 ** 31


def test_add_rx_samples_endianness():
    le = io.add_rx_samples(bytes([0x34, 0x12, 0x00, 0x80]), 2, False)[0]       # I = 0x1234 << 16, Q = 0x8000 << 16
    assert le == float(0x12340000) - 1j * 2 ** 31
    be = io.add_rx_samples(bytes([0x12, 0x34, 0x80, 0x00]), 2, True)[0]
    assert be == le
    b3 = io.add_rx_samples(bytes([0xff, 0xff, 0xfe, 0x00, 0x00, 0x02]), 3, True)[0]    # -2 and +2, left-justified
    assert b3 == -512 + 512j


def test_hermes_frame_layout_and_iq_order():
    f = bytearray(512)
    f[0:3] = b"\x7f\x7f\x7f"
    # record 0, receiver 0: first triple 0x000001 goes to the IMAGINARY part, second 0xFFFFFE to the real part
    f[8:14] = bytes([0, 0, 1, 0xff, 0xff, 0xfe])
    # record 1 starts 8 bytes later when nrx = 1 (6 sample bytes + 2 microphone bytes)
    f[16:22] = bytes([0x7f, 0xff, 0xff, 0x80, 0, 0])
    y = io.hermes_frames(bytes(f), 1)
    assert y.shape == (1, 63)
    assert y[0, 0] == -512 + 256j
    assert y[0, 1] == -2.0 ** 31 + 1j * float(0x7fffff00)
    f2 = bytearray(512)
    f2[8 + 14 + 6:8 + 14 + 12] = bytes([0, 0, 2, 0, 0, 3])      # nrx = 2: record stride 14, receiver 1 at +6; record 1
    y2 = io.hermes_frames(bytes(f2), 2)
    assert y2.shape == (2, 36) and y2[1, 1] == 768 + 512j and y2[0, 1] == 0


def test_udp17_two_streams_marks_and_dc():
    """read_rx_udp17, quisk.c:3917-3996: by hand.  Packet of 2 + 4 records."""
    pk = bytearray(2 + 24)
    pk[0], pk[1] = 7, 0x02                                  # sequence number, overrange bit
    # record 0: I = 0x000002 (LSB clear: channel 0), Q = 0x000005
    pk[2:8] = bytes([2, 0, 0, 5, 0, 0])
    # record 1: I = 0x000003 (LSB set: channel 1), Q = 0x000004 (LSB clear: start-of-first-block mark)
    pk[8:14] = bytes([3, 0, 0, 4, 0, 0])
    # record 2: I = 0xFFFFFF (-1, LSB set: channel 1), Q = 0x000007 (LSB set: no mark)
    pk[14:20] = bytes([0xff, 0xff, 0xff, 7, 0, 0])
    # record 3: I = 0x800000 (most negative, LSB clear: channel 0), Q = 0x7FFFFF
    pk[20:26] = bytes([0, 0, 0x80, 0xff, 0xff, 0x7f])
    ch0, ch1, marks, over, dcs = io.read_rx_udp17(bytes(pk), packet_bytes=26, gain=2.0, invert_spectrum=True, dc=10 - 20j)
    assert over == 1
    assert list(ch0) == [2.0 * (512 + 1280j), 2.0 * (-2.0 ** 31 + 1j * float(0x7fffff00))]
    # channel 1: conjugated, then dc removed; the sum is taken before the removal
    raw1 = [2.0 * (768 - 1024j), 2.0 * (-256 - 1792j)]
    assert list(ch1) == [raw1[0] - (10 - 20j), raw1[1] - (10 - 20j)]
    assert list(marks) == [0] and dcs == raw1[0] + raw1[1]
