"""RXASetNC 8192 ... 65536 (wdsp/RXA.c:934-946 sets nbp0, bpsnba, bp1 and the FM filters alike; fircore takes any multiple of the
block size, firmin.c:290-346): impulse responses longer than one tile run as partitions of 4096 taps over delayed views of the stream
(Engine::run_band, quisk_amd/csrc/qh_engine.hip).  Against the oracle's uniformly partitioned fircore: SSB with and without the meters,
AM (nbp0 + bp1), FM (nbp0, de-emphasis, audio filter), ragged calls down to one DSP block, nc changed in mid-stream (the delay lines
restart, setNc_fircore firmin.c:454-466), channels with different nc in one engine, and through the WDSP names block by block.  -m gpu."""
import ctypes as C

import numpy as np
import pytest

from conftest import rel_rms
from quisk_amd import synth
from test_gpu_band_tile_8192 import STEP_DB, _oracle, _setup

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nc", [8192, 16384, 32768, 65536])
def test_ssb_ragged_calls_with_and_without_meters(qh, oracle, nc):
    nch = 3
    calls = [30, 1, 7, 50, 2, 24, 1, 1, 40]            # 7680 ... 256 DSP-rate samples per call
    x = synth.make_input_numpy(nch, sum(calls) * 1024)
    for meters in (False, True):
        e = qh.RxaEngine(nch)
        _setup(e, nch, nc, agc_db=6.0)
        e.enable_meters(meters)
        refs = [_oracle(oracle, c, nc, agc_db=6.0) for c in range(nch)]
        pos = 0
        for k, nb in enumerate(calls):
            seg = np.ascontiguousarray(x[:, pos * 1024:(pos + nb) * 1024])
            pos += nb
            y = e.process_host(seg)
            for c in range(nch):
                ref = refs[c].xrxa(seg[c])
                assert rel_rms(y[c], ref) < 1e-9 or np.abs(ref).max() < 1e-12, (meters, k, c, rel_rms(y[c], ref))
                if meters:
                    for mt in (0, 1, 2, 3, 5, 6):
                        got, want = e.GetRXAMeter(c, mt), refs[c].GetRXAMeter(mt)
                        assert abs(got - want) < 1.01 * STEP_DB, (k, c, mt, got, want)
        assert e.band_tile() == 8192
        e.close()


@pytest.mark.parametrize("nc", [8192, 16384])
@pytest.mark.parametrize("mode,sig,passband", [(6, "am", (-4000.0, 4000.0)), (5, "fm", (-8000.0, 8000.0))], ids=["am", "fm"])
def test_am_and_fm_every_fircore_stage(qh, oracle, nc, mode, sig, passband):
    nch, nblk = 3, 240
    x = np.stack([synth.make_mode_input_numpy(sig, c, nblk * 1024) for c in range(nch)])
    e = qh.RxaEngine(nch)
    _setup(e, nch, nc, mode=mode, passband=passband)
    y = np.concatenate([e.process_host(np.ascontiguousarray(x[:, a * 1024:b * 1024])) for a, b in ((0, 90), (90, 91), (91, 170), (170, 240))], axis=1)
    settle = (nc // 256 + 150) * 256 if mode == 5 else 0      # FM: the loop's start-up depends on the filters' last bit (DESIGN.md parity caveat)
    for c in range(nch):
        ref = _oracle(oracle, c, nc, mode=mode, passband=passband).xrxa(x[c])
        assert np.abs(ref[settle:]).max() > 1e-3
        assert rel_rms(y[c][settle:], ref[settle:]) < (1e-6 if mode == 5 else 1e-9), (c, rel_rms(y[c][settle:], ref[settle:]))
    e.close()


def test_nc_changed_in_mid_stream_and_mixed_in_one_engine(qh, oracle):
    """channel 0: 2048 -> 8192 -> 16384 -> 1024; channel 1 stays at 2048; channel 2 at 8192 throughout."""
    nch = 3
    plan = [(20, (None, None, 8192)), (30, (8192, None, None)), (25, (16384, None, None)), (1, (None, None, None)), (30, (1024, None, None))]
    x = synth.make_input_numpy(nch, sum(p[0] for p in plan) * 1024)
    e = qh.RxaEngine(nch)
    _setup(e, nch)
    refs = [_oracle(oracle, c) for c in range(nch)]
    pos = 0
    for k, (nb, ncs) in enumerate(plan):
        for c, nc in enumerate(ncs):
            if nc:
                e.RXASetNC(c, nc); refs[c].RXASetNC(nc)
        seg = np.ascontiguousarray(x[:, pos * 1024:(pos + nb) * 1024])
        pos += nb
        y = e.process_host(seg)
        for c in range(nch):
            ref = refs[c].xrxa(seg[c])
            assert rel_rms(y[c], ref) < 1e-9, (k, c, rel_rms(y[c], ref))
    e.close()


def test_a_stage_changing_form_takes_the_other_channels_delay_lines_along(qh, oracle):
    """One channel's nc going over 4096 turns the stage's passes into partitions for every channel of the engine (and back): the
    others' delay lines must not notice -- they were found restarting by a seeded setter walk (tests/test_gpu_rxa_fuzz.py)."""
    nch = 3
    plan = [(20, None), (30, 8192), (1, None), (25, 2048), (20, 16384), (15, 512)]      # channel 1's nc ahead of the stretch
    for mode, passband, sig in ((1, (300.0, 3000.0), None), (6, (-4000.0, 4000.0), "am")):
        n = sum(p[0] for p in plan) * 1024
        x = synth.make_input_numpy(nch, n) if sig is None else np.stack([synth.make_mode_input_numpy(sig, c, n) for c in range(nch)])
        e = qh.RxaEngine(nch)
        _setup(e, nch, mode=mode, passband=passband)
        refs = [_oracle(oracle, c, mode=mode, passband=passband) for c in range(nch)]
        pos = 0
        for k, (nb, nc) in enumerate(plan):
            if nc:
                e.RXASetNC(1, nc); refs[1].RXASetNC(nc)
            seg = np.ascontiguousarray(x[:, pos * 1024:(pos + nb) * 1024])
            pos += nb
            y = e.process_host(seg)
            for c in range(nch):
                ref = refs[c].xrxa(seg[c])
                assert rel_rms(y[c], ref) < 1e-9, (mode, k, c, rel_rms(y[c], ref))
        e.close()


def test_long_fm_delay_lines_of_a_channel_that_leaves_the_mode_and_returns(qh, oracle):
    """the partitioned forms (nc > 4096) keep 16383-sample delay lines per stage in a ping-pong pair that flips for the
    LISTED channels only.  A channel that leaves FM while the others keep flipping the pair and comes back an odd number of calls later
    found its de-emphasis and audio filter reading the other half (stale or zero); WDSP's fircores keep their delay lines across that
    (SetRXAMode only clears fmd's run flag, RXA.c:758-776).  Channel 0: FM, USB for three calls, FM again; channels 1 and 2 stay."""
    nch, nc = 3, 8192
    plan = [(220, None), (5, 1), (6, None), (7, None), (120, 5)]         # (blocks, channel 0's new mode ahead of the call)
    n = sum(p[0] for p in plan) * 1024
    x = np.stack([synth.make_mode_input_numpy("fm", c, n) for c in range(nch)])
    e = qh.RxaEngine(nch)
    _setup(e, nch, nc, mode=5, passband=(-8000.0, 8000.0))
    refs = [_oracle(oracle, c, nc, mode=5, passband=(-8000.0, 8000.0)) for c in range(nch)]
    pos, outs, wants = 0, [[] for _ in range(nch)], [[] for _ in range(nch)]
    for nb, m in plan:
        if m is not None:
            e.SetRXAMode(0, m); refs[0].SetRXAMode(m)
        seg = np.ascontiguousarray(x[:, pos * 1024:(pos + nb) * 1024])
        pos += nb
        y = e.process_host(seg)
        for c in range(nch):
            outs[c].append(y[c]); wants[c].append(refs[c].xrxa(seg[c]))
    settle = (nc // 256 + 150) * 256              # the loop's start-up depends on the filters' last bit (DESIGN.md parity caveat)
    for c in range(nch):
        got, want = np.concatenate(outs[c]), np.concatenate(wants[c])
        assert np.abs(want[settle:]).max() > 1e-3
        assert rel_rms(got[settle:], want[settle:]) < 1e-6, (c, rel_rms(got[settle:], want[settle:]))
    back = sum(p[0] for p in plan[:4]) * 256      # channel 0 is FM again from here: its filters' first 8191 outputs see the old delay lines
    got, want = np.concatenate(outs[0])[back:back + 2 * nc], np.concatenate(wants[0])[back:back + 2 * nc]
    assert rel_rms(got, want) < 1e-6, rel_rms(got, want)
    e.close()


def test_through_the_wdsp_names_block_by_block(qh, oracle):
    """OpenChannel + RXASetNC(8192) + fexchange0: one DSP block per call, the partitions inside every block."""
    from test_gpu_wdsp_dropin import _open, _run
    from test_gpu_wdsp_dropin import _oracle as _chan
    lib = qh.load()
    in_size, out_size, ch = 1024, 256, 14
    x = synth.make_input_numpy(1, 160 * in_size)[0]
    _open(lib, ch, in_size, 256, 192000, nbp=True, shift_freq=10000.0, nc=8192)
    try:
        y = _run(lib, ch, x, in_size, out_size)
    finally:
        lib.CloseChannel(ch)
    ref, _ = _chan(oracle, in_size, 256, 192000, True, 10000.0, nc=8192).fexchange0(x)
    assert np.abs(ref).max() > 1e-3 and rel_rms(y, ref[:y.size]) < 1e-9, rel_rms(y, ref[:y.size])


def test_limits(qh):
    e = qh.RxaEngine(1)
    with pytest.raises(Exception):
        e.RXASetNC(0, 131072)
    with pytest.raises(Exception):
        e.RXASetNC(0, 12288)                    # a power of two, like the 4096-tap partitions expect
    e.close()
