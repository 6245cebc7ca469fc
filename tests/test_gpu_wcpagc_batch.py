"""xwcpagc (wdsp/wcpAGC.c:177-338) in its two GPU forms: sample by sample, and 64 samples per step of the wavefront with only the level
detector stepped in order (qh_demod.hpp).  Same state, same arithmetic in the same order: outputs are bit-identical, the forms can
alternate between calls, and both follow the oracle.  Signals that walk the detector through all five states: bursts over a quiet
floor (attack, fast decay, hang, hang decay), a fade (decay), silence (the min_volts clamp), a click (pop ratio).  -m gpu."""
import numpy as np
import pytest

from conftest import rel_rms
from quisk_amd import synth

pytestmark = pytest.mark.gpu


def _input(nch, nblk, seed=5):
    rng = np.random.default_rng(seed)
    n = nblk * 1024
    t = np.arange(n)
    x = np.zeros((nch, n), dtype=np.complex128)
    for c in range(nch):
        f = (synth.shift_freq(c) + 1000.0 + 150.0 * c) / 192000.0
        env = np.full(n, 1e-3)
        for k in range(6):                                   # bursts of different height and length
            a = int(rng.integers(0, n - 40000))
            env[a:a + int(rng.integers(3000, 40000))] = 10.0 ** rng.uniform(-2.5, -0.3)
        env[n // 2:n // 2 + n // 8] *= np.linspace(1.0, 0.01, n // 8)          # a fade
        env[3 * n // 4:3 * n // 4 + 30000] = 0.0                                # silence
        x[c] = env * np.exp(2j * np.pi * f * t) + 1e-5 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        x[c, n // 3 + 17] += 0.9                                                # a click
    return x


def _engine(qh, nch, mode, form):
    e = qh.RxaEngine(nch)
    for c in range(nch):
        e.SetRXAShiftRun(c, 1); e.SetRXAShiftFreq(c, synth.shift_freq(c)); e.RXANBPSetRun(c, 1); e.SetRXAMode(c, 1)
        e.RXASetPassband(c, 300.0, 3000.0); e.SetRXAAGCMode(c, mode[c % len(mode)])
    e.debug_agc(form)
    return e


@pytest.mark.parametrize("modes", [[3], [1, 2, 3, 4]], ids=["med", "long-slow-med-fast"])
def test_batch_form_is_bit_identical_to_the_sample_loop_and_follows_the_oracle(qh, oracle, modes):
    nch, nblk = 4, 240
    x = _input(nch, nblk)
    calls = [3, 1, 50, 7, 100, 79]
    outs = {}
    for form in (2, 1):
        e = _engine(qh, nch, modes, form)
        ys, pos = [], 0
        for nb in calls:
            ys.append(e.process_host(np.ascontiguousarray(x[:, pos * 1024:(pos + nb) * 1024]))); pos += nb
        outs[form] = np.concatenate(ys, axis=1)
    assert np.array_equal(outs[2], outs[1])
    # the forms alternate between calls on one engine: still the same samples
    e = _engine(qh, nch, modes, 2)
    ys, pos = [], 0
    for k, nb in enumerate(calls):
        e.debug_agc(2 - (k & 1))
        ys.append(e.process_host(np.ascontiguousarray(x[:, pos * 1024:(pos + nb) * 1024]))); pos += nb
    assert np.array_equal(np.concatenate(ys, axis=1), outs[2])
    for c in range(nch):
        ref = _oracle_run(oracle, c, modes[c % len(modes)], x[c])
        assert np.abs(ref).max() > 1e-3
        assert rel_rms(outs[2][c], ref) < 1e-9, (c, rel_rms(outs[2][c], ref))


def _oracle_run(oracle, c, mode, xc):
    o = oracle.WdspChannel(1024, 256, 192000, 48000, 48000)
    o.SetRXAShiftRun(1); o.SetRXAShiftFreq(synth.shift_freq(c)); o.RXANBPSetRun(1); o.SetRXAMode(1)
    o.RXASetPassband(300.0, 3000.0); o.SetRXAAGCMode(mode)
    return o.xrxa(xc)


@pytest.mark.parametrize("modes", [[3], [1, 2, 3, 4]], ids=["med", "long-slow-med-fast"])
def test_time_tiles_follow_the_sample_loop_and_the_oracle(qh, oracle, modes):
    """Long calls (>= 16384 detector samples) run the level detector one lane per tile from the call's start state, check every tile's
    start against its predecessor's end and re-run what had not met (qh_agc_tiled.hpp).  The signal -- bursts, a fade, silence, a
    click -- makes tiles fail; the result is the sample loop's to rounding, the carried state lets short and long calls alternate."""
    nch, nblk = 4, 640
    x = _input(nch, nblk, seed=21)
    calls = [3, 200, 1, 130, 64, 5, 237]                     # long calls between short ones
    outs = {}
    for form in (0, 1):
        e = _engine(qh, nch, modes, form)
        ys, pos = [], 0
        for nb in calls:
            ys.append(e.process_host(np.ascontiguousarray(x[:, pos * 1024:(pos + nb) * 1024]))); pos += nb
        outs[form] = np.concatenate(ys, axis=1)
        if form == 0:
            repairs = e.agc_repairs()
    for c in range(nch):
        assert rel_rms(outs[0][c], outs[1][c]) < 1e-10, (c, rel_rms(outs[0][c], outs[1][c]))
        ref = _oracle_run(oracle, c, modes[c % len(modes)], x[c])
        assert rel_rms(outs[0][c], ref) < 1e-9, (c, rel_rms(outs[0][c], ref))
    assert repairs >= 0


def test_time_tiles_on_a_steady_signal_need_no_repairs(qh, oracle):
    """A channel whose level is steady sits where it sat when the call began: every tile verifies."""
    nch, nblk = 3, 300
    x = np.stack([synth.make_mode_input_numpy("usb", c, nblk * 1024) for c in range(nch)])
    e = _engine(qh, nch, [3], 0)
    y = np.concatenate([e.process_host(np.ascontiguousarray(x[:, :40 * 1024])), e.process_host(np.ascontiguousarray(x[:, 40 * 1024:170 * 1024])),
                        e.process_host(np.ascontiguousarray(x[:, 170 * 1024:]))], axis=1)
    assert e.agc_repairs() == 0
    for c in range(nch):
        ref = _oracle_run(oracle, c, 3, x[c])
        assert rel_rms(y[c], ref) < 1e-9, (c, rel_rms(y[c], ref))


def test_attack_window_changes_and_odd_lengths(qh, oracle):
    """SetRXAAGCAttack moves in_index (wcpAGC.c:120): the ring_max the reference keeps across the change is kept here as well, in
    both forms; calls whose lengths are not multiples of the 64-sample step.  (Against the oracle only up to the first change: a
    longer attack window reads ring slots the reference last wrote 30 721 samples ago and this ring 2 048 samples ago, DESIGN.md
    section 7.)"""
    nch, nblk = 2, 64
    x = _input(nch, nblk, seed=9)
    outs = {}
    for form in (2, 1):
        e = _engine(qh, nch, [3], form)
        ys = [e.process_host(np.ascontiguousarray(x[:, :20 * 1024 + 0]))]
        for c in range(nch): e.SetRXAAGCAttack(c, 4)
        ys.append(e.process_host(np.ascontiguousarray(x[:, 20 * 1024:41 * 1024])))
        for c in range(nch): e.SetRXAAGCAttack(c, 1)
        ys.append(e.process_host(np.ascontiguousarray(x[:, 41 * 1024:])))
        outs[form] = np.concatenate(ys, axis=1)
    assert np.array_equal(outs[2], outs[1])
    assert np.abs(outs[2]).max() > 1e-3
    for c in range(nch):
        ref = _oracle_run(oracle, c, 3, x[c, :20 * 1024])
        assert rel_rms(outs[2][c][:20 * 256], ref) < 1e-9, c


def test_time_tiles_replayed_from_a_graph(qh):
    """The launch sequence of a long call with the AGC tiles (nine kernels per list) replayed from a hipGraph: the same samples."""
    nch, nblk = 3, 80
    x = _input(nch, 4 * nblk, seed=31)
    outs = []
    for replay in (False, True):
        e = _engine(qh, nch, [3, 2], 0)
        e.set_graph_replay(replay)
        import torch
        d = torch.from_numpy(np.ascontiguousarray(x).view(np.float64).copy()).cuda()
        o = torch.empty((nch, 4 * nblk * 256 * 2), dtype=torch.float64, device="cuda")
        ys = []
        for k in range(4):      # the same buffers every call (what a captured sequence needs), new samples copied in
            seg = torch.from_numpy(np.ascontiguousarray(x[:, k * nblk * 1024:(k + 1) * nblk * 1024]).view(np.float64).copy()).cuda()
            d[:, :nblk * 2048] = seg
            torch.cuda.synchronize()                # the engine runs on a stream of its own: the copy has to be there
            e.process_ptr(d.data_ptr(), 4 * nblk * 1024, o.data_ptr(), 4 * nblk * 256, nblk)
            e.synchronize()
            ys.append(o[:, :nblk * 512].cpu().numpy().view(np.complex128).copy())
        outs.append(np.concatenate(ys, axis=1))
        if replay:
            assert e.graph_launches() > 0
    assert np.array_equal(outs[0], outs[1])
    ref = _engine(qh, nch, [3, 2], 0)
    want = np.concatenate([ref.process_host(np.ascontiguousarray(x[:, k * nblk * 1024:(k + 1) * nblk * 1024])) for k in range(4)], axis=1)
    assert np.array_equal(outs[0], want)


@pytest.mark.parametrize("rounds,warm", [(3, 0), (1, 0), (0, 2), (0, 0)], ids=["rounds-3", "rounds-1", "in-order-only-warm-2", "in-order-only"])
def test_boundary_pass_over_super_segments(qh, oracle, monkeypatch, rounds, warm):
    """The boundary pass of a long call as eight super-segments walked at once (QH_AGC_SEGS; the engine picks the number by call length):
    each from the call's start state (behind a warm-up if asked for), the ones that did not start on the true trajectory walked again --
    by the parallel repair rounds (three by default), by one round and the in-order kernel behind it for what a single round leaves, or
    by the in-order kernel alone (round 4's form).  Bursts and fades make that happen; the result is the sample loop's either way."""
    monkeypatch.setenv("QH_AGC_SEGS", "8")
    monkeypatch.setenv("QH_AGC_WARM", str(warm))            # (round 4 ran 400 attack windows of warm-up; 2: segments must miss)
    monkeypatch.setenv("QH_AGC_ROUNDS", str(rounds))
    nch, nblk = 4, 700
    x = _input(nch, nblk, seed=41)
    calls = [300, 2, 398]
    outs = {}
    for form in (0, 1):
        e = _engine(qh, nch, [3, 1, 2, 4], form)
        ys, pos = [], 0
        for nb in calls:
            ys.append(e.process_host(np.ascontiguousarray(x[:, pos * 1024:(pos + nb) * 1024]))); pos += nb
        outs[form] = np.concatenate(ys, axis=1)
        if form == 0:
            print("segments walked again: %d, tiles re-run: %d" % (e.agc_segments_rerun(), e.agc_repairs()))
            assert e.agc_segments_rerun() > 0 and e.agc_repairs() == 0
    for c in range(nch):
        assert rel_rms(outs[0][c], outs[1][c]) < 1e-10, (c, rel_rms(outs[0][c], outs[1][c]))


def test_sixteen_segments_of_a_full_length_call_in_all_four_modes(qh):
    """One call of 2^20 detector samples per channel (the length of the driver's AGC-on leg: the engine itself cuts it into sixteen
    super-segments, nothing forced), eight channels in modes long / slow / med / fast -- the hang states and the fast decay among the
    regimes the walks scan -- on bursts, a fade, silence and a click: the time tiles with their parallel repair rounds against the
    sample loop, and the rounds did have segments to walk again."""
    nch, nblk = 8, 4096
    x = _input(nch, nblk, seed=77)
    outs = {}
    for form in (0, 1):
        e = _engine(qh, nch, [1, 2, 3, 4], form)
        outs[form] = e.process_host(x)
        if form == 0:
            print("segments walked again: %d, tiles re-run: %d" % (e.agc_segments_rerun(), e.agc_repairs()))
            assert e.agc_segments_rerun() > 0 and e.agc_repairs() == 0
    scale = np.abs(outs[1]).max()
    assert scale > 0.1
    for c in range(nch):
        assert rel_rms(outs[0][c], outs[1][c]) < 1e-10, (c, rel_rms(outs[0][c], outs[1][c]))


@pytest.mark.parametrize("dsp_rate,attack_ms,mode", [(96000, 1, 3), (48000, 5, 2), (48000, 10, 4)], ids=["96k", "attack-5ms", "attack-10ms"])
def test_time_tiles_at_other_window_lengths(qh, oracle, dsp_rate, attack_ms, mode):
    """The attack window (4 x rate x tau_attack samples: 384 at 96 kHz, 960 and 1920 with longer attacks set ahead of the stream) sizes
    the sliding maximum, the tiles' halos and the delay of the gain multiply."""
    nch, nblk = 2, 400
    x = _input(nch, nblk, seed=51)
    dsz = 256 * dsp_rate // 48000
    outs = {}
    for form in (0, 1):
        e = qh.RxaEngine(nch, dsp_size=dsz, in_rate=192000, dsp_rate=dsp_rate, out_rate=dsp_rate)
        for c in range(nch):
            e.SetRXAShiftRun(c, 1); e.SetRXAShiftFreq(c, synth.shift_freq(c)); e.RXANBPSetRun(c, 1); e.SetRXAMode(c, 1)
            e.RXASetPassband(c, 300.0, 3000.0); e.SetRXAAGCMode(c, mode); e.SetRXAAGCAttack(c, attack_ms)
        e.debug_agc(form)
        outs[form] = np.concatenate([e.process_host(np.ascontiguousarray(x[:, :190 * 1024])), e.process_host(np.ascontiguousarray(x[:, 190 * 1024:193 * 1024])),
                                     e.process_host(np.ascontiguousarray(x[:, 193 * 1024:]))], axis=1)
    for c in range(nch):
        assert rel_rms(outs[0][c], outs[1][c]) < 1e-10, (c, rel_rms(outs[0][c], outs[1][c]))
    o = oracle.WdspChannel(1024, dsz, 192000, dsp_rate, dsp_rate)
    o.SetRXAShiftRun(1); o.SetRXAShiftFreq(synth.shift_freq(0)); o.RXANBPSetRun(1); o.SetRXAMode(1)
    o.RXASetPassband(300.0, 3000.0); o.SetRXAAGCMode(mode); o.SetRXAAGCAttack(attack_ms)
    ref = o.xrxa(x[0])
    assert np.abs(ref).max() > 1e-3
    assert rel_rms(outs[0][0], ref) < 1e-9, rel_rms(outs[0][0], ref)


@pytest.mark.parametrize("form", [2, 1, 0], ids=["batch", "sample-loop", "tiles-when-long"])
def test_attack_window_changed_mid_stream_reads_the_full_ring(qh, oracle, form):
    """SetRXAAGCAttack in mid-stream (wcpAGC.c:414-420 -> loadWcpAGC :119-120): in_index jumps to out_index + the new attack_buffsize.
    A LONGER window jumps over ring entries that then come out as they are -- zeros while the stream is younger than the ring's
    RB_SIZE = 30721 entries (wcpAGC.h:30-33), the samples of a lap ago later on; a shorter one abandons samples that were waiting.
    The engine keeps the reference's ring in full beside its 2048-entry working ring and takes the window again from it."""
    nch = 3
    #        blocks   attack (ms) set ahead of the stretch: 2 -> 4 while the ring is young, -> 1, -> 8 after more than a lap, -> 3
    plan = [(40, None), (30, 4), (90, 1), (100, 8), (5, 3), (60, None)]
    nblk = sum(p[0] for p in plan)
    x = _input(nch, nblk, seed=33)
    x += 2e-3 * np.exp(2j * np.pi * ((synth.shift_freq(0) + 1500.0) / 192000.0) * np.arange(x.shape[1]))[None, :]      # never silent: stale entries are seen
    e = _engine(qh, nch, [3, 1, 4], form)
    refs = []
    for c in range(nch):
        o = oracle.WdspChannel(1024, 256, 192000, 48000, 48000)
        o.SetRXAShiftRun(1); o.SetRXAShiftFreq(synth.shift_freq(c)); o.RXANBPSetRun(1); o.SetRXAMode(1)
        o.RXASetPassband(300.0, 3000.0); o.SetRXAAGCMode([3, 1, 4][c])
        refs.append(o)
    ys, rs, pos = [], [[] for _ in range(nch)], 0
    for nb, attack in plan:
        if attack is not None:
            for c in range(nch):
                e.SetRXAAGCAttack(c, attack)
                refs[c].SetRXAAGCAttack(attack)
        seg = np.ascontiguousarray(x[:, pos * 1024:(pos + nb) * 1024])
        pos += nb
        ys.append(e.process_host(seg))
        for c in range(nch):
            rs[c].append(refs[c].xrxa(seg[c]))
    y = np.concatenate(ys, axis=1)
    for c in range(nch):
        ref = np.concatenate(rs[c])
        assert np.abs(ref).max() > 1e-3
        # stretch by stretch: the samples right behind each change are the ones the jump decides
        a = 0
        for nb, attack in plan:
            b = a + nb * 256
            assert rel_rms(y[c][a:b], ref[a:b]) < 1e-9, (c, attack, rel_rms(y[c][a:b], ref[a:b]))
            a = b


@pytest.mark.parametrize("together", [False, True], ids=["block-at-a-time", "three-blocks-a-call"])
def test_attack_window_moved_while_the_ring_holds_exact_zeros(qh, oracle, together):
    """xwcpagc looks its window over again only when the sample that leaves is `> 0.0` (wcpAGC.c:197): a window moved while the ring holds
    EXACT zeros -- here behind an EMNR (and with it bp1) just switched on, whose first frames are zeros -- is not looked over until the
    first sample of a lap ago comes out.  Block at a time (the WDSP names' way) the engine's zeros are the reference's.  In a call of
    several DSP blocks bp1's overlap-save tile spans blocks with signal in them and leaves its rounding floor (1e-17 of the signal)
    where the reference's per-block transform gives 0.0 -- the engine used to look the window over at once and ran 4e-4 .. 1e-2 off in
    gain for the few hundred samples of the jump (found by tools/dbg/fuzz_sweep.py (a one-off script, in git history), wide seed 1252; tools/dbg/agc_attack_probe.py (a one-off script, in git history)).
    When a window moves, entries below 1e-13 of the largest of the last RB_SIZE samples are now taken for the zeros they are in the
    reference (agc_rewindow_kernel)."""
    nblk = 230
    x = synth.make_input_numpy(4, nblk * 1024)[2:3].copy()
    e = qh.RxaEngine(1); e.load_emnr_tables()
    o = oracle.WdspChannel(1024, 256, 192000, 48000, 48000)
    for t, lead in ((e, (0,)), (o, ())):
        t.SetRXAShiftRun(*lead, 1); t.SetRXAShiftFreq(*lead, synth.shift_freq(2)); t.RXANBPSetRun(*lead, 1)
        t.SetRXAMode(*lead, 0); t.RXASetPassband(*lead, -3000.0, -300.0); t.SetRXAAGCMode(*lead, 3)

    def run(b0, b1):
        seg = np.ascontiguousarray(x[:, b0 * 1024:b1 * 1024])
        return e.process_host(seg)[0], o.xrxa(seg[0])

    y, r = run(0, 200)
    assert rel_rms(y, r) < 1e-9
    e.RXASetNC(0, 256); o.RXASetNC(256)
    run(200, 201)
    e.SetRXAEMNRRun(0, 1); o.SetRXAEMNRRun(1)
    zeros = 0
    for b0, b1 in ([(201, 204)] if together else [(201, 202), (202, 203), (203, 204)]):
        y, r = run(b0, b1)
        zeros += int((r == 0).sum())
        assert np.abs(y - r).max() <= 1e-9 * max(np.abs(r).max(), 1.0)
    assert zeros >= 256                              # the restatement's AGC is being fed exact zeros
    e.SetRXAAGCAttack(0, 4); o.SetRXAAGCAttack(4)
    y, r = run(204, 207)
    assert np.abs(r).max() > 0.1 and (r[:192] == 0).all() and (y[:192] == 0).all()
    err = np.abs(y - r).max() / np.abs(r).max()
    assert err < 1e-9, err
    e.close()


def test_a_moved_attack_window_takes_only_its_own_channel_off_the_time_tiles(qh, oracle):
    """A channel whose attack window moves in mid-stream keeps the reference's ring_max bookkeeping, stale values and all (wcpAGC.c:197-210
    rescans only when the sample that leaves equals it), so it is stepped by one wavefront from then on; the other channels of the
    engine stay on the time tiles -- and all of them follow the oracle.  A flush puts the channel back."""
    nch, nb = 6, 80                                        # 80 blocks = 20 480 detector samples per call: tiles
    moved = {1: 4, 4: 3}                                   # channel -> attack in ms ahead of the second call (create_rxa: 1 ms)
    modes = [3, 1, 4, 2, 3, 4]
    x = _input(nch, 4 * nb, seed=77)
    x += 2e-3 * np.exp(2j * np.pi * ((synth.shift_freq(0) + 1500.0) / 192000.0) * np.arange(x.shape[1]))[None, :]
    e = _engine(qh, nch, modes, 0)
    refs = []
    for c in range(nch):
        o = oracle.WdspChannel(1024, 256, 192000, 48000, 48000)
        o.SetRXAShiftRun(1); o.SetRXAShiftFreq(synth.shift_freq(c)); o.RXANBPSetRun(1); o.SetRXAMode(1)
        o.RXASetPassband(300.0, 3000.0); o.SetRXAAGCMode(modes[c])
        refs.append(o)
    tiled = []
    for k in range(4):
        if k == 1:
            for c, ms in moved.items():
                e.SetRXAAGCAttack(c, ms); refs[c].SetRXAAGCAttack(ms)
        seg = np.ascontiguousarray(x[:, k * nb * 1024:(k + 1) * nb * 1024])
        y = e.process_host(seg)
        tiled.append(e.agc_tiled_channels())
        for c in range(nch):
            ref = refs[c].xrxa(seg[c])
            assert np.abs(ref).max() > 1e-3
            assert rel_rms(y[c], ref) < 1e-9, (k, c, rel_rms(y[c], ref))
    assert tiled == [nch, nch - 2, nch - 2, nch - 2], tiled
    e.flush()
    e.process_host(np.ascontiguousarray(x[:, :nb * 1024]))
    assert e.agc_tiled_channels() == nch
