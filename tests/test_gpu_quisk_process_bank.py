"""quisk_process_samples for a BANK of receivers (qh_qps_*, include/quiskhip.h group 9b) against the block-level restatement of the
reference's function (oracle qo_ps_*, quisk.c:2289-2742), one restatement per receiver: test tone / inversion / NoiseBlanker ahead
of the panadapter feed and the bank, cFracDecim, the interpolation to the playback rate, process_agc, the FM squelch behind it --
with per-receiver tune frequencies, filters and squelch levels, calls cut into time pieces (the AGC of one beside the filters of the
next) and calls that are not.  And the one-receiver block API against the bank with one receiver: the same kernels (qh_ps_kernels.hpp).
-m gpu."""
import numpy as np
import pytest

from conftest import rel_rms
from quisk_amd import rxfilter

pytestmark = pytest.mark.gpu
NAMES = {3: "USB", 4: "AM", 5: "FM", 1: "CWU", 0: "CWL", 2: "LSB", 7: "DGT-U", 8: "DGT-L", 9: "DGT-IQ", 10: "IMD", 13: "DGT-FM"}
BW = {3: 2700, 4: 6000, 5: 12000, 1: 500, 0: 500, 2: 2700, 7: 3200, 8: 3200, 9: 8000, 10: 2700, 13: 12000}


def _filters(mode, fs, bw=None):
    bw = bw or BW[mode]
    frate = rxfilter.get_filter_rate(fs, mode, bw)
    return rxfilter.make_filter_coef(frate, None, bw, rxfilter.get_filter_center(NAMES[mode], bw))


def _signal(mode, c, n, fs, tune, amp=2.0 ** 22):
    rng = np.random.default_rng(4000 + c)
    t = np.arange(n)
    car = lambda f: np.exp(2j * np.pi * ((f / fs) * t % 1.0))
    noise = amp / 512 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    if mode == 4:
        return amp * (1 + 0.5 * np.cos(2 * np.pi * 1000.0 / fs * t)) * car(tune) + noise
    if mode == 5:
        return amp * car(tune) * np.exp(3j * np.sin(2 * np.pi * 1000.0 / fs * t)) + noise
    return amp * car(tune + 900.0 + 31 * c) + 0.5 * amp * car(tune + 30000.0) + noise


def _refs(oracle, nch, fs, play, mode, tunes, filt, **settings):
    tabs = rxfilter.coefficient_tables()
    out = []
    for c in range(nch):
        r = oracle.OracleQuiskBlock(fs, play, tabs)
        r.set_rx_mode(mode); r.set_tune(tunes[c]); r.set_filters(filt[c][0], filt[c][1], BW[mode])
        for k, v in settings.items():
            getattr(r, k)(*v) if isinstance(v, tuple) else getattr(r, k)(v)
        out.append(r)
    return out


@pytest.mark.parametrize("mode,fs,play,extras", [
    (3, 192000, 48000, {}),
    (3, 192000, 96000, {"add_tone": 10800, "invert_spectrum": 1}),
    (4, 96000, 48000, {"set_noise_blanker": 2}),
    (5, 192000, 48000, {"add_tone": 9000}),
    (3, 111111, 48000, {}),                                # SDR-IQ rate: cFracDecim
    (1, 133333, 192000, {"set_noise_blanker": 1}),         # cFracDecim, then x4
], ids=["usb", "usb-x2-tone-inverted", "am-blanker", "fm-tone", "usb-fracdecim", "cw-fracdecim-x4-blanker"])
def test_bank_against_one_restatement_per_receiver(qh, oracle, mode, fs, play, extras):
    nch = 5
    tunes = [7000 + 1300 * c for c in range(nch)]
    # a filter per receiver (USB: another bandwidth for the odd ones)
    filt = [_filters(mode, fs, 2400 if (mode == 3 and c % 2) else None) for c in range(nch)]
    bank = qh.QuiskProcessBank(nch, fs, mode, BW[mode], playback_rate=play, fft_size=2048, data_width=512)
    refs = _refs(oracle, nch, fs, play, mode, tunes, filt, **extras)
    graphs = []
    for c in range(nch):
        bank.set_tune(c, tunes[c]); bank.set_filters(c, *filt[c])
        g = oracle.OracleGraph(2048, 512, float(fs))
        refs[c].set_graph(g)
        graphs.append(g)
    for k, v in extras.items():
        getattr(bank, k)(v)
    # the reference takes at most 66000 samples per call (SAMP_BUFFER_SIZE, quisk.h:15) and its interpolators stop at 52800 outputs
    # (filter.c:158); the two long calls are cut into 4 pieces here
    sizes = [4096, 4001, 34000, 2222, 33000, 8191]
    n = sum(sizes)
    x = np.stack([_signal(mode, c, n, fs, float(tunes[c])) for c in range(nch)])
    if "set_noise_blanker" in extras:
        x[:, 5000::9973] += 2.0 ** 27
    outs, wants, pos = [], [[] for _ in range(nch)], 0
    for s in sizes:
        seg = x[:, pos:pos + s]
        pos += s
        outs.append(bank.process_host(seg))
        for c in range(nch):
            wants[c].append(refs[c].process(seg[c]))
        assert outs[-1].shape[1] == wants[0][-1].size, (s, outs[-1].shape, wants[0][-1].size)
    y = np.concatenate(outs, axis=1)
    skip = 6 * 1024 * (play // 48000) if mode == 5 else 0     # FM: arg() of rounding-level numbers while the filters fill
    for c in range(nch):
        want = np.concatenate(wants[c])
        assert np.abs(want[skip:]).max() > 2.0 ** 16
        assert rel_rms(y[c][skip:], want[skip:]) < 1e-8, (c, rel_rms(y[c][skip:], want[skip:]))
    pix, sm, cnt = bank.get_graph(1.0, 0.0)
    for c in range(nch):
        rp, rs, rc = graphs[c].get(1.0, 0.0)
        assert cnt == rc and np.abs(pix[c] - rp).max() < 1e-7 and abs(sm[c] - rs) < 1e-7
    bank.close()


def test_pieces_do_not_change_the_stream(qh):
    """One long call as 1, 3 and 8 pieces: the filters carry their state, cFracDecim and the interpolator their phases, the AGC its
    machine (its first call only initialises, so the bank's first call is one piece in every form)."""
    nch, fs, mode, play = 6, 185185, 3, 96000
    filt = _filters(mode, fs)
    x = np.stack([_signal(mode, c, 70000 + 150000, fs, 6000.0 + 500 * c) for c in range(nch)])
    ys = []
    for pieces in (1, 3, 8):
        bank = qh.QuiskProcessBank(nch, fs, mode, BW[mode], playback_rate=play)
        bank.set_pieces(pieces)
        for c in range(nch):
            bank.set_tune(c, 6000 + 500 * c)
        bank.set_filters(-1, *filt)
        ys.append(np.concatenate([bank.process_host(x[:, :70000]), bank.process_host(x[:, 70000:])], axis=1))
        bank.close()
    assert ys[0].shape == ys[1].shape == ys[2].shape and np.abs(ys[0]).max() > 2.0 ** 24
    assert rel_rms(ys[1], ys[0]) < 1e-9 and rel_rms(ys[2], ys[0]) < 1e-9


def test_fm_squelch_per_receiver_behind_the_agc(qh, oracle):
    """set_squelch per receiver (quisk.c:4721): the flag of quisk_process_demodulate mutes the block behind process_agc
    (quisk.c:2712-2728).  Receivers with a carrier open, receivers on noise stay shut -- like the restatement, flag for flag."""
    nch, fs, mode, blk, nblk = 4, 96000, 5, 4800, 12
    filt = [_filters(mode, fs)] * nch
    tunes = [5000, 9000, -7000, 12000]
    bank = qh.QuiskProcessBank(nch, fs, mode, BW[mode])
    refs = _refs(oracle, nch, fs, 48000, mode, tunes, filt)
    levels = [-70.0, -70.0, -40.0, -80.0]                 # dB re full scale (quisk.c:2077-2084); the carriers sit at -54, the noise at -96
    for c in range(nch):
        bank.set_tune(c, tunes[c]); bank.set_filters(c, *filt[c]); bank.set_squelch(c, levels[c])
        refs[c].set_squelch(levels[c])
    n = blk * nblk
    x = np.stack([_signal(mode, c, n, fs, float(tunes[c])) for c in range(nch)])
    rng = np.random.default_rng(3)
    x[1] = 2.0 ** 14 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))        # no carrier on receiver 1
    seen = []
    for k in range(nblk):
        seg = x[:, k * blk:(k + 1) * blk]
        y = bank.process_host(seg)
        flags = bank.squelch_flags()
        for c in range(nch):
            want = refs[c].process(seg[c])
            assert y[c].size == want.size
            assert (flags[c] != 0) == ((refs[c].squelch_flags() & 1) != 0), (k, c)
            if k >= 3:
                assert rel_rms(y[c], want) < 1e-8 or (np.abs(want).max() == 0 and np.abs(y[c]).max() == 0), (k, c)
        seen.append(flags.copy())
    seen = np.array(seen)
    assert list(seen[-1]) == [0, 1, 1, 0]                 # a carrier above the level opens; noise and a carrier below it stay shut
    bank.close()


def test_the_one_receiver_api_is_the_bank_with_one_receiver(qh):
    """qh_quisk_process_samples and a one-receiver bank run the same engines and kernels in the same order: the same bits."""
    fs, play, mode = 133333, 96000, 3
    filt = _filters(mode, fs)
    api = qh.quiskapi
    api.open(fs, playback_rate=play)
    api.set_rx_mode(mode); api.set_tune(8000); api.set_filters(filt[0], filt[1], BW[mode]); api.add_tone(8700); api.set_noise_blanker(1)
    bank = qh.QuiskProcessBank(1, fs, mode, BW[mode], playback_rate=play)
    bank.set_tune(0, 8000); bank.set_filters(0, *filt); bank.add_tone(8700); bank.set_noise_blanker(1)
    x = _signal(mode, 0, 30000, fs, 8000.0)
    pos = 0
    for s in (4096, 7000, 4097, 14807):
        seg = x[pos:pos + s]
        pos += s
        a = api.process(seg)
        b = bank.process_host(seg[None, :])[0]
        assert a.size == b.size and np.array_equal(a.view(np.float64), b.view(np.float64)), s
    api.close(); bank.close()


def test_pipelined_calls_give_the_same_samples(qh):
    """qh_qps_set_pipelined: a call returns with its AGC still running on the bank's second stream and the next call's filters start beside
    it (what this call overwrites -- scratch halves, its stretch of the bank's output rows -- waits for the AGC piece that read it).
    Five calls back to back without a wait in between, every call into rows of its own: the bits of the calls that end on their stream."""
    import torch
    dev = torch.device("cuda", 0)
    # without and with the scratch halves (cFracDecim, interpolation); and two pieces of a length the bank's decimation (240 ksps / 5) does
    # not divide, no scratch: the pieces' stretches of the bank's rows have fixed starts (laid end to end by the counts that
    # came out, piece 0 of the next call reached a sample into the stretch piece 1's AGC was still reading)
    for fs, play, mode, pieces in ((192000, 48000, 3, 4), (185185, 96000, 4, 4), (240000, 48000, 3, 2)):
        nch, n, calls = 8, 1 << 16, 5
        filt = _filters(mode, fs)
        x = torch.from_numpy(np.stack([_signal(mode, c, n * calls, fs, 6000.0 + 500 * c, amp=2.0 ** 18) for c in range(nch)])).to(dev)
        ys = []
        for pipelined in (0, 1):
            st = torch.cuda.Stream(dev)
            bank = qh.QuiskProcessBank(nch, fs, mode, BW[mode], playback_rate=play, stream=st.cuda_stream)
            bank.set_pieces(pieces); bank.set_pipelined(pipelined)
            for c in range(nch):
                bank.set_tune(c, 6000 + 500 * c)
            bank.set_filters(-1, *filt)
            cap = bank.out_capacity(n) + 64            # (the capacity follows the decimators' phases: a few samples up or down from call to call)
            outs = [torch.zeros((nch, cap), dtype=torch.complex128, device=dev) for _ in range(calls)]
            torch.cuda.synchronize(dev)
            got = [bank.process_ptr(x[:, k * n:].data_ptr(), x.shape[1], n, outs[k].data_ptr(), cap) for k in range(calls)]
            bank.synchronize()
            ys.append(torch.cat([outs[k][:, :got[k]] for k in range(calls)], dim=1).cpu())
            bank.close()
        assert ys[0].shape == ys[1].shape and float(ys[0].abs().max()) > 2.0 ** 20
        assert torch.equal(ys[0], ys[1]), (fs, float((ys[0] - ys[1]).abs().max()))


def test_pipelined_calls_whose_piece_layout_changes(qh):
    """in pipelined mode a piece waited for the AGC event of its own parity only, which covers what it overwrites only
    while consecutive calls cut the bank's rows and the scratch halves alike.  Here the block length grows, shrinks, and the piece
    count moves between 1, 3 and 4 from call to call (a call that ends on parity 1 followed by one whose piece 0 is longer): the bits
    of the calls that end on their stream."""
    import torch
    dev = torch.device("cuda", 0)
    plan = [(1 << 15, 4), (3 << 14, 3), (1 << 16, 1), (1 << 15, 4), (5 << 13, 3), (1 << 16, 4), (1 << 14, 1), (1 << 16, 3)]
    total = sum(n for n, _ in plan)
    for fs, play, mode in ((192000, 48000, 3), (185185, 96000, 4)):
        nch = 8
        filt = _filters(mode, fs)
        x = torch.from_numpy(np.stack([_signal(mode, c, total, fs, 6000.0 + 500 * c, amp=2.0 ** 18) for c in range(nch)])).to(dev)
        ys = []
        for pipelined in (0, 1):
            st = torch.cuda.Stream(dev)
            bank = qh.QuiskProcessBank(nch, fs, mode, BW[mode], playback_rate=play, stream=st.cuda_stream)
            bank.set_pipelined(pipelined)
            for c in range(nch):
                bank.set_tune(c, 6000 + 500 * c)
            bank.set_filters(-1, *filt)
            outs, got, pos = [], [], 0
            torch.cuda.synchronize(dev)
            for n, pieces in plan:
                bank.set_pieces(pieces)
                cap = bank.out_capacity(n) + 64
                outs.append(torch.empty((nch, cap), dtype=torch.complex128, device=dev))      # (no fill: it would run on torch's stream, beside the bank's)
                got.append(bank.process_ptr(x[:, pos:].data_ptr(), x.shape[1], n, outs[-1].data_ptr(), cap))
                pos += n
            bank.synchronize()
            ys.append(torch.cat([outs[k][:, :got[k]] for k in range(len(plan))], dim=1).cpu())
            bank.close()
        assert ys[0].shape == ys[1].shape and float(ys[0].abs().max()) > 2.0 ** 20
        assert torch.equal(ys[0], ys[1]), (fs, float((ys[0] - ys[1]).abs().max()))


def test_set_tune_all_is_the_per_receiver_setter_in_one_launch(qh):
    """qh_qps_set_tune_all / qh_qrx_set_tune_all: every receiver's set_tune (quisk.c:4702) with one launch per table for the whole bank
    instead of one per receiver (round 4's trace: 2 x 256 launches and ~95 ms to tune 256 receivers).  The same arithmetic per
    receiver: two banks, one tuned receiver by receiver and one tuned at once, give the same bits -- at the start of the stream and
    when every receiver is retuned in mid-stream (the raw history is re-expressed for the new phase law in both forms)."""
    for mode, fs, play in ((3, 192000, 48000), (5, 96000, 48000)):
        nch, n = 7, 1 << 14
        filt = _filters(mode, fs)
        x = np.stack([_signal(mode, c, 3 * n, fs, 6000.0 + 700 * c, amp=2.0 ** 18) for c in range(nch)])
        tunes = [[6000 + 700 * c for c in range(nch)], [-9000 + 1100 * c for c in range(nch)], [6000 + 700 * c for c in range(nch)]]
        outs = []
        for at_once in (False, True):
            bank = qh.QuiskProcessBank(nch, fs, mode, BW[mode], playback_rate=play)
            bank.set_filters(-1, *filt)
            ys = []
            for k in range(3):
                if at_once:
                    bank.set_tune_all(tunes[k])
                else:
                    for c in range(nch):
                        bank.set_tune(c, tunes[k][c])
                ys.append(bank.process_host(np.ascontiguousarray(x[:, k * n:(k + 1) * n])))
            outs.append(np.concatenate(ys, axis=1))
            bank.close()
        assert outs[0].shape == outs[1].shape and np.abs(outs[0]).max() > 2.0 ** 10
        assert np.array_equal(outs[0].view(np.float64), outs[1].view(np.float64)), (mode, np.abs(outs[0] - outs[1]).max())
    # the receiver bank alone (qh_qrx_*)
    nch, n, fs, mode = 5, 1 << 14, 192000, 3
    x = np.stack([_signal(mode, c, n, fs, 5000.0 + 900 * c, amp=2.0 ** 18) for c in range(nch)])
    outs = []
    for at_once in (False, True):
        bank = qh.QuiskRxBank(nch, fs, mode, BW[mode])
        bank.set_filters(-1, *_filters(mode, fs))
        if at_once:
            bank.set_tune_all([5000 + 900 * c for c in range(nch)])
        else:
            for c in range(nch):
                bank.set_tune(c, 5000 + 900 * c)
        outs.append(bank.process_host(x))
        bank.close()
    assert np.array_equal(outs[0].view(np.float64), outs[1].view(np.float64))
