"""qh_rxa_set_graph_replay: a block-at-a-time caller gets the launch sequence of qh_rxa_process replayed from hipGraphs.
The replayed engine must produce the plain engine's bits through every mode, through setters that arrive after an odd
and an even number of replayed blocks, and through a flush.  -m gpu."""
import numpy as np
import pytest
import torch

from quisk_amd import synth

pytestmark = pytest.mark.gpu

MODES = [1, 6, 5, 10, 0, 3]          # USB, AM, FM, SAM, LSB, CWL (wdsp/RXA.h rxaMode)


def _engine(qh, replay):
    e = qh.RxaEngine(len(MODES))
    for ch, m in enumerate(MODES):
        e.SetRXAMode(ch, m)
        e.SetRXAShiftRun(ch, 1)
        e.SetRXAShiftFreq(ch, synth.shift_freq(ch))
        e.RXANBPSetRun(ch, 1)
        e.RXASetPassband(ch, *((-3000.0, -300.0) if m in (0, 3) else (300.0, 3000.0) if m == 1 else (-4000.0, 4000.0)))
        e.SetRXAAGCMode(ch, 3)
    e.enable_meters(True)
    e.set_graph_replay(replay)
    return e


def _script(e, b):
    # setters land after 7 (odd), 18 and 29 (odd again) blocks, a flush after 24
    if b == 7:
        e.SetRXAShiftFreq(0, 9000.0)
        e.RXASetPassband(0, 200.0, 2500.0)
    if b == 18:
        e.SetRXAMode(1, 1)
        e.SetRXAAGCMode(2, 0)
        e.SetRXAAGCFixed(2, 10.0)
    if b == 24:
        e._L.qh_rxa_flush(e._h)
    if b == 29:
        e.SetRXAPanelGain1(-1, 0.5)


def test_replayed_blocks_are_bit_identical_to_plain_launches(qh):
    nb = 40
    x = synth.make_input_numpy(len(MODES), nb * 1024)
    dev = torch.device("cuda:0")
    outs = []
    for replay in (False, True):
        e = _engine(qh, replay)
        d_in = torch.zeros((len(MODES), 1024), dtype=torch.complex128, device=dev)
        d_out = torch.zeros((len(MODES), 256), dtype=torch.complex128, device=dev)
        y = np.zeros((len(MODES), nb * 256), dtype=np.complex128)
        for b in range(nb):
            _script(e, b)
            d_in.copy_(torch.from_numpy(x[:, b * 1024:(b + 1) * 1024]))
            torch.cuda.synchronize()
            e.process_ptr(d_in.data_ptr(), 1024, d_out.data_ptr(), 256, 1)
            e.synchronize()
            y[:, b * 256:(b + 1) * 256] = d_out.cpu().numpy()
        outs.append(y)
        if replay:
            # each of the five parameter epochs costs one plain block and two captures (which also launch)
            assert e.graph_launches() >= nb - 5
            meters = [e.GetRXAMeter(ch, 0) for ch in range(len(MODES))]
        else:
            assert e.graph_launches() == 0
            ref_meters = [e.GetRXAMeter(ch, 0) for ch in range(len(MODES))]
        e.close()
    assert np.all(np.isfinite(outs[0])) and np.abs(outs[0]).max() > 1e-3
    assert np.array_equal(outs[0], outs[1])
    assert meters == ref_meters


def test_changed_arguments_drop_the_graphs(qh):
    # a caller that alternates between two output buffers never repeats its arguments: every block runs plain, and right
    e, p = _engine(qh, True), _engine(qh, False)
    dev = torch.device("cuda:0")
    x = synth.make_input_numpy(len(MODES), 12 * 1024)
    d_in = torch.zeros((len(MODES), 1024), dtype=torch.complex128, device=dev)
    d_out = [torch.zeros((len(MODES), 256), dtype=torch.complex128, device=dev) for _ in range(3)]
    for b in range(12):
        d_in.copy_(torch.from_numpy(x[:, b * 1024:(b + 1) * 1024]))
        torch.cuda.synchronize()
        e.process_ptr(d_in.data_ptr(), 1024, d_out[b % 2].data_ptr(), 256, 1)
        p.process_ptr(d_in.data_ptr(), 1024, d_out[2].data_ptr(), 256, 1)
        e.synchronize(); p.synchronize()
        assert torch.equal(d_out[b % 2], d_out[2])
    assert e.graph_launches() == 0


def test_buffers_grown_through_another_entry_point_drop_the_graphs(qh):
    # captured graphs hold the addresses of the engine's intermediate buffers; process_host (which does not go through the
    # replay key) with a larger block count re-allocates them: the next replayed block must not run the stale graph
    e, p = _engine(qh, True), _engine(qh, False)
    dev = torch.device("cuda:0")
    nb = 10
    x = synth.make_input_numpy(len(MODES), (nb + 16) * 1024)
    d_in = torch.zeros((len(MODES), 1024), dtype=torch.complex128, device=dev)
    d_out = torch.zeros((len(MODES), 256), dtype=torch.complex128, device=dev)
    d_ref = torch.zeros((len(MODES), 256), dtype=torch.complex128, device=dev)
    pos = 0
    for b in range(nb):
        if b == 5:      # 16 blocks in one host call on both engines
            big = np.ascontiguousarray(x[:, pos:pos + 16 * 1024])
            ya, yb = e.process_host(big), p.process_host(big)
            assert np.array_equal(ya, yb)
            pos += 16 * 1024
        d_in.copy_(torch.from_numpy(x[:, pos:pos + 1024]))
        pos += 1024
        torch.cuda.synchronize()
        e.process_ptr(d_in.data_ptr(), 1024, d_out.data_ptr(), 256, 1)
        p.process_ptr(d_in.data_ptr(), 1024, d_ref.data_ptr(), 256, 1)
        e.synchronize(); p.synchronize()
        assert torch.equal(d_out, d_ref), "block %d" % b
    assert e.graph_launches() > 0
