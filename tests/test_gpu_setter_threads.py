"""Setters from another thread (section 8(b): Quisk's GUI thread sets parameters while its sound thread runs the blocks; WDSP
serialises them with csDSP): every engine call takes the engine's lock, a setter edits the host-side configuration only, and
the next process call uploads it before the block is enqueued -- so a parameter changes on a block boundary, never inside a
block.  One thread flips the panel gain of a running engine as fast as it can, another runs 300 single-block calls; every
output block must be the reference block scaled by exactly one of the two gains.  -m gpu."""
import threading

import numpy as np
import pytest

from quisk_amd import synth

pytestmark = pytest.mark.gpu


def test_setter_thread_changes_land_on_block_boundaries(qh):
    nblk = 300
    x = synth.make_input_numpy(1, nblk * 1024)

    def engine():
        e = qh.RxaEngine(1)
        e.SetRXAShiftRun(0, 1); e.SetRXAShiftFreq(0, synth.shift_freq(0)); e.RXANBPSetRun(0, 1); e.SetRXAMode(0, 1)
        e.RXASetPassband(0, 300.0, 3000.0); e.SetRXAAGCMode(0, 0); e.SetRXAAGCFixed(0, 0.0)
        return e
    ref = engine()
    ref.SetRXAPanelGain1(0, 1.0)
    want = ref.process_host(x)[0].reshape(nblk, 256)
    e = engine()
    gains = (1.0, 3.0)
    stop = threading.Event()
    flips = [0]

    def setter():
        k = 0
        while not stop.is_set():
            e.SetRXAPanelGain1(0, gains[k & 1])
            e.SetRXAPanelGain2(0, 1.0, 1.0)
            k += 1
        flips[0] = k
    th = threading.Thread(target=setter)
    th.start()
    try:
        got = np.stack([e.process_host(np.ascontiguousarray(x[:, b * 1024:(b + 1) * 1024]))[0] for b in range(nblk)])
    finally:
        stop.set()
        th.join()
    assert flips[0] > 50
    seen = set()
    for b in range(4, nblk):                    # (the first blocks are the filters' start-up: too small to tell a gain)
        r = np.vdot(want[b], got[b]).real / np.vdot(want[b], want[b]).real
        g = min(gains, key=lambda v: abs(v - r))
        assert np.abs(got[b] - g * want[b]).max() < 1e-9 * np.abs(want[b]).max() * g, (b, r)
        seen.add(g)
    assert seen == set(gains)                   # both values were picked up while the blocks ran
