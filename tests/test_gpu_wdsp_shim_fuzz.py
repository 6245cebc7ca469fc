"""Seeded walks over Quisk's hand-off to WDSP (wdspFexchange0, quisk_wdsp.c:24-69, in front of fexchange0's double rings and slews,
wdsp/iobuffs.c:464-516) with the caller CHANGING SIDES from call to call: host pointers (wdspFexchange0) or samples that are already on
the GPU (qh_wdsp_fexchange0_device) -- the shim's ring, both iobuffs rings and the up-slew's data-dependent trigger then move between
host and device memory in mid-stream -- ragged call lengths (0, 1, primes, several blocks), in_use dropped and raised, and the SetRXA*
names between calls on a channel whose DSP blocks are replayed from hipGraphs.  Against the restated shim in front of the restated
channel with the same setters.  -m gpu."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from conftest import rel_rms
from quisk_amd import synth
from test_gpu_rxa_fuzz import _Without, _apply, _apply2
from test_gpu_wdsp_device_handoff import CLIP32, _dev_call, _host_call
from test_gpu_wdsp_names_fuzz import _Names, _NoEmnr

pytestmark = pytest.mark.gpu
GEOMETRY = [(1024, 192000), (256, 48000), (64, 192000), (512, 96000), (2048, 192000), (128, 48000)]      # in_size, in_rate (dsp 256 at 48 kHz)


@pytest.mark.parametrize("seed", list(range(1, 19)) + list(range(5001, 5007)))
def test_random_walk_over_the_hand_off_with_the_caller_changing_sides(qh, oracle, seed):
    lib = qh.load()
    rng = np.random.default_rng(61000 + seed)
    in_size, in_rate = GEOMETRY[(seed - 1) % len(GEOMETRY)]
    ch = 8 + seed % 8
    D = C.c_double
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    lib.OpenChannel(ch, in_size, 256, in_rate, 48000, 48000, 0, 1, D(0.010), D(0.025), D(0.0), D(0.010), 1)
    assert lib.qh_wdsp_status() == 0, lib.qh_last_error()
    chan = oracle.WdspChannel(in_size, 256, in_rate, 48000, 48000)
    L = oracle.lib()

    def own_fx(pin, pout):
        err = C.c_int(0)
        L.wo_fexchange0(chan.h, pin, pout, err)
        return err.value
    shim = oracle.OracleWdspShim(own_fx)
    names, o = _Names(lib, ch, 256), _NoEmnr(chan, 256)
    # QH_TWIN=1 (diagnostics): a second restated channel fed 1e-13 of noise per sample -- how far the restatement is from itself
    twin = oracle.WdspChannel(in_size, 256, in_rate, 48000, 48000) if os.environ.get("QH_TWIN") else None

    def twin_fx(pin, pout):
        err = C.c_int(0)
        L.wo_fexchange0(twin.h, pin, pout, err)
        return err.value
    tshim = oracle.OracleWdspShim(twin_fx) if twin else None
    targets = [(names, ()), (o, ())] + ([(_NoEmnr(twin, 256), ())] if twin else [])
    tw, pert = [], np.random.default_rng(5)
    try:
        for t, _ in targets:
            t.SetRXAShiftRun(1); t.SetRXAShiftFreq(float(synth.shift_freq(seed % 4))); t.RXANBPSetRun(1); t.SetRXAMode(1)
            t.RXASetPassband(300.0, 3000.0); t.SetRXAAGCMode(0)
        lib.qh_wdsp_set_parameter(ch, in_size, 1); shim.set_parameter(in_size=in_size, in_use=1)
        if twin: tshim.set_parameter(in_size=in_size, in_use=1)
        sizes = [int(rng.choice([0, 1, 2, 5, 17, 100, 333, 777, 1023, 1024, 1025, 3000, 4096, 6000, 2048 * 4 + 17])) for _ in range(40)]
        x = synth.make_input_numpy(4, sum(sizes) * 192000 // in_rate + 8)[seed % 4][::192000 // in_rate][:sum(sizes)].copy() * CLIP32
        lead = int(rng.choice([0, 0, 131, 700]))
        x[:lead] = 0.0                                     # the up-slew waits for the first non-zero sample (iobuffs.c:104-113)
        got, want, lms, pos, sides, in_use, log = [], [], False, 0, [0, 0], 1, []
        notches = [0]
        for k, n in enumerate(sizes):
            if k and rng.integers(0, 3) == 0:
                if seed > 5000 and rng.integers(0, 2):      # (from 5000 up: the second menu, the FM detector's setters among it)
                    done = _apply2(rng, [(_Without(t, "SetRXAAMDRun"), lead) for t, lead in targets], notches, fm=True)
                else:
                    done = _apply(rng, [(_Without(t, "SetRXAAMDRun"), lead) for t, lead in targets] if seed > 5000 else targets)
                    notches[0] += sum(1 for d in done if d[0] == "RXANBPAddNotch")
                lms = lms or any(d[0] in ("SetRXAANFRun", "SetRXAANRRun") and d[1] for d in done)
                log.append((k, done))
            if k and rng.integers(0, 12) == 0:             # in_use dropped or raised: the shim rewinds its ring (quisk_wdsp.c:32-37)
                in_use ^= 1
                lib.qh_wdsp_set_parameter(ch, -1, in_use); shim.set_parameter(in_use=in_use)
                if twin: tshim.set_parameter(in_use=in_use)
                log.append((k, "in_use", in_use))
            seg = np.ascontiguousarray(x[pos:pos + n])
            pos += n
            side = int(rng.integers(0, 2))
            sides[side] += 1
            y = _dev_call(lib, ch, seg, in_size, dev, stream) if side else _host_call(lib, ch, seg, in_size)
            assert lib.qh_wdsp_status() == 0, (seed, k, lib.qh_last_error())
            work = np.zeros(n + 2 * in_size, dtype=np.complex128); work[:n] = seg
            m = shim.fexchange0(work, n)
            assert y.size == m, (seed, k, n, "device" if side else "host", y.size, m)
            got.append(y); want.append(work[:m].copy())
            if twin:
                w2 = np.zeros(n + 2 * in_size, dtype=np.complex128); w2[:n] = seg * (1.0 + 1e-13 * pert.standard_normal(n))
                tw.append(w2[:tshim.fexchange0(w2, n)].copy())
            log.append((k, n, "device" if side else "host", float(np.abs(y - work[:m]).max()) if m else 0.0))
        y, r = np.concatenate(got), np.concatenate(want)
        if twin:
            print("seed %d: engine %.3e, the restatement against its twin %.3e" % (seed, rel_rms(y, r), rel_rms(np.concatenate(tw), r)), flush=True)
        assert min(sides) >= 8 and np.all(np.isfinite(r))
        if np.abs(r).max() > 1e-3:
            assert rel_rms(y, r) < (1e-4 if lms else 1e-6), (seed, (in_size, in_rate), rel_rms(y, r), float(np.abs(r).max()), log)
    finally:
        lib.qh_wdsp_set_parameter(ch, 0, 0)
        lib.wdspFexchange0(ch, None, 0)                    # not in use: the shim rewinds its ring, nothing is left for the next test
        lib.CloseChannel(ch)


def test_a_channel_reopened_with_another_block_size_starts_its_ring_again(qh, oracle):
    """Found by the walks above run one after another on one channel number: the reference's shim only ever grows its ring
    (quisk_wdsp.c:44-49), so after a life with in_size 128 and a long call (67 blocks: 8576 samples) a life with in_size 256 keeps a ring
    of 33.5 blocks -- Rindex steps past its end and fexchange0 reads beyond the allocation (quisk_wdsp.c:57-60).  This library starts
    the ring again when in_size changes; the second life must be what a fresh shim gives."""
    lib = qh.load()
    ch, D = 12, C.c_double
    L = oracle.lib()
    for in_size, sizes in ((128, [8209, 100, 1000]), (256, [6000, 3000, 777, 6000, 4096, 6000])):
        lib.OpenChannel(ch, in_size, 256, 48000, 48000, 48000, 0, 1, D(0.010), D(0.025), D(0.0), D(0.010), 1)
        assert lib.qh_wdsp_status() == 0, lib.qh_last_error()
        chan = oracle.WdspChannel(in_size, 256, 48000, 48000, 48000)

        def own_fx(pin, pout, chan=chan):
            err = C.c_int(0)
            L.wo_fexchange0(chan.h, pin, pout, err)
            return err.value
        shim = oracle.OracleWdspShim(own_fx)
        try:
            for t in (_Names(lib, ch, 256), chan):
                t.RXANBPSetRun(1); t.SetRXAMode(1); t.RXASetPassband(300.0, 3000.0); t.SetRXAAGCMode(0)
            lib.qh_wdsp_set_parameter(ch, in_size, 1); shim.set_parameter(in_size=in_size, in_use=1)
            x = synth.make_input_numpy(1, sum(sizes) * 4)[0][::4].copy() * CLIP32
            pos = 0
            for n in sizes:
                seg = np.ascontiguousarray(x[pos:pos + n])
                pos += n
                y = _host_call(lib, ch, seg, in_size)
                work = np.zeros(n + 2 * in_size, dtype=np.complex128); work[:n] = seg
                m = shim.fexchange0(work, n)
                assert y.size == m
                if m:
                    assert np.abs(y - work[:m]).max() <= 1e-6 * max(np.abs(work[:m]).max(), 1.0), (in_size, n)
        finally:
            lib.qh_wdsp_set_parameter(ch, -1, 0)
            lib.wdspFexchange0(ch, None, 0)
            lib.CloseChannel(ch)
