"""Every leg of the driver's line (bench.py `other_configs`) at the CALL SHAPE it times, built from tools/bench_configs.py's own
setup functions -- same setters, sizes, buffers and replay mode, so that what is timed and what is checked cannot drift apart:

  config 3        64 ch x 2^20: FirBank + Panadapter on one stream, and the fused qh_pan_feed_decimate (panfir16k_kernel)
  config 4        256 ch x 2^22, mode by c mod 3 = USB / AM / FM, two streams inside the engine, launches replayed from a hipGraph
  config 5        one 2^26-sample fp32 stream through the 4 + 4 half-band cascade, the 245-tap / 5 and the bandpass
  Quisk-native    256 receivers x 2^20, USB / AM / FM, without and with process_agc
  config2_agc_on  256 ch x 2^22 with SetRXAAGCMode 3 (super-segments, warm-ups, repairs)

Each: >= 8 channels spread over the range, sample for sample against the oracle over the WHOLE call (detectors and AGCs from their
first sample), and ALL channels against the same stream fed in uneven pieces (other tile boundaries, the short-call kernels).
Time-tiled detectors, segment grids, super-segments, stream splits and cascade cuts change behaviour with call length and channel
count; these are the shapes that switch them on.  The re-run counters are printed.  -m gpu."""
import importlib.util
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch          # before libquiskhip: one HIP runtime per process (torch's), as in bench.py
import pytest

from conftest import rel_rms, ROOT
from quisk_amd import synth, rxfilter

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bc():
    spec = importlib.util.spec_from_file_location("bench_configs", os.path.join(ROOT, "tools", "bench_configs.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture()
def dev():
    d = torch.device("cuda", 0)
    torch.cuda.set_device(d)
    yield d
    torch.cuda.synchronize(d)
    torch.cuda.empty_cache()


def spread(nch, k=8):
    """k channels over the range, both ends included"""
    return sorted({int(round(i * (nch - 1) / (k - 1))) for i in range(k)})


def pmap(fn, items, workers=8):
    """the oracle channels side by side on the host cores (ctypes releases the GIL inside the C call)"""
    with ThreadPoolExecutor(max_workers=workers) as ex:
        return list(ex.map(fn, items))


def dev_max_diff(a, b):
    return float((a - b).abs().max().item())


# ------------------------------------------------------------------------------------------------------------------ config 3
def test_config3_two_engines_and_fused_at_the_timed_shape(qh, oracle, bc, dev):
    """64 ch x 2^20 per call, two calls (the bench repeats the call on one engine: FIR history and the |X| sums carry over)."""
    L = bc.setup_config3(torch, qh, dev)
    nch, n, m = L.nch, L.n, L.n // 32
    assert (nch, n) == (64, 1 << 20)
    ya = torch.empty((2, nch, m), dtype=torch.complex128, device=dev)
    yf = torch.empty_like(ya)
    sync = lambda: torch.cuda.synchronize(dev)       # the engines run on streams of their own (stream handle 0), torch on its null stream
    for k in range(2):
        L.step_both()
        sync()
        ya[k].copy_(L.y)
        assert L.step_fused() == m
        sync()
        yf[k].copy_(L.y_fused)
    sync()
    chans = spread(nch, 8)
    xs = {c: L.x[c].cpu().numpy() for c in chans}
    pa, sa, ca = L.pan.get_graph(1.0, 0.0)
    pf, sf, cf = L.fused.get_graph(1.0, 0.0)
    assert ca == cf == 2 * n // 16384

    def check(c):
        x2 = np.concatenate([xs[c], xs[c]])
        want = oracle.OracleFir(L.taps).cDecimate(x2, 32)
        g = oracle.OracleGraph(16384, 1024, L.fs)
        g.feed(x2)
        rp, rs, rc = g.get(1.0, 0.0)
        return c, want, rp, rs, rc
    for c, want, rp, rs, rc in pmap(check, chans):
        got_a = torch.cat([ya[0, c], ya[1, c]]).cpu().numpy()
        got_f = torch.cat([yf[0, c], yf[1, c]]).cpu().numpy()
        assert want.size == 2 * m
        ea, ef = rel_rms(got_a, want), rel_rms(got_f, want)
        assert ea < 1e-12 and ef < 1e-11, (c, ea, ef)
        assert np.abs(got_a - want).max() < 1e-11 * np.abs(want).max(), c        # no tile anywhere in the call is off
        assert np.abs(got_f - want).max() < 1e-10 * np.abs(want).max(), c
        assert rc == ca
        assert np.abs(pa[c] - rp).max() < 1e-8 and np.abs(pf[c] - rp).max() < 1e-8, c
        assert abs(sa[c] - rs) < 1e-8 and abs(sf[c] - rs) < 1e-8, c
    # all 64 channels: the same two calls' stream in uneven pieces through fresh engines (FIR: any length; the panadapter and
    # the fused form: whole 16384-sample blocks per call here, the shape its callers use)
    s = torch.cuda.current_stream(dev).cuda_stream
    bank2 = qh.FirBank(nch, L.taps, 32, stream=s)
    pan2 = qh.Panadapter(nch, 16384, 1024, L.fs, stream=s)
    fus2 = qh.Panadapter(nch, 16384, 1024, L.fs, stream=s)
    fus2.attach_fir(L.taps, 32)
    yb = torch.empty_like(ya)
    yg = torch.empty_like(ya)
    sync()
    for k in range(2):
        pos = 0
        for blocks in (1, 37, 3, 23):
            cnt = blocks * 16384
            got = bank2.process_ptr(L.x.data_ptr() + 16 * pos, n, cnt, yb[k].data_ptr() + 16 * (pos // 32), m)
            assert got == cnt // 32
            pan2.feed_ptr(L.x.data_ptr() + 16 * pos, n, cnt)
            assert fus2.feed_decimate_ptr(L.x.data_ptr() + 16 * pos, n, cnt, yg[k].data_ptr() + 16 * (pos // 32), m) == cnt // 32
            pos += cnt
        assert pos == n
    torch.cuda.synchronize(dev)
    scale = float(ya.abs().max().item())
    assert dev_max_diff(ya, yb) < 1e-12 * scale
    assert dev_max_diff(yf, yg) < 1e-10 * scale
    p2, s2, c2 = pan2.get_graph(1.0, 0.0)
    p3, s3, c3 = fus2.get_graph(1.0, 0.0)
    assert c2 == c3 == ca
    assert np.abs(p2 - pa).max() < 1e-9 and np.abs(p3 - pf).max() < 1e-9


# ------------------------------------------------------------------------------------------------------------------ config 4
class _Ch:
    """the oracle's WdspChannel behind the engine's (channel, ...) setter signature, so that bench_configs' setter functions drive both"""

    def __init__(self, o):
        self.o = o

    def __getattr__(self, name):
        f = getattr(self.o, name)
        return lambda c, *a: f(*a)


def test_config4_mixed_modes_graph_replay_at_the_timed_shape(qh, oracle, bc, dev):
    """256 ch x 2^22 per call, USB / AM / FM by c mod 3, the engine's two streams, the launch sequence captured and REPLAYED: five
    calls with the bench's own pointers (the replay key holds them) -- plain, capture, capture of the other ping-pong state, replay,
    replay -- one continuous 5 x 2^22-sample stream per channel, each call's output copied aside before the next.

    The detectors are switched in on primed filters, as tests/test_gpu_acquisition.py does (a detector that starts on the 1e-19
    rounding noise of empty filters is ill-conditioned in both implementations): 12 blocks as USB, then bench_configs' own
    c4_set_modes on both sides, then the timed call shape.  From there: nine channels (three of each mode, spread over the range)
    against the oracle from the FIRST detector sample over all three calls, and all 256 against a second engine fed the same
    stream in uneven pieces without replay."""
    L = bc.setup_config4(torch, qh, dev, modes_now=False)
    nch, nblk, n_in, n_out = L.nch, L.nblk, L.n_in, L.n_out
    assert (nch, nblk) == (256, 4096)
    prime = 12
    ncall = 5
    e = L.eng
    yp = torch.empty((nch, prime * 256), dtype=torch.complex128, device=dev)
    ys = []
    torch.cuda.synchronize(dev)
    # the priming stretch is the head of the same input buffer; the calls then take the whole buffer again and again
    e.process_ptr(L.x.data_ptr(), n_in, yp.data_ptr(), prime * 256, prime)
    for c in range(nch):
        bc.c4_set_modes(e, c)
    e.set_graph_replay(True)
    launches = []
    for k in range(ncall):
        L.step()                                           # bench_configs' own call: same pointers every time
        e.synchronize()
        torch.cuda.synchronize(dev)
        ys.append(L.y.clone())
        launches.append(e.graph_launches())
    assert launches[-1] >= 3 and launches[-1] > launches[-2] > launches[-3]          # the last calls ran from the graph
    print("config 4 shape: graph launches %d, FM tiles re-run %d" % (e.graph_launches(), e.pll_repairs()))
    chans = [0, 1, 2, 126, 127, 128, 253, 254, 255]        # c mod 3 = 0, 1, 2 three times: USB, AM, FM at both ends and the middle
    xs = {c: L.x[c].cpu().numpy() for c in chans}
    got = {c: np.concatenate([ys[k][c].cpu().numpy() for k in range(ncall)]) for c in chans}
    gotp = {c: yp[c].cpu().numpy() for c in chans}

    def check(c):
        o = oracle.WdspChannel(1024, 256, 192000, 48000, 48000)
        a = _Ch(o)
        bc.c4_common(a, c, synth.shift_freq(c))
        o.SetRXAMode(1); o.RXASetPassband(300.0, 3000.0)
        wp = o.xrxa(xs[c][:prime * 1024])
        bc.c4_set_modes(a, c)
        want = np.concatenate([o.xrxa(xs[c]) for _ in range(ncall)])
        return c, wp, want
    for c, wp, want in pmap(check, chans, workers=9):
        m = bc.C4_MODES[c % 3]
        assert rel_rms(gotp[c], wp) < 1e-9, c
        assert np.abs(want).max() > 0.05
        first = slice(0, 40 * 256)                          # the pull-in behind the switch
        # USB: the oracle's shift oscillator is the reference's rotation recurrence (wdsp/shift.c:66-80), the kernel's a closed form
        # of the sample index; they part by ~1e-16 per sample (hard part 2): 2e-9 after the 2 x 10^7 samples here
        tol = 1e-8 if m == 1 else 1e-6
        e1, e2 = rel_rms(got[c][first], want[first]), rel_rms(got[c], want)
        print("config 4 shape: channel %d mode %d rel RMS first 40 blocks %.2e, whole 5 x 2^20 outputs %.2e, max abs %.2e" % (c, m, e1, e2, np.abs(got[c] - want).max()))
        assert e1 < tol and e2 < tol, (c, m, e1, e2)
        assert np.abs(got[c] - want).max() < 100 * tol * np.abs(want).max(), (c, m)       # no tile anywhere is off
    # all 256 channels: a second engine, same priming, the stream in uneven pieces, launched directly
    e2 = qh.RxaEngine(nch, stream=torch.cuda.current_stream(dev).cuda_stream)
    for c in range(nch):
        bc.c4_common(e2, c, synth.shift_freq(c))
        e2.SetRXAMode(c, 1); e2.RXASetPassband(c, 300.0, 3000.0)
    y2p = torch.empty_like(yp)
    torch.cuda.synchronize(dev)
    e2.process_ptr(L.x.data_ptr(), n_in, y2p.data_ptr(), prime * 256, prime)
    for c in range(nch):
        bc.c4_set_modes(e2, c)
    y2 = torch.empty((nch, n_out), dtype=torch.complex128, device=dev)
    pieces = [(1000, 3000, 96), (4000, 7, 89), (2048, 2048), (1, 4095), (300, 3796)]
    worst = 0.0
    for k in range(ncall):
        pos = 0
        for nb in pieces[k]:
            e2.process_ptr(L.x.data_ptr() + 16 * pos * 1024, n_in, y2.data_ptr() + 16 * pos * 256, n_out, nb)
            pos += nb
        assert pos == nblk
        e2.synchronize()
        torch.cuda.synchronize(dev)
        scale = float(ys[k].abs().max().item())
        d = (ys[k] - y2).abs().amax(dim=1)
        worst = max(worst, float(d.max().item()) / scale)
        assert float(d.max().item()) < 1e-8 * scale, (k, int(d.argmax().item()), float(d.max().item()))
    assert torch.equal(yp, y2p)
    print("config 4 shape: one call against uneven pieces, all 256 channels: max abs difference %.2e of full scale" % worst)
    e.close(); e2.close()


# ------------------------------------------------------------------------------------------------------------------ config 2, AGC on
@pytest.mark.parametrize("fading", [False, True], ids=["steady", "overs"])
def test_config2_agc_on_at_the_timed_shape(qh, oracle, bc, dev, fading):
    """256 ch x 2^22 per call with SetRXAAGCMode 3, two calls of the periodic buffer (one continuous stream): the AGC from its first
    sample (call 1) and in the steady state (call 2).  Eight channels against the oracle's xwcpagc over both calls; all 256 against
    the stream in uneven pieces (the short ones take the sequential kernel).  "overs": the leg's second input, keyed between full
    level and -40 dB every few seconds -- after every drop the tiles' warm-ups miss and segments are walked again in order
    (agc_bounds_fix_kernel): the path a steady input never takes."""
    L = bc.setup_config2_agc(torch, qh, dev, fading=fading)
    nch, nblk, n_in, n_out = L.nch, L.nblk, L.n_in, L.n_out
    assert (nch, nblk) == (256, 4096)
    e = L.eng
    ys = [torch.empty((nch, n_out), dtype=torch.complex128, device=dev) for _ in range(2)]
    torch.cuda.synchronize(dev)
    for k in range(2):
        e.process_ptr(L.x.data_ptr(), n_in, ys[k].data_ptr(), n_out, nblk)
    e.synchronize()
    torch.cuda.synchronize(dev)
    print("config 2 AGC shape (%s): agc_tiles_rerun %d, agc_segments_rerun %d" % ("overs" if fading else "steady", e.agc_repairs(), e.agc_segments_rerun()))
    if fading:
        assert e.agc_segments_rerun() > 0               # the fallback did run
    chans = spread(nch, 8)
    xs = {c: L.x[c].cpu().numpy() for c in chans}

    def check(c):
        o = oracle.WdspChannel(1024, 256, 192000, 48000, 48000)
        bc.c2agc_setters(_Ch(o), c, synth.shift_freq(c))
        return c, np.concatenate([o.xrxa(xs[c]) for _ in range(2)])
    for c, want in pmap(check, chans):
        got = np.concatenate([ys[k][c].cpu().numpy() for k in range(2)])
        assert np.abs(want).max() > 0.05
        head = slice(0, 64 * 256)
        e1, e2 = rel_rms(got[head], want[head]), rel_rms(got, want)
        print("config 2 AGC shape: channel %d rel RMS first 64 blocks %.2e, both calls %.2e, max abs %.2e" % (c, e1, e2, np.abs(got - want).max()))
        assert e1 < 1e-9 and e2 < 1e-9, (c, e1, e2)
        assert np.abs(got - want).max() < 1e-8 * np.abs(want).max(), c
    e2_ = qh.RxaEngine(nch, stream=torch.cuda.current_stream(dev).cuda_stream)
    for c in range(nch):
        bc.c2agc_setters(e2_, c, synth.shift_freq(c))
    y2 = torch.empty((nch, n_out), dtype=torch.complex128, device=dev)
    torch.cuda.synchronize(dev)
    for k, pieces in enumerate([(5, 1000, 3000, 91), (2500, 1596)]):
        pos = 0
        for nb in pieces:
            e2_.process_ptr(L.x.data_ptr() + 16 * pos * 1024, n_in, y2.data_ptr() + 16 * pos * 256, n_out, nb)
            pos += nb
        assert pos == nblk
        e2_.synchronize()
        torch.cuda.synchronize(dev)
        scale = float(ys[k].abs().max().item())
        d = (ys[k] - y2).abs().amax(dim=1)
        assert float(d.max().item()) < 1e-8 * scale, (k, int(d.argmax().item()), float(d.max().item()) / scale)
    e.close(); e2_.close()


# ------------------------------------------------------------------------------------------------------------------ config 5
def test_config5_fused_chain_at_the_timed_shape(qh, oracle, bc, dev):
    """One 2^26-sample fp32 stream per call through the 4 + 4 cascade, the 245-tap / 5 and the bandpass; two calls (state carries).
    The fp64 oracle (filter.c restatement, bit-pinned to the compiled reference) over the whole 2 x 2^26 samples; north_star's gate
    for fp32 is 1e-3 relative RMS, the measured figure is printed and held at 2e-5."""
    L = bc.setup_config5(torch, qh, dev, unfused=False)
    n = L.n
    assert n == 1 << 26
    outs = []
    for k in range(2):
        m = L.step_fused()
        torch.cuda.synchronize(dev)
        outs.append(L.yo[0, :m].cpu().numpy().astype(np.complex128))
    got = np.concatenate(outs)
    x = L.x[0].cpu().numpy().astype(np.complex128)

    def hb_chain(seg):
        return seg

    # the oracle's half-band cascade over 2 x 2^26 samples, stage by stage (state inside each OracleHB45)
    stages = [oracle.OracleHB45() for _ in range(8)]
    d5 = oracle.OracleFir(L.taps245)
    y5 = []
    for k in range(2):
        y = x
        for st in stages:
            y = st.cDecim2(y)
        y5.append(d5.cDecimate(y, 5))
    y5 = np.concatenate(y5)
    assert y5.size == got.size
    # the bandpass as the direct convolution with the bank's own taps (fp64), from sample 0
    want = np.convolve(y5, L.bp)[:y5.size]
    err = rel_rms(got, want)
    print("config 5 shape: 2 x 2^26 fp32 samples, %d outputs, rel RMS vs fp64 oracle %.3e" % (got.size, err))
    assert err < 2e-5
    assert np.abs(got - want).max() < 2e-4 * np.abs(want).max()
    # the same stream in uneven pieces (multiples of the cascade's 256-sample unit) through fresh engines
    L2 = bc.setup_config5(torch, qh, dev, n=n, unfused=False)
    L2.x.copy_(L.x)
    torch.cuda.synchronize(dev)
    parts = []
    for k in range(2):
        pos = 0
        for units in (1, 100000, 3, 162140):
            cnt = 256 * units
            m = L2.step_fused(L2.x.data_ptr() + 8 * pos, cnt)
            torch.cuda.synchronize(dev)
            parts.append(L2.yo[0, :m].cpu().numpy().astype(np.complex128))
            pos += cnt
        assert pos == n
    got2 = np.concatenate(parts)
    assert got2.size == got.size
    assert rel_rms(got2, got) < 2e-6        # fp32: tile boundaries fall elsewhere, the sums round differently


# ------------------------------------------------------------------------------------------------------------------ Quisk-native
@pytest.mark.parametrize("name", ["USB", "AM", "FM"])
def test_quisk_native_at_the_timed_shape(qh, oracle, bc, dev, name):
    """The whole of quisk_process_samples for 256 receivers x 2^20 per call (qh_qps_*, process_agc on at the bench's release gain,
    calls cut into bench_configs.QN_PIECES time pieces with the AGC on a second stream): three calls, one continuous stream of
    3 x 2^20 samples per receiver; process_agc's first call only initialises, as in the reference.

    process_agc is a DISCONTINUOUS function of its input (threshold tests decide when a ramp starts, steepens and ends, quisk.c:
    2220-2262): two streams that differ in the thirteenth digit part for good once a test falls the other way.  So the leg is held in
    two exact pieces and one statistical one:
      * the filters: the receiver bank alone (the same engine under the whole function), fed in the very pieces the whole function
        cuts, gives the stream process_agc sees in every call; eight receivers of it against the staged restatement of
        quisk_process_samples (1e-9), and the first call of the whole function (AGC initialising) bit for bit against it;
      * the AGC at this shape (256 streams, 4 x 65536 samples per call, beside the next piece's filters): 32 receivers of the whole
        function's output BIT FOR BIT against the restatement's process_agc run on the very stream the GPU's AGC saw;
      * end to end (restatement's filters + its AGC): within 1e-2, next to the restatement's OWN answer to a 1e-13 relative change
        of its input (printed: the same 1e-6 .. 1e-3 -- a flipped test changes one ramp's slope, with this much limiting every
        receiver has one in a call);
    and all 256 against uneven calls the same way."""
    L = bc.setup_quisk_native(torch, qh, dev, name)
    nch, n = L.nch, L.n
    assert (nch, n) == (256, 1 << 20)
    cuts = bc.qn_pieces(n)
    tabs = rxfilter.coefficient_tables()
    B = bc.setup_quisk_native(torch, qh, dev, name, whole=False)          # the same receivers: the bank alone
    B.x.copy_(L.x)
    torch.cuda.synchronize(dev)
    ys, pre = [], []
    for k in range(3):
        m = L.step()
        torch.cuda.synchronize(dev)
        ys.append(L.y[:, :m].clone())
        # the bank alone: the first call in one piece (as the whole function's first call), then in its pieces
        parts, pos = [], 0
        for cnt in ([n] if k == 0 else cuts):
            mb = B.bank.process_ptr(B.x.data_ptr() + 16 * pos, n, cnt, B.y.data_ptr(), B.m)
            torch.cuda.synchronize(dev)
            parts.append(B.y[:, :mb].clone())
            pos += cnt
        assert pos == n
        pre.append(torch.cat(parts, dim=1))
        assert pre[k].shape == ys[k].shape
    assert torch.equal(ys[0], pre[0])                 # the AGC's first call only initialises (quisk.c:2173-2190)
    assert not torch.equal(ys[1], pre[1])
    chans = spread(nch, 8)
    xs = {c: L.x[c].cpu().numpy() for c in chans}

    def check(c):
        # the reference takes at most SAMP_BUFFER_SIZE = 66000 samples per call (quisk.h:15) and its interpolators stop at 0.8 of that
        # (filter.c:158): the restatement gets the stream in 16384-sample blocks.  process_agc then runs once per GPU call on that
        # call's whole output, as quisk_process_samples runs it on its block (quisk.c:2686-2702)
        r = oracle.OracleQuiskRx(L.fs, tabs)
        r.set_mode(L.mode); r.set_bandwidth(L.bw); r.set_tune(bc.qn_tune(c)); r.set_filters(L.fI, L.fQ)
        agc = oracle.OracleQuiskAgc(48000)           # Agc1 runs at the playback rate (quisk.c:2174-2175), here = decim_srate
        out, outa = [], []
        for k in range(3):
            y = np.concatenate([r.process(xs[c][i:i + 16384]) for i in range(0, n, 16384)])
            assert r.decim_srate() == 48000
            out.append(y)
            outa.append(agc.process(y, False, bc.QN_AGC_GAIN))
        # how far the restatement's process_agc moves when its input moves by 1e-13 relative (the distance between two fp64 filters)
        agc2 = oracle.OracleQuiskAgc(48000)
        outb = [agc2.process(out[k] * (1.0 + 1e-13), False, bc.QN_AGC_GAIN) for k in range(3)]
        return c, out, outa, rel_rms(outb[2], outa[2])
    e2e, own = [], []
    for c, wants, wants_agc, sens in pmap(check, chans):
        own.append(sens)
        for k in range(3):
            want = wants[k]
            got = pre[k][c].cpu().numpy()
            assert got.size == want.size, (c, k)
            assert np.abs(want).max() > 2.0 ** 10
            skip = 2000 if (L.mode == 5 and k == 0) else 0          # FM: arg() of the filters' round-off floor while they fill
            err = rel_rms(got[skip:], want[skip:])
            assert err < 1e-9, (name, c, k, err)
            assert np.abs(got[skip:] - want[skip:]).max() < 1e-8 * np.abs(want).max(), (name, c, k)
        e2e.append(rel_rms(ys[2][c].cpu().numpy(), wants_agc[2]))
    print("Quisk-native %s shape: filters of 8 receivers within 1e-9 over 3 x 2^20 samples; end to end with process_agc (call 3): %s"
          % (name, " ".join("%.1e" % v for v in e2e)))
    print("Quisk-native %s shape: the restatement's process_agc against itself with the input scaled by 1 + 1e-13:           %s"
          % (name, " ".join("%.1e" % v for v in own)))
    assert max(e2e) < 1e-2, e2e
    # the AGC on the stream the GPU's AGC saw: bit for bit (|x| of a real stream is exact on both sides)
    chans32 = spread(nch, 32)
    p = [{c: pre[k][c].cpu().numpy() for c in chans32} for k in range(3)]
    g = [{c: ys[k][c].cpu().numpy() for c in chans32} for k in range(3)]

    def check_agc(c):
        agc = oracle.OracleQuiskAgc(48000)
        return c, [agc.process(p[k][c], False, bc.QN_AGC_GAIN) for k in range(3)]
    for c, wants in pmap(check_agc, chans32):
        assert np.abs(wants[2]).max() > 1e8                               # the limiter is at work
        for k in (1, 2):
            assert np.array_equal(g[k][c].view(np.float64), wants[k].view(np.float64)), (name, c, k)
    # all receivers, uneven calls through a fresh bank of the whole function
    L2 = bc.setup_quisk_native(torch, qh, dev, name)
    L2.x.copy_(L.x)
    torch.cuda.synchronize(dev)
    for k in range(3):
        pos, opos = 0, 0
        y2 = torch.zeros_like(ys[k])
        # process_agc's first call only initialises (quisk.c:2173-2190): call 0 stays ONE call, as in the run it is compared with
        for cnt in ((n,) if k == 0 else (1000, 1001, n // 3, n - n // 3 - 2001)):
            m = L2.bank.process_ptr(L2.x.data_ptr() + 16 * pos, n, cnt, L2.y.data_ptr(), L2.m)
            torch.cuda.synchronize(dev)
            y2[:, opos:opos + m] = L2.y[:, :m]
            pos += cnt
            opos += m
        assert pos == n and opos == ys[k].shape[1]
        scale = float(ys[k].abs().max().item())
        skip = 2000 if (L.mode == 5 and k == 0) else 0        # FM's first call: the fill-up phase noise differs by tile
        d = (ys[k][:, skip:] - y2[:, skip:]).abs().amax(dim=1) / scale
        if k == 0:
            assert float(d.max().item()) < 1e-9, (name, k, int(d.argmax().item()), float(d.max().item()))
        else:       # with the AGC running: a call boundary moves the filters' rounding, and a threshold test may fall the other way
            same = int((d < 1e-9).sum().item())
            print("Quisk-native %s shape: call %d against uneven calls with process_agc: %d of %d receivers within 1e-9, worst %.1e" % (name, k, same, nch, float(d.max().item())))
            assert float(d.max().item()) < 0.1
    # the pipelined form of the same calls (the leg's `pipelined_*` keys: a call returns with its AGC still running, the next call's filters
    # start beside it): the same pieces, the same kernels, the same bits
    del L2
    L3 = bc.setup_quisk_native(torch, qh, dev, name, pipelined=True)
    L3.x.copy_(L.x)
    torch.cuda.synchronize(dev)
    for k in range(3):
        m = L3.step()
        torch.cuda.synchronize(dev)
        assert m == ys[k].shape[1] and torch.equal(L3.y[:, :m], ys[k]), (name, k)
