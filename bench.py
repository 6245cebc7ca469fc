#!/usr/bin/env python3
"""bench.py -- throughput of the WDSP-RXA SSB receive chain on MI355X.

Workload (BASELINE.json configs[1]): 256 channels x 192 kHz IQ -> 48 kHz, fp64,
shift -> 561-tap resampler /4 -> NBP fircore (nc 2048, 300..3000 Hz) -> fixed-gain AGC -> panel,
one pass ("step") = 2^22 input samples per channel, inputs resident in HBM.
With --gpus N every rank runs its own 256 channels (independent receivers shard by channel, no
collective on the data path): weak scaling, value = all ranks' input samples / max-over-ranks time.
--total-channels C instead splits C channels over the ranks (strong scaling; north_star's shape is 256 over 8).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NCH = 256
LOG2_SAMPLES = 22
IN_RATE, DSP_RATE = 192000, 48000
DSP_SIZE = 256
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
COPY_CEILING_GBPS = 6290.0      # MI355X_MICROARCH.md: what a float4 copy measures (79 % of the 8 TB/s)
VALU_PEAK_LANE_OPS = 39.3e12    # 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz: one lane operation (an FMA = 2 flop) per lane and clock = 78.6 TFLOP/s fp64 vector
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "c2_traffic.json")


def kernel_source_sha16():
    """Fingerprint of the sources the two measured kernels are compiled from -- qh_engine.hip and every file of quisk_amd/csrc it
    includes, directly or not: tools/pmc_pass.py stamps the counter-derived HBM traffic with it, and a stamp that does not match the
    sources this run was built from is not reported (counters cannot be read from inside the run).  Other translation units of the
    library (panadapter, receiver bank, ...) do not enter."""
    import hashlib
    import re
    d = os.path.join(ROOT, "quisk_amd", "csrc")
    seen, todo = [], ["qh_engine.hip"]
    while todo:
        f = todo.pop()
        if f in seen or not os.path.exists(os.path.join(d, f)):
            continue
        seen.append(f)
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(os.path.join(d, f)).read(), re.M):
            if "/" not in inc:
                todo.append(inc)
    h = hashlib.sha256()
    for f in sorted(seen):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def _trace(what):
    """--dry-run plumbing tests only: with QH_BENCH_TRACE=<file> every rank appends `rank what unix-time` lines (who built, when each
    rank left its warm-up, entered and left the timed region), which tests/test_bench_plumbing.py reads back."""
    f = os.environ.get("QH_BENCH_TRACE")
    if f:
        with open(f, "a") as fh:
            fh.write("%s %s %.6f\n" % (os.environ.get("RANK", "0"), what, time.time()))


class DryEngine:
    """--dry-run only (tests/test_bench_plumbing.py): stands where RxaEngine stands so that the argument / rank / channel
    split / barrier / max-over-ranks / JSON plumbing of this file can run in CPU processes.  It does no DSP; a line made
    with it says "dry_run": true and is not a measurement."""

    def __init__(self, nch, **kw):
        self.nch, self.calls, self.rank = nch, [], int(os.environ.get("RANK", "0"))
        self.setters = 0

    def __getattr__(self, name):
        if name.startswith(("Set", "RXA")):
            def setter(*a):
                self.setters += 1
            return setter
        raise AttributeError(name)

    def enable_meters(self, on):
        self.meters = on

    def process_ptr(self, *a):
        time.sleep(0.001 * (self.rank + 1))

    def enable_timing(self, on):
        pass

    def timing_ms(self):
        return [1.0, 0.5, 0.1]

    def GetRXAMeter(self, ch, mt):
        return -20.0


def cpu_baseline(log2_single=25, log2_each=24):
    """The CPU oracle (oracle/wdsp_oracle.c, a restatement of the reference's WDSP path; own radix-2 FFT, not FFTW) timed on the
    host cores of this box, outside the timed region, the two ways BASELINE.md section 3 plans:
      (i)  one channel on one core (a worker process pinned to the first core this process may use);
      (ii) one channel per core: one worker PROCESS per core, each pinned, each with its own input and output buffers made on
           its core (first touch), all started at one agreed wall-clock time; rate = all samples / (last end - first start).
    The workers are fresh processes (tools/cpu_baseline_worker.py: numpy + ctypes, no torch, no GPU)."""
    import subprocess
    from oracle import pyoracle as po
    po.build(ref=False)
    avail = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    visible = len(avail)
    # The container's CPU-time quota (cgroup cpu.max / cfs_quota_us): the GPU boxes of this pool show 256 hardware threads but
    # grant 16 CPUs' worth of time, so 256 busy workers each get a sixteenth of a core (that was round 2's 0.41 Msamp/s "per
    # core").  The baseline uses as many cores as the quota grants, spread over the visible ones, and says so.
    quota = None
    for f_quota, f_period in (("/sys/fs/cgroup/cpu.max", None), ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")):
        try:
            if f_period is None:
                q, per = open(f_quota).read().split()[:2]
            else:
                q, per = open(f_quota).read().strip(), open(f_period).read().strip()
            if q not in ("max", "-1"):
                quota = float(q) / float(per)
            break
        except (OSError, ValueError):
            continue
    if quota is not None and quota < visible:
        ncores = max(1, int(quota))
        stride = visible // ncores
        avail = [avail[i * stride] for i in range(ncores)]
    worker = os.path.join(ROOT, "tools", "cpu_baseline_worker.py")

    def run(cores, log2n, lead):
        t_start = time.time() + lead
        procs = []
        outs = []
        try:
            for i, c in enumerate(cores):
                procs.append(subprocess.Popen([sys.executable, worker, str(c), str(i), str(log2n), repr(t_start)], stdout=subprocess.PIPE, text=True))
            deadline = time.time() + lead + 240.0           # the whole set, not 600 s per worker
            for pr in procs:
                o, _ = pr.communicate(timeout=max(1.0, deadline - time.time()))
                if pr.returncode != 0:
                    raise RuntimeError("cpu baseline worker failed")
                outs.append(json.loads(o.strip().splitlines()[-1]))
        finally:                                            # a timeout or a failed worker leaves no pinned process behind
            for pr in procs:
                if pr.poll() is None:
                    pr.kill()
                    try:
                        pr.communicate(timeout=5)
                    except Exception:
                        pass
        span = max(o["t1"] for o in outs) - min(o["t0"] for o in outs)
        late = max(o["t0"] for o in outs) - t_start            # > 0.5 s: a worker was not ready at the start time
        return sum(o["samples"] for o in outs) / span / 1e6, span, late, outs

    single, span1, _, _ = run(avail[:1], log2_single, 4.0)
    # workers need a few seconds to import numpy and make 2^log2_each samples; give 256 of them room
    lead = 6.0 + 0.05 * len(avail)
    allc, span, late, outs = run(avail, log2_each, lead)
    per = sorted((1 << log2_each) / o["seconds"] / 1e6 for o in outs)
    return {"value": allc, "unit": "Mcomplex-samples/s", "cores": len(avail), "nproc": os.cpu_count(), "cpu_quota": quota, "kind": "port",
            "single_thread_value": single, "per_core_value": allc / len(avail),
            "per_core_min_median_max": [per[0], per[len(per) // 2], per[-1]], "start_skew_s": late,
            "sample": "(i) 1 channel x 2^%d input samples on one pinned core, %.1f s; (ii) %d channels, one pinned process per core (as many cores as the container's CPU quota "
                      "grants), each 2^%d samples in buffers of its own, %.1f s; oracle/wdsp_oracle.c -O3 -march=native, own radix-2 FFT (not FFTW)"
                      % (log2_single, span1, len(avail), log2_each, span)}


def other_configs(timeout_s=900.0):
    """BASELINE.json configs 3, 4, 5 (one GPU's share each), the headline chain with the AGC on and the Quisk-native chain (path A):
    tools/bench_configs.py's legs, run in a FRESH CHILD PROCESS (`tools/bench_configs.py driver`) after the headline workload's
    timed region and after this process has given its device memory back.  A fault, a hang or a time-out in any extra leg costs
    that leg (the child reports leg by leg, so the finished ones are kept), never the headline measurement.  Reported as extra
    keys; `value` is configuration 2's alone."""
    import subprocess
    import tempfile
    out = {}
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "legs.jsonl")
        cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_configs.py"), "driver", path]
        try:
            r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=timeout_s)
            note = None if r.returncode == 0 else "child exit code %d: %s" % (r.returncode, (r.stderr or "")[-300:])
        except subprocess.TimeoutExpired:
            note = "child timed out after %.0f s" % timeout_s
        try:
            for line in open(path):
                d = json.loads(line)
                out[d["key"]] = d["leg"]
        except OSError:
            pass
        if note:
            out["child_failed"] = note
    return out


def live_traffic(meters, timeout_s=240.0):
    """HBM bytes per input sample of the two config-2 kernels measured IN THIS RUN: two `rocprofv3 --pmc` passes (FETCH_SIZE, then
    WRITE_SIZE: separate passes as MI355X_MICROARCH.md prescribes) over a short child run of this very file -- counters cannot be read
    inside a process, and the child starts from scratch (a profiler's preloaded library and an exec do not mix).  rocprofv3 reports KiB;
    FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 bytes, same guide).  Returns {"front": B, "band": B, ...} or raises."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not rocprof:
        raise RuntimeError("no rocprofv3 on this box")
    log2 = 20
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--log2-samples", str(log2), "--meters", meters,
             "--no-cpu-baseline", "--no-other-configs", "--no-le24", "--no-host-fed", "--no-live-traffic"]
    samples = 256.0 * (1 << log2)
    got = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(td, counter)
            r = subprocess.run([rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--"] + child,
                               capture_output=True, text=True, timeout=timeout_s, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp")
            acc = {"front": [], "band": []}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    name = row["Kernel_Name"]
                    if row["Counter_Name"] != counter or "osfir_kernel<double, 4096," not in name:
                        continue
                    if "osfir_kernel<double, 4096, 4, false, false" in name:
                        acc["front"].append(float(row["Counter_Value"]))
                    elif "osfir_kernel<double, 4096, 1" in name and (("true, false" in name.split("4096, 1,")[1][:30]) == (meters == "on")):
                        acc["band"].append(float(row["Counter_Value"]))
            if not acc["front"] or not acc["band"]:
                raise RuntimeError("rocprofv3 --pmc %s gave no rows for the two kernels (exit %d): %s" % (counter, r.returncode, (r.stderr or r.stdout)[-300:]))
            for k in acc:
                got.setdefault(k, {})[counter] = sum(acc[k]) / len(acc[k])
    res = {}
    for k in ("front", "band"):
        res[k] = (2048.0 * got[k]["FETCH_SIZE"] + 1024.0 * got[k]["WRITE_SIZE"]) / samples
        res[k + "_fetch_x2_bytes_per_sample"] = 2048.0 * got[k]["FETCH_SIZE"] / samples
        res[k + "_write_bytes_per_sample"] = 1024.0 * got[k]["WRITE_SIZE"] / samples
    res["child_log2_samples"] = log2
    return res


def host_fed(torch, eng, x, y, nch, n_in, n_out, nblk, dev, stream, log2_chunk=18, passes=2):
    """SURVEY.md 8(d)'s "H2D reported separately": the reference's samples arrive on the HOST (quisk.c:3284-3423 UDP, sound.c:990), so
    this is what a deployment whose samples do not already sit in HBM gets.  The step's input streams from PINNED host memory in
    time chunks of 2^log2_chunk samples per channel, two device buffers, the copy of chunk k + 1 on a copy stream beside the kernels
    of chunk k, the chunk's output copied back to pinned host memory behind them (events order the three streams): once as fp64
    complex (16 B per sample over the link, 4 B back) and once in the 24-bit wire format (6 B per sample, decoded in the front
    kernel's load).  Never `value`: the timed region of the headline starts with the input resident in HBM."""
    from quisk_amd import IqFormat
    ck = 1 << log2_chunk
    if n_in % ck:
        return {"skipped": "chunk does not divide the step"}
    nck = n_in // ck
    cb, co = nblk // nck, n_out // nck
    copy_s, back_s = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    out = {"chunk_samples_per_channel": ck, "chunks_per_step": nck, "link": "PCIe Gen5 x16, 63 GB/s spec (MI355X_MICROARCH.md)"}
    fmt24 = IqFormat.le24(2.0 ** -31)
    for kind in ("f64", "le24"):
        bps = 16 if kind == "f64" else 6
        if kind == "f64":
            host = torch.view_as_real(x[:, :ck]).contiguous().cpu().pin_memory()           # one chunk's worth, fed again and again
            dbuf = [torch.empty((nch, ck), dtype=torch.complex128, device=dev) for _ in range(2)]
        else:
            codes = torch.view_as_real(x[:, :ck]).mul(2.0 ** 23).round_().to(torch.int32)
            host = codes.view(torch.uint8).reshape(nch, ck, 2, 4)[..., :3].contiguous().cpu().pin_memory()
            del codes
            dbuf = [torch.empty((nch, ck, 2, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
        hout = torch.empty((nch, co, 2), dtype=torch.float64).pin_memory()
        dout = [torch.empty((nch, co), dtype=torch.complex128, device=dev) for _ in range(2)]
        ev_in = [torch.cuda.Event() for _ in range(2)]          # chunk's input has landed
        ev_done = [torch.cuda.Event() for _ in range(2)]        # chunk's kernels are done (its input buffer and output buffer are free / ready)
        ev_back = [torch.cuda.Event() for _ in range(2)]        # chunk's output has left

        def one_pass():
            for k in range(nck):
                b = k & 1
                with torch.cuda.stream(copy_s):
                    if k >= 2:
                        copy_s.wait_event(ev_done[b])           # the kernels of chunk k - 2 have read this buffer
                    (torch.view_as_real(dbuf[b]) if kind == "f64" else dbuf[b]).copy_(host, non_blocking=True)
                    ev_in[b].record(copy_s)
                stream.wait_event(ev_in[b])
                if k >= 2:
                    stream.wait_event(ev_back[b])               # the output of chunk k - 2 has left this buffer
                if kind == "f64":
                    eng.process_ptr(dbuf[b].data_ptr(), ck, dout[b].data_ptr(), co, cb)
                else:
                    eng.process_packed_ptr(dbuf[b].data_ptr(), dbuf[b].numel(), fmt24, 6 * ck, dout[b].data_ptr(), co, cb)
                ev_done[b].record(stream)
                with torch.cuda.stream(back_s):
                    back_s.wait_event(ev_done[b])
                    hout.copy_(torch.view_as_real(dout[b]), non_blocking=True)
                    ev_back[b].record(back_s)
            torch.cuda.synchronize(dev)
        one_pass()                                              # warm-up (buffers of the engine at this call shape)
        t0 = time.perf_counter()
        for _ in range(passes):
            one_pass()
        dt = (time.perf_counter() - t0) / passes
        samples = float(nch) * n_in
        out[kind] = {"ms_per_step": dt * 1e3, "Msamp_per_s": samples / dt / 1e6, "link_GBps_in": bps * samples / dt / 1e9,
                     "link_GBps_out": 4.0 * samples / dt / 1e9, "bytes_per_sample_in": bps, "bytes_per_sample_out": 4.0}
        del host, dbuf, hout, dout
        torch.cuda.empty_cache()
    out["note"] = ("input from pinned host memory in chunks, H2D beside the kernels, output D2H behind them; the same engine and step as `value`, "
                   "whose input is resident in HBM")
    return out


class DryLeg:
    """--dry-run stand-in for a configuration 4 / 5 leg (tests/test_bench_plumbing.py): the plumbing around it is what is tested."""

    def __init__(self, rank):
        self.rank, self.calls = rank, 0

    def step(self):
        self.calls += 1
        time.sleep(0.001 * (self.rank + 1))


def sharded_config(args, torch, dist, shard, rank, world, local_rank, dev, dry, sync):
    """BASELINE.json configurations 4 and 5 on N GPUs: channels are independent (wdsp/RXA.c:29, channel.c:29), every rank owns a
    contiguous range and runs the leg tools/bench_configs.py builds for it (the same setup_* functions the parity tests of the call
    shapes use, tests/test_gpu_bench_shapes.py); barrier-bracketed timing, max over ranks, one line from rank 0 with every rank's own
    kernel time and roofline.  Config 4: 256 channels per GPU, mode by the job-wide channel index mod 3 (2048 on eight GPUs);
    --total-channels splits a fixed job instead.  Config 5: one 61.44 Msps fp32 channel per GPU (8 on eight)."""
    cfg = args.config
    if cfg == 5 and (args.channels != 1 or args.total_channels):
        raise SystemExit("bench.py --config 5 runs one 61.44 Msps channel per GPU")
    if args.total_channels > 0:
        mine = shard.split_channels(args.total_channels, world)[rank]
        first, nch, scaling, total_channels = mine.start, len(mine), "strong", args.total_channels
        if nch == 0:
            raise SystemExit("bench.py: rank %d has no channel (--total-channels %d over %d ranks)" % (rank, args.total_channels, world))
    else:
        nch = args.channels
        first = shard.channel_range(rank, world, nch)[0]
        scaling, total_channels = "weak", nch * world
    if dry:
        leg, samples_rank = DryLeg(rank), float(nch) * (1 << args.log2_samples)
        step = leg.step
    else:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_configs as bc
        import quisk_amd as qh
        if cfg == 4:
            nblk = (1 << args.log2_samples) // 1024
            leg = bc.setup_config4(torch, qh, dev, nch=nch, nblk=nblk, first=first)
            samples_rank = float(nch) * leg.n_in
            step = leg.step
        else:
            n = 1 << (args.log2_samples if args.log2_samples != LOG2_SAMPLES else 26)       # 2^26 samples of the one stream per step
            leg = bc.setup_config5(torch, qh, dev, n=n, unfused=False)
            samples_rank = float(n)
            step = leg.step_fused
    sync()

    def timed_run():
        for _ in range(args.warmup):
            step()
        sync()
        _trace("warm_done")
        if world > 1:
            dist.barrier()
        sync()
        _trace("timed_start")
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        _trace("steps_done")
        if world > 1:
            dist.barrier()
        sync()
        _trace("timed_end")
        return shard.max_over_ranks(time.perf_counter() - t0, dev if not dry else None)

    dt = timed_run()
    # the dominant kernel's own time on this rank, HIP events on the stream it is launched on
    if dry:
        kname, kms, algo = "dry", 1.0, 20.0
    elif cfg == 4:
        leg.eng.enable_timing(True)
        kt = [0.0, 0.0, 0.0]
        for _ in range(3):
            leg.step()
            kt = [a + b / 3 for a, b in zip(kt, leg.eng.timing_ms())]
        leg.eng.enable_timing(False)
        kname, kms, algo = "osfir_kernel<f64,4096,D=4,OUTMIX> front (shift + resample /4, shared by USB / AM / FM)", kt[0], 20.0
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        x, bufs, n = leg.x, leg.bufs, leg.n
        sync()
        e0.record(leg.stream)
        for _ in range(5):
            leg.casc.process_ptr(x.data_ptr(), n, n, bufs[-1].data_ptr(), bufs[-1].shape[1])
        e1.record(leg.stream)
        sync()
        kname, kms, algo = "hb45_cascade_kernel<float,4> x 2 (4 + 4 half-band stages, one HBM pass)", e0.elapsed_time(e1) / 5, 8.0
    achieved = algo * samples_rank / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
    mine_rec = {"rank": rank, "device": local_rank, "channels": [first, first + nch], "dominant_kernel_ms": kms,
                "roofline": {"kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS}}
    per_rank = [mine_rec]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine_rec)
    if rank == 0:
        per_step = (float(total_channels) * (1 << args.log2_samples)) if (dry or cfg == 4) else samples_rank * world
        total = per_step * args.steps
        bytes_per_sample = 20.0 if cfg == 4 else 8.0
        workload = ("config 4: %d channels/GPU x 192 kHz, mode by job-wide channel mod 3 = USB / AM / FM (amd / fmd / nbp), fp64, 2^%d input samples per channel per step"
                    % (nch, args.log2_samples)) if cfg == 4 else \
                   ("config 5: 1 channel/GPU x 61.44 Msps fp32: 8 x HB45 (fused cascade) + 245-tap /5 + overlap-save bandpass nc 2048, %d samples per step" % int(samples_rank))
        line = {"metric": "Mcomplex-samples/s through RXA chain", "value": total / dt / 1e6, "unit": "Mcomplex-samples/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
                "vs_baseline": None, "dtype": "f64" if cfg == 4 else "f32", "data": "synthetic",
                "config": {"workload": workload, "baseline_config": cfg, "channels_per_gpu": nch, "total_channels": total_channels,
                           "parallelism": "channel-sharded x%d (%s), no collective" % (world, scaling)},
                "chain_algorithmic_GBps": bytes_per_sample * total / dt / 1e9,
                "roofline": dict(per_rank[0]["roofline"], bound="latency", traffic=None,
                                 algorithmic_bytes_per_launch=algo * samples_rank,
                                 traffic_note="counter passes of this leg: profiles/ (tools/pmc_pass.py)"),
                "per_rank": per_rank}
        if dry:
            line["dry_run"] = True
            line["dry"] = {"rank0_channels": [first, first + nch], "steps_run": leg.calls}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def self_spawn(args):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): N fresh child processes, one per GPU, with
    the rank variables torch.distributed.run would set, started BEFORE this process imports torch or touches a GPU.  Rank 0's
    line is relayed; the exit code is the worst child's."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import tempfile
    procs = []
    with tempfile.TemporaryFile("w+") as out0:
        try:
            for rank in range(args.gpus):
                env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                           MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), QH_BENCH_SPAWNED="1")
                env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
                procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                              stdout=out0 if rank == 0 else subprocess.DEVNULL))
            failed_at = None
            while any(pr.poll() is None for pr in procs):
                time.sleep(0.2)
                if failed_at is None and any(pr.poll() not in (None, 0) for pr in procs):
                    failed_at = time.time()                 # a rank died: the others may sit in a barrier for good
                if failed_at is not None and time.time() - failed_at > 20.0:
                    break
        finally:
            for pr in procs:
                if pr.poll() is None:
                    pr.kill()
        rc = [pr.wait() for pr in procs]
        out0.seek(0)
        out0 = out0.read()
    sys.stdout.write(out0)
    sys.stdout.flush()
    return max(abs(r) for r in rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, choices=[2, 4, 5], default=2,
                    help="BASELINE.json configuration: 2 (the headline: 256 ch SSB chain, what the driver runs), 4 (mixed USB / AM / FM, 256 channels per GPU) "
                         "or 5 (61.44 Msps fp32 half-band cascade + /5 + bandpass, one channel per GPU); 4 and 5 shard over --gpus N like 2 does")
    ap.add_argument("--channels", type=int, default=None, help="channels per GPU (weak scaling); default 256 (configs 2, 4) or 1 (config 5)")
    ap.add_argument("--total-channels", type=int, default=0,
                    help="strong scaling: this many channels in all, split over the ranks in contiguous ranges (SURVEY.md 8(e))")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the barrier / max-over-ranks (nccl = RCCL)")
    ap.add_argument("--dry-run", action="store_true", help="plumbing test only: no GPU, no DSP (DryEngine); the line is not a measurement")
    ap.add_argument("--log2-samples", type=int, default=LOG2_SAMPLES, help="input samples per channel per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-le24", action="store_true", help="skip the extra run from 24-bit wire samples (the test of the front kernel's HBM bound)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the extra legs (BASELINE configs 3, 4, 5 and the Quisk-native chain)")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from the stamped profiles/c2_traffic.json only: no rocprofv3 --pmc child passes in this run")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the host-fed leg (pinned host buffers, copies beside the kernels: what a receiver whose samples arrive on the host gets)")
    ap.add_argument("--meters", choices=["on", "off"], default="on",
                    help="on (default): xrxa's three meters run as in the reference (adc, S, agc: wdsp/RXA.c:566,569,589), "
                         "fused into the nbp0 launch; the line also carries the rate of a second, untimed-for-`value` run with them off")
    ap.add_argument("--chunks", type=int, default=1,
                    help="split a step into this many consecutive time chunks (engine calls); the intermediate "
                         "buffer of a chunk then stays in the 256 MiB Infinity Cache")
    ap.add_argument("--band-tile", type=int, default=0, help="tile of the fircore (nbp0) stage: 0 = the engine's default, 4096, 6144 or 8192")
    ap.add_argument("--ingest", choices=["f64", "le24"], default="f64",
                    help="f64: complex double input resident in HBM (the BASELINE workload); le24: the same signal as "
                         "24-bit little-endian IQ bytes (quisk_read_rx_udp wire format), decoded in the front kernel's load")
    args = ap.parse_args()
    if args.channels is None:
        args.channels = 1 if args.config == 5 else NCH

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:          # no launcher: be one (before torch, before any GPU call)
        raise SystemExit(self_spawn(args))

    import torch
    import torch.distributed as dist
    from quisk_amd import synth, shard

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    dry = args.dry_run
    if dry:
        dev = torch.device("cpu")
        sync = lambda: None
        if rank == 0:
            _trace("build")             # where the real run builds the library: rank 0 only, ahead of the first barrier
    else:
        from quisk_amd import build as qbuild
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device: the RXA chain has no CPU path")
        if rank == 0:
            qbuild.build()
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        sync = lambda: torch.cuda.synchronize(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if (dry and args.backend == "nccl") else args.backend
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        dist.barrier()

    if args.config != 2:
        return sharded_config(args, torch, dist, shard, rank, world, local_rank, dev, dry, sync)

    n_in = 1 << args.log2_samples
    nblk = n_in // (DSP_SIZE * (IN_RATE // DSP_RATE))
    n_out = nblk * DSP_SIZE
    if args.total_channels > 0:
        # strong scaling: the job's channels in contiguous ranges (256 over 8 GPUs = 32 each)
        mine = shard.split_channels(args.total_channels, world)[rank]
        first, nch, scaling = mine.start, len(mine), "strong"
        total_channels = args.total_channels
        if nch == 0:
            raise SystemExit("bench.py: rank %d has no channel (--total-channels %d over %d ranks)" % (rank, args.total_channels, world))
    else:
        nch = args.channels
        first = shard.channel_range(rank, world, nch)[0]       # rank r owns channels [r*nch, (r+1)*nch)
        scaling, total_channels = "weak", nch * world

    if dry:
        eng = DryEngine(nch)
        stream = None
    else:
        from quisk_amd import RxaEngine
        stream = torch.cuda.current_stream(dev)
        eng = RxaEngine(nch, dsp_size=DSP_SIZE, in_rate=IN_RATE, dsp_rate=DSP_RATE, out_rate=DSP_RATE,
                        device=local_rank, stream=stream.cuda_stream)
    eng.SetRXAShiftRun(-1, 1)
    for c in range(nch):
        eng.SetRXAShiftFreq(c, synth.shift_freq(first + c))
    eng.RXANBPSetRun(-1, 1)
    eng.SetRXAMode(-1, 1)
    eng.RXASetPassband(-1, 300.0, 3000.0)
    eng.SetRXAAGCMode(-1, 0)
    eng.SetRXAAGCFixed(-1, 0.0)
    eng.enable_meters(args.meters == "on")
    if args.band_tile and not dry:
        eng.set_band_tile(args.band_tile)

    if dry:
        x = torch.zeros((nch, 8), dtype=torch.complex128)
        y = torch.zeros((nch, 8), dtype=torch.complex128)
    else:
        x = synth.make_input_torch(nch, n_in, dev, fs=float(IN_RATE), first_channel=first)
        y = torch.empty((nch, n_out), dtype=torch.complex128, device=dev)
    sync()

    nchunk = max(1, args.chunks)
    if nblk % nchunk:
        raise SystemExit("--chunks must divide %d" % nblk)
    cb = nblk // nchunk

    if args.ingest == "le24":
        from quisk_amd import IqFormat
        if nchunk != 1:
            raise SystemExit("--ingest le24 runs unchunked")
        codes = torch.view_as_real(x).mul(2.0 ** 23).round_().to(torch.int32)       # [nch, n, 2] 24-bit ADC codes
        packed = codes.view(torch.uint8).reshape(nch, n_in, 2, 4)[..., :3].contiguous()     # 6 bytes per sample
        del codes
        fmt = IqFormat.le24(2.0 ** -31)                     # left-justified int32 (code * 2^8) back to +-1.0 full scale
        sync()

        def step():
            eng.process_packed_ptr(packed.data_ptr(), packed.numel(), fmt, 6 * n_in, y.data_ptr(), n_out, nblk)
    else:
        def step():
            for k in range(nchunk):
                eng.process_ptr(x.data_ptr() + 16 * k * cb * (n_in // nblk), n_in, y.data_ptr() + 16 * k * cb * (n_out // nblk), n_out, cb)

    def timed_run():
        for _ in range(args.warmup):
            step()
        sync()
        _trace("warm_done")
        if world > 1:
            dist.barrier()
        sync()
        _trace("timed_start")
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        _trace("steps_done")
        if world > 1:
            dist.barrier()
        sync()
        _trace("timed_end")
        return shard.max_over_ranks(time.perf_counter() - t0, dev if not dry else None)

    dt = timed_run()                    # the measurement `value` reports
    meter_db = [eng.GetRXAMeter(0, mt) for mt in (1, 3, 6)] if args.meters == "on" else None
    dt_off = None
    if args.meters == "on":             # the same steps once more with the meters off, reported beside it
        eng.enable_meters(False)
        dt_off = timed_run()
        eng.enable_meters(True)

    # per-kernel durations with HIP events on the engine's stream (separate short run, not in the timed region)
    eng.enable_timing(True)
    kt = [0.0, 0.0, 0.0]
    reps = 5
    for _ in range(reps):
        step()
        t = eng.timing_ms()
        kt = [a + b for a, b in zip(kt, t)]
    kt = [v / reps for v in kt]
    eng.enable_timing(False)

    # sanity: the in-band tone must come out with the panel gain of 4.0
    tail = y[0, -4096:]
    gain = float(tail.abs().mean().item()) / 0.1

    # The same steps from 24-bit wire samples (6 B instead of 16 B per input sample through the SAME front kernel, decoded in its
    # load): the direct test of whether that kernel waits for HBM.  Reported as an extra key; `value` is the fp64-input run's.
    le24 = None
    if args.ingest == "f64" and world == 1 and not dry and not args.no_le24 and nchunk == 1:
        try:
            from quisk_amd import IqFormat
            codes = torch.view_as_real(x).mul(2.0 ** 23).round_().to(torch.int32)
            packed = codes.view(torch.uint8).reshape(nch, n_in, 2, 4)[..., :3].contiguous()
            del codes
            fmt = IqFormat.le24(2.0 ** -31)
            sync()

            def step24():
                eng.process_packed_ptr(packed.data_ptr(), packed.numel(), fmt, 6 * n_in, y.data_ptr(), n_out, nblk)
            for _ in range(2):
                step24()
            sync()
            t0 = time.perf_counter()
            for _ in range(5):
                step24()
            sync()
            dt24 = (time.perf_counter() - t0) / 5
            eng.enable_timing(True)
            k24 = [0.0, 0.0, 0.0]
            for _ in range(3):
                step24()
                k24 = [a + b / 3 for a, b in zip(k24, eng.timing_ms())]
            eng.enable_timing(False)
            le24 = {"ms_per_step": dt24 * 1e3, "Msamp_per_s": float(nch) * n_in / dt24 / 1e6, "front_ms": k24[0], "band_ms": k24[1],
                    "front_ms_f64_input": kt[0], "front_algorithmic_bytes_per_sample": 10.0,
                    "front_algorithmic_GBps": 10.0 * float(nch) * n_in / (k24[0] * 1e-3) / 1e9,
                    "note": "24-bit little-endian IQ (quisk_read_rx_udp's wire format) decoded in the front kernel's load: 6 B in + 16/4 B out per "
                            "input sample instead of 16 + 4.  A front kernel bound by HBM would run ~2x faster here."}
            del packed
            torch.cuda.empty_cache()
        except Exception as exc:                             # reported, never required
            le24 = {"failed": repr(exc)}

    hostfed = None
    if world == 1 and not dry and not args.no_host_fed and args.ingest == "f64" and nchunk == 1:
        try:
            hostfed = host_fed(torch, eng, x, y, nch, n_in, n_out, nblk, dev, stream)
        except Exception as exc:                             # reported, never required
            hostfed = {"failed": repr(exc)}

    # every rank's own figures (over the group that already exists), so that a slow GPU is attributable
    algo_b = [20.0 if args.ingest == "f64" else 10.0, 8.0]
    mine_rec = {"rank": rank, "device": local_rank, "channels": [first, first + nch], "kernel_ms": {"front_shift_resample": kt[0], "band_nbp": kt[1], "state_bookkeeping": kt[2]},
                "check_inband_gain": gain,
                "roofline": {"kernel": "front" if kt[0] >= kt[1] else "band", "unit": "GB/s", "peak": HBM_PEAK_GBPS,
                             "achieved": (algo_b[0] if kt[0] >= kt[1] else algo_b[1]) * float(nch) * n_in / (max(kt[0], kt[1]) * 1e-3) / 1e9 if max(kt[0], kt[1]) > 0 else None}}
    if mine_rec["roofline"]["achieved"] is not None:
        mine_rec["roofline"]["frac"] = mine_rec["roofline"]["achieved"] / HBM_PEAK_GBPS
    per_rank = [mine_rec]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine_rec)

    if rank == 0:
        samples_per_step = float(nch) * n_in                     # this rank's launch (kernel-level figures below)
        job_samples_per_step = float(total_channels) * n_in      # all ranks
        total = job_samples_per_step * args.steps
        value = total / dt / 1e6
        # dominant kernel and its algorithmic bytes per launch:
        #   front  (shift + resample /4): reads 16 B, writes 16/4 B per input sample        = 20 B / input sample
        #   band   (NBP overlap-save)   : reads 16 B, writes 16 B per DSP-rate sample (x1/4) =  8 B / input sample
        names = ["osfir_kernel<f64,4096,D=4,OUTMIX> (561-tap resample /4 + shift)", "osfir_kernel<f64,4096,D=1> (nbp fircore)"]
        algo = [20.0 if args.ingest == "f64" else 10.0, 8.0]      # 24-bit ingest: 6 B in + 16/4 B out
        k = 0 if kt[0] >= kt[1] else 1
        achieved = algo[k] * samples_per_step / (kt[k] * 1e-3) / 1e9
        # HBM traffic per launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE, profiles/c2_traffic.json, stamped with the kernel sources' hash),
        # scaled by the number of samples: counters cannot be read inside this process
        traffic, traffic_note = None, None
        valu = None                 # executed VALU lane operations per second of each kernel, from the same stamped counter passes
        try:
            tj = json.load(open(TRAFFIC_JSON))
            if tj.get("source_sha16") != kernel_source_sha16():
                traffic_note = "profiles/c2_traffic.json was measured on other kernel sources (%s): not reported" % tj.get("source_sha16")
            elif args.ingest == "f64" and tj.get("meters") == args.meters:
                traffic = tj["kernels"]["front" if k == 0 else "band"]["bytes_per_input_sample"] * samples_per_step
                valu = {}
                for i, key in enumerate(("front", "band")):
                    w = tj["kernels"][key].get("valu_wave_insts_per_input_sample")
                    if w is not None:
                        ops = 64.0 * w * samples_per_step / (kt[i] * 1e-3)
                        valu[key] = {"lane_ops_per_s": ops, "frac": ops / VALU_PEAK_LANE_OPS,
                                     "hbm_frac": algo[i] * samples_per_step / (kt[i] * 1e-3) / 1e9 / HBM_PEAK_GBPS}
            else:
                traffic_note = "profiles/c2_traffic.json holds another variant (ingest / meters)"
        except Exception as exc:
            traffic_note = "no counter-derived traffic: %r" % (exc,)
        dom = "front" if k == 0 else "band"
        valu_dom = valu.get(dom) if valu else None
        # What bounds the dominant kernel, from evidence rather than from the larger of two fractions: "hbm" needs its bytes to move at
        # >= 0.85 of what a plain copy of the same read / write mix reaches on this part (6.29 TB/s, MI355X_MICROARCH.md) AND the kernel
        # to speed up when its input bytes are halved (the 24-bit ingest run of the SAME kernel: < 0.8 of the fp64-input time); "valu"
        # needs >= 0.8 of the vector unit's issue rate.  Neither: "latency" (the tile's dependent phases -- load, exchange, barrier --
        # at the occupancy its registers and LDS image allow).
        hbm_frac_of_copy = achieved / (COPY_CEILING_GBPS)
        halved = (le24["front_ms"] / kt[0]) if (le24 and k == 0 and "front_ms" in le24 and kt[0] > 0) else None
        if valu_dom and valu_dom["frac"] >= 0.8:
            bound = "valu"
        elif hbm_frac_of_copy >= 0.85 and (halved is None or halved < 0.8):
            bound = "hbm"
        else:
            bound = "latency"
        bound_evidence = {"hbm_frac_of_peak": achieved / HBM_PEAK_GBPS, "hbm_frac_of_measured_copy_ceiling": hbm_frac_of_copy, "copy_ceiling_GBps": COPY_CEILING_GBPS,
                          "valu_frac": valu_dom["frac"] if valu_dom else None,
                          "time_with_input_bytes_halved_over_time": halved,
                          "rule": "hbm: >= 0.85 of the copy ceiling and < 0.8 of the time with 24-bit input; valu: >= 0.8 of 39.3 T lane-ops/s; else latency"}
        line = {
            "metric": "Mcomplex-samples/s through RXA chain",
            "value": value,
            "unit": "Mcomplex-samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%d channels/GPU x 192 kHz IQ -> 48 kHz SSB RXA chain (shift + 561-tap resample/4 + adc meter + NBP nc 2048 "
                                   "+ S meter + fixed AGC + agc meter + panel), 2^%d input samples per channel per step" % (nch, args.log2_samples),
                       "meters": args.meters, "ingest": args.ingest, "channels_per_gpu": nch, "total_channels": total_channels,
                       "in_rate": IN_RATE, "dsp_rate": DSP_RATE, "dsp_size": DSP_SIZE,
                       "parallelism": "channel-sharded x%d (%s), no collective" % (world, scaling)},
            "chain_algorithmic_GBps": 20.0 * total / dt / 1e9,
            # ~650 flop per input sample if the chain is evaluated in direct form like the reference; the
            # overlap-save kernels execute about a quarter of that, which is how `value` can sit above the fp64-vector bound
            # the survey derives from this count (78.6e12 / 650 = 121 Gsamp/s)
            "chain_direct_form_equivalent_TFLOPs": 650.0 * total / dt / 1e12,
            "kernel_ms": {"front_shift_resample": kt[0], "band_nbp": kt[1], "state_bookkeeping": kt[2]},
            "roofline": {"bound": bound, "bound_evidence": bound_evidence, "kernel": names[k], "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_note": traffic_note,
                         "algorithmic_bytes_per_launch": algo[k] * samples_per_step,
                         "valu": None if not valu_dom else {"achieved": valu_dom["lane_ops_per_s"], "peak": VALU_PEAK_LANE_OPS, "unit": "VALU lane-ops/s",
                                                            "frac": valu_dom["frac"],
                                                            "note": "SQ_INSTS_VALU x 64 per launch (stamped counter pass, profiles/c2_traffic.json) / the kernel's HIP-event time"},
                         "per_kernel": valu},
            "check_inband_gain": gain,
        }
        if le24 is not None:
            line["ingest_le24"] = le24
        if hostfed is not None:
            line["host_fed"] = hostfed
        if world > 1:
            line["per_rank"] = per_rank
        if dry:
            line["dry_run"] = True
            line["dry"] = {"rank0_channels": [first, first + nch], "setters": eng.setters}
        if dt_off is not None:
            line["value_meters_off"] = total / dt_off / 1e6
            line["ms_per_step_meters_off"] = dt_off / args.steps * 1e3
            line["check_meters_dB"] = {"S_AV": meter_db[0], "ADC_AV": meter_db[1], "AGC_AV": meter_db[2]}
        free_first = world == 1 and not dry and args.ingest == "f64" and (not args.no_other_configs or not args.no_live_traffic)
        if free_first:
            del x, y, eng, tail                                  # the headline workload's 21 GB go back before the child processes allocate
            torch.cuda.empty_cache()
        # roofline.traffic measured in THIS run (the stamped file could not be confirmed by the driver's own run)
        if world == 1 and not dry and args.ingest == "f64" and not args.no_live_traffic and nchunk == 1:
            try:
                lt = live_traffic(args.meters)
                rf = line["roofline"]
                rf["traffic_stamped"] = rf["traffic"]
                rf["traffic"] = lt["front" if k == 0 else "band"] * samples_per_step
                rf["traffic_note"] = ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes over a child run of bench.py at 2^%d samples per "
                                      "channel, bytes per input sample x this launch's samples (FETCH_SIZE x 2: gfx950 tallies 128-byte requests at 64 bytes); "
                                      "traffic_stamped = the same from profiles/c2_traffic.json when its source stamp matches" % lt["child_log2_samples"])
                rf["traffic_live"] = lt
            except Exception as exc:                             # reported, never required: the stamped figure (or null) stays
                line["roofline"]["traffic_live"] = {"failed": repr(exc)[:400]}
        if not args.no_other_configs and world == 1 and not dry and args.ingest == "f64":
            line["other_configs"] = other_configs()
        if not args.no_cpu_baseline and world == 1 and not dry:       # the CPU baseline is reported by the single-GPU run only
            try:
                line["cpu_baseline"] = cpu_baseline()
            except Exception as exc:                             # the baseline is reported, never required
                line["cpu_baseline"] = {"value": None, "unit": "Mcomplex-samples/s", "cores": 0, "kind": "port",
                                        "sample": "failed: %r" % (exc,)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
