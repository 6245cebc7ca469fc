"""bench.py's multi-process plumbing without a GPU: argument parsing, RANK / WORLD_SIZE handling, the weak and the strong
channel split, the barrier-bracketed timing, the max over ranks and the one JSON line of rank 0 -- run in two fresh child
processes over gloo with `--dry-run`, which puts bench.DryEngine where RxaEngine stands (no DSP: the printed line says
"dry_run": true and is not a measurement).  The real launch differs in the backend (nccl = RCCL) and the engine only."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(world, extra, trace=None):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        if trace:
            env["QH_BENCH_TRACE"] = trace
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--dry-run", "--backend", "gloo",
                                       "--steps", "4", "--warmup", "1", "--log2-samples", "12"] + extra,
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    js = [[l for l in so.splitlines() if l.startswith("{")] for so, _ in outs]
    assert len(js[0]) == 1 and all(not j for j in js[1:])       # rank 0 prints the one JSON line, nobody else
    return json.loads(js[0][0])


def test_strong_split_two_ranks():
    j = _launch(2, ["--total-channels", "256"])
    assert j["dry_run"] is True and j["n_gpus"] == 2 and j["steps"] == 4 and j["warmup"] == 1
    assert j["scaling"] == "strong"
    assert j["config"]["total_channels"] == 256 and j["config"]["channels_per_gpu"] == 128
    assert j["dry"]["rank0_channels"] == [0, 128]
    # rank 1's fake step takes 2 ms: the job time is the max over ranks, and value is the whole job's samples over it
    assert j["ms_per_step"] >= 2.0
    want = 256 * 4096 * 4 / (j["ms_per_step"] * 4e-3) / 1e6
    assert abs(j["value"] - want) < 1e-6 * want
    assert j["config"]["meters"] == "on" and "value_meters_off" in j


def test_weak_split_two_ranks_and_uneven_strong_split():
    j = _launch(2, ["--channels", "8"])
    assert j["scaling"] == "weak" and j["config"]["total_channels"] == 16 and j["config"]["channels_per_gpu"] == 8
    want = 16 * 4096 * 4 / (j["ms_per_step"] * 4e-3) / 1e6
    assert abs(j["value"] - want) < 1e-6 * want
    j = _launch(2, ["--total-channels", "5"])          # 3 + 2
    assert j["config"]["channels_per_gpu"] == 3 and j["config"]["total_channels"] == 5


def test_single_process_dry_run():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "2", "--warmup", "0", "--log2-samples", "12"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 1 and j["dry_run"] is True and j["scaling"] == "weak"


def _read_trace(path):
    ev = {}
    for line in open(path):
        r, what, t = line.split()
        ev.setdefault(what, {}).setdefault(int(r), []).append(float(t))
    return ev


def test_eight_ranks_strong_256_and_weak_2048(tmp_path):
    """The driver's 8-GPU shapes on eight gloo ranks: north_star's 256 channels over 8 GPUs (32 each, strong) and BASELINE config 4's
    2048 (256 each, weak).  Rank 0 alone is where the library gets built, ahead of the first barrier; no rank enters the timed region
    before every rank has left its warm-up, and none leaves it before every rank has done its steps."""
    for extra, per, total, scaling in ((["--total-channels", "256"], 32, 256, "strong"), (["--channels", "256"], 256, 2048, "weak")):
        trace = str(tmp_path / ("trace_%s.txt" % scaling))
        j = _launch(8, extra, trace=trace)
        assert j["n_gpus"] == 8 and j["scaling"] == scaling
        assert j["config"]["channels_per_gpu"] == per and j["config"]["total_channels"] == total
        assert j["dry"]["rank0_channels"] == [0, per]
        assert j["ms_per_step"] >= 8.0                    # rank 7's fake step takes 8 ms: max over ranks
        want = total * 4096 * 4 / (j["ms_per_step"] * 4e-3) / 1e6
        assert abs(j["value"] - want) < 1e-6 * want
        ev = _read_trace(trace)
        assert sorted(ev["build"]) == [0]                 # rank 0 builds, nobody else
        for k in (0, 1):                                  # two timed runs per launch: meters on, meters off
            warm = max(ev["warm_done"][r][k] for r in range(8))
            start = min(ev["timed_start"][r][k] for r in range(8))
            done = max(ev["steps_done"][r][k] for r in range(8))
            end = min(ev["timed_end"][r][k] for r in range(8))
            assert start >= warm - 1e-3 and end >= done - 1e-3
        assert min(ev["warm_done"][r][0] for r in range(8)) >= ev["build"][0][0]


def test_self_spawn_eight_ranks_without_a_launcher(tmp_path):
    """`python bench.py --gpus 8` with no WORLD_SIZE in the environment (a driver that does not use torch.distributed.run): bench.py
    starts the eight ranks itself, before it imports torch, and relays rank 0's one line; every rank's kernel times and channel
    range are on it."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    trace = str(tmp_path / "trace.txt")
    env["QH_BENCH_TRACE"] = trace
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--backend", "gloo", "--steps", "3",
                        "--warmup", "1", "--log2-samples", "12"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    js = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(js) == 1
    j = json.loads(js[0])
    assert j["n_gpus"] == 8 and j["dry_run"] is True and j["scaling"] == "weak" and j["config"]["total_channels"] == 2048
    assert j["ms_per_step"] >= 8.0
    assert [p["rank"] for p in j["per_rank"]] == list(range(8))
    assert [p["channels"] for p in j["per_rank"]] == [[256 * k, 256 * (k + 1)] for k in range(8)]
    assert all(set(p["kernel_ms"]) == {"front_shift_resample", "band_nbp", "state_bookkeeping"} for p in j["per_rank"])
    ev = _read_trace(trace)
    assert sorted(ev["timed_start"]) == list(range(8))


def test_a_failing_rank_fails_the_self_spawned_job():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--backend", "gloo", "--steps", "1",
                        "--warmup", "0", "--log2-samples", "12", "--total-channels", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0                 # rank 1 has no channel and says so


def test_eight_ranks_of_configs_4_and_5(tmp_path):
    """BASELINE configs 4 (2048 mixed-mode channels, 256 per GPU) and 5 (8 x 61.44 Msps, one channel per GPU) launch on N GPUs like
    config 2 does (`--config`): eight gloo ranks each, every rank's channel range, kernel time and roofline on rank 0's line, the
    timed region bracketed by barriers.  The strong form of config 4 (a fixed 256-channel job) too."""
    for extra, per, total, scaling in ((["--config", "4"], 256, 2048, "weak"), (["--config", "5"], 1, 8, "weak"),
                                       (["--config", "4", "--total-channels", "256"], 32, 256, "strong")):
        trace = str(tmp_path / ("trace_%s_%s.txt" % (extra[1], scaling)))
        j = _launch(8, extra, trace=trace)
        assert j["n_gpus"] == 8 and j["scaling"] == scaling and j["dry_run"] is True
        assert j["config"]["baseline_config"] == int(extra[1])
        assert j["config"]["channels_per_gpu"] == per and j["config"]["total_channels"] == total
        assert [p["rank"] for p in j["per_rank"]] == list(range(8))
        assert [p["channels"] for p in j["per_rank"]] == [[per * k, per * (k + 1)] for k in range(8)]
        assert all({"achieved", "peak", "frac", "unit", "kernel"} <= set(p["roofline"]) for p in j["per_rank"])
        assert j["ms_per_step"] >= 8.0                    # rank 7's fake step takes 8 ms: max over ranks
        want = total * 4096 * 4 / (j["ms_per_step"] * 4e-3) / 1e6
        assert abs(j["value"] - want) < 1e-6 * want
        assert {"bound", "achieved", "peak", "frac", "unit"} <= set(j["roofline"])
        ev = _read_trace(trace)
        warm = max(ev["warm_done"][r][0] for r in range(8))
        start = min(ev["timed_start"][r][0] for r in range(8))
        done = max(ev["steps_done"][r][0] for r in range(8))
        end = min(ev["timed_end"][r][0] for r in range(8))
        assert start >= warm - 1e-3 and end >= done - 1e-3


def test_config_2_line_carries_every_ranks_roofline():
    j = _launch(2, ["--channels", "8"])
    assert all({"achieved", "peak", "unit", "kernel"} <= set(p["roofline"]) for p in j["per_rank"])
    assert j["roofline"]["bound"] in ("hbm", "valu", "latency") and "bound_evidence" in j["roofline"]
