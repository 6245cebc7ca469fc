"""The launch the driver uses for N > 1 -- `python -m torch.distributed.run ... bench.py --gpus N` over RCCL -- with the one rank a
one-GPU box allows: rendezvous on 127.0.0.1, the nccl barrier and max-over-ranks around the timed steps, the JSON line of rank 0.
Also `python bench.py` without a launcher.  Child processes only (this process may hold the GPU already).  -m gpu."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--steps", "3", "--warmup", "1", "--log2-samples", "18", "--no-cpu-baseline"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _line(cmd):
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    js = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(js) == 1, r.stdout[-1000:]
    return json.loads(js[0])


def _check(j):
    assert j["metric"] == "Mcomplex-samples/s through RXA chain" and j["unit"] == "Mcomplex-samples/s"
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["higher_is_better"] is True
    assert j["scaling"] == "weak" and j["dtype"] == "f64" and j["data"] == "synthetic" and j["vs_baseline"] is None
    assert j["config"]["meters"] == "on" and j["config"]["channels_per_gpu"] == 256
    want = 256 * (1 << 18) / (j["ms_per_step"] * 1e-3) / 1e6
    assert abs(j["value"] - want) < 1e-6 * want and j["value"] > 1e4
    r = j["roofline"]
    assert r["bound"] in ("hbm", "valu", "latency") and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.05 < r["frac"] < 1.0
    ev = r["bound_evidence"]            # the label is a rule over measured numbers, and the numbers ride along
    assert abs(ev["hbm_frac_of_peak"] - r["frac"]) < 1e-9 and 0.0 < ev["hbm_frac_of_measured_copy_ceiling"] < 1.2
    if r["bound"] == "hbm":
        assert ev["hbm_frac_of_measured_copy_ceiling"] >= 0.85
    assert abs(j["check_inband_gain"] - 4.0) < 1e-2


def test_torchrun_one_rank_over_rccl():
    j = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), "bench.py", "--gpus", "1", "--no-live-traffic"] + ARGS)
    _check(j)
    assert "x1" in j["config"]["parallelism"]


def test_plain_python_launch():
    j = _line([sys.executable, "bench.py"] + ARGS)
    _check(j)
    hf = j["host_fed"]                  # what a receiver whose samples arrive on the host gets, never `value`
    for kind, bps in (("f64", 16), ("le24", 6)):
        assert hf[kind]["bytes_per_sample_in"] == bps and 0.0 < hf[kind]["Msamp_per_s"] < j["value"]
        assert hf[kind]["link_GBps_in"] < 70.0          # a PCIe Gen5 x16 link
    # roofline.traffic is counted in the run itself (two rocprofv3 --pmc child passes): the dominant kernel's HBM bytes per launch, within a
    # few per cent of its algorithmic bytes (the front kernel re-reads its tiles' pre-roll from L2, not from HBM)
    r = j["roofline"]
    live = r.get("traffic_live")
    assert live and "failed" not in live, live
    assert r["traffic"] is not None and 0.9 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.3, (r["traffic"], r["algorithmic_bytes_per_launch"])
    assert 15.0 < live["front"] < 26.0 and 6.0 < live["band"] < 12.0
