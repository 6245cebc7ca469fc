"""The config-2-with-AGC leg alone, steady and fading input: ms per step and the repair counters.  usage: agc_leg.py [steps]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import quisk_amd as qh
import bench_configs as bc
dev = torch.device("cuda:0")
r = bc.config2_agc(torch, qh, dev)
d = r["detail"] if "detail" in r else r
print(json.dumps({"warm": os.environ.get("QH_AGC_WARM"), "rounds": os.environ.get("QH_AGC_ROUNDS"), "steady_ms": round(d["ms"], 3), "steady_segs": d["agc_segments_rerun"], "steady_tiles": d["agc_tiles_rerun"],
                  "fading_ms": round(d["fading_input"]["ms"], 3), "fading_segs": d["fading_input"]["agc_segments_rerun"], "fading_tiles": d["fading_input"]["agc_tiles_rerun"]}), flush=True)
