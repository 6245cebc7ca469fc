import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, quisk_amd as qh
dev = torch.device("cuda", 0)
n = 1 << 26
s = torch.cuda.current_stream(dev).cuda_stream
x = torch.randn((1, n), dtype=torch.float32, device=dev) + 1j * torch.randn((1, n), dtype=torch.float32, device=dev)
y = torch.empty((1, n >> 1), dtype=torch.complex64, device=dev)
NS = int(os.environ.get("NS", "8"))
for seg in sys.argv[1:]:
    os.environ["QH_HBC_SEG_STEPS"] = seg
    c = qh.HalfBandCascade(1, NS, dtype=1, stream=s)
    for _ in range(3): c.process_ptr(x.data_ptr(), n, n, y.data_ptr(), n >> 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): c.process_ptr(x.data_ptr(), n, n, y.data_ptr(), n >> 1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(json.dumps({"ns": NS, "seg_steps": int(seg), "ms": dt * 1e3, "GBps": 8 * n / dt / 1e9}), flush=True)
