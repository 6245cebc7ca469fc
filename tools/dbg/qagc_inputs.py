"""process_agc alone (256 streams x 2^18 samples at 48 ksps) on inputs that make the machine turn more or less often: what of its time
is the chain of relax steps and what the turns (cycle starts every 720 samples, overloads, ramps)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import quisk_amd as qh
dev = torch.device("cuda:0")
nch, n = 256, 1 << 18
st = torch.cuda.Stream(dev)
t = torch.arange(n, device=dev, dtype=torch.float64)
def run(name, x, gain):
    a = qh.QuiskAgc(nch, 48000, 0.7, 1.0, False, 0, st.cuda_stream)
    a.set_agc(-1, gain)
    y = torch.empty_like(x)
    for _ in range(3):
        a.process2_ptr(x.data_ptr(), n, y.data_ptr(), n, n)
    st.synchronize()
    t0 = time.perf_counter()
    K = 8
    for _ in range(K):
        a.process2_ptr(x.data_ptr(), n, y.data_ptr(), n, n)
    st.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("%-46s %.3f ms  %.1f clocks per sample at 2.1 GHz" % (name, dt * 1e3, dt / n * 2.1e9), flush=True)
    a.close()
with torch.cuda.stream(st):
    noise = (torch.randn((nch, n), dtype=torch.float64, device=dev) + 0j) * 2.0 ** 22
    tone = (torch.cos(2 * torch.pi * 1000.0 / 48000.0 * t)[None, :].repeat(nch, 1) + 0j) * 2.0 ** 22
    quiet = tone * 1e-6
    speech = tone * (0.05 + 0.95 * (torch.sin(2 * torch.pi * 3.0 / 48000.0 * t) ** 2))[None, :]
st.synchronize()
run("gaussian noise, release gain 5000 (the bench's)", noise, 5000.0)
run("steady tone, release gain 5000", tone, 5000.0)
run("steady tone, release gain 1 (never near the limit)", quiet, 1.0)
run("tone under a 3 Hz envelope (syllables)", speech, 5000.0)
