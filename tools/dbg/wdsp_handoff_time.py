"""the WDSP hand-off of quisk_process_samples (quisk.c:2660-2661) per call of 1000 samples at 48 ksps, in_size 256: host-pointer
wdspFexchange0 fed from / drained to the GPU (what the block API did through round 3) against qh_wdsp_fexchange0_device"""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quisk_amd as qh
lib = qh.load()
D = C.c_double
dev = torch.device("cuda", 0)
n, in_size, calls = 1000, 256, 400
for ch in (0, 1):
    lib.OpenChannel(ch, in_size, 256, 48000, 48000, 48000, 0, 1, D(0.010), D(0.025), D(0.0), D(0.010), 1)
    lib.SetRXAShiftRun(ch, 0); lib.RXANBPSetRun(ch, 0); lib.SetRXAAMSQRun(ch, 0); lib.SetRXAMode(ch, 1)
    lib.RXASetPassband(ch, D(300.0), D(3000.0)); lib.RXASetNC(ch, 256); lib.RXASetMP(ch, 0)
    lib.SetRXAAGCMode(ch, 0); lib.SetRXAAGCFixed(ch, D(0.0))
    lib.qh_wdsp_set_parameter(ch, in_size, 1)
x = (torch.randn(n + 2 * in_size, dtype=torch.complex128, device=dev) * 1e8)
st = torch.cuda.Stream(dev)
host = torch.zeros(n + 2 * in_size, dtype=torch.complex128).pin_memory()
def host_way():
    host[:n].copy_(x[:n], non_blocking=True); torch.cuda.current_stream().synchronize()
    k = lib.wdspFexchange0(0, C.c_void_p(host.data_ptr()), n)
    x[:k].copy_(host[:k], non_blocking=True); torch.cuda.current_stream().synchronize()
def dev_way():
    lib.qh_wdsp_fexchange0_device(1, C.c_void_p(x.data_ptr()), n, C.c_void_p(st.cuda_stream))
for name, f, tail in (("host rings", host_way, lambda: None), ("device rings", dev_way, st.synchronize)):
    with torch.cuda.stream(st):
        for _ in range(20): f()
        tail(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls): f()
        t_host = time.perf_counter() - t0
        tail(); torch.cuda.synchronize()
        t1 = time.perf_counter() - t0
    print("%-13s %.1f us per call on the host thread, %.1f us per call until the GPU is done" % (name, t_host / calls * 1e6, t1 / calls * 1e6))
