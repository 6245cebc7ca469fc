import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
def run(ns):
    import torch
    import quisk_amd as qh
    dev = torch.device("cuda:0")
    n = 1 << 26
    s = torch.cuda.current_stream(dev).cuda_stream
    x = torch.randn((1, n), dtype=torch.float32, device=dev) + 1j * torch.randn((1, n), dtype=torch.float32, device=dev)
    c = qh.HalfBandCascade(1, ns, dtype=1, stream=s)
    out = torch.empty((1, n >> ns), dtype=torch.complex64, device=dev)
    for _ in range(3): c.process_ptr(x.data_ptr(), n, n, out.data_ptr(), out.shape[1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30): c.process_ptr(x.data_ptr(), n, n, out.data_ptr(), out.shape[1])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 30 * 1e3
if len(sys.argv) > 1:
    print("%.4f" % run(int(sys.argv[1])))
else:
    for ns in (3, 8):
        for seg in (16, 24, 32, 43, 48, 64, 86):
            r = subprocess.run([sys.executable, __file__, str(ns)], capture_output=True, text=True, env=dict(os.environ, QH_HBC_SEG_STEPS=str(seg)))
            print("ns %d seg %3d (%4d wgs): %s ms" % (ns, seg, (32768 + seg - 1) // seg, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-200:]), flush=True)
