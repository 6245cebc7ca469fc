"""one cascade shape, a few launches: for rocprofv3 --pmc / --kernel-trace (tools/dbg/hbc_pmc.sh (a one-off script, in git history))"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import quisk_amd as qh
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
n = 1 << 26
s = torch.cuda.current_stream(dev).cuda_stream
x = torch.randn((1, n), dtype=torch.float32, device=dev) + 1j * torch.randn((1, n), dtype=torch.float32, device=dev)
c = qh.HalfBandCascade(1, ns, dtype=1, stream=s)
out = torch.empty((1, n >> ns), dtype=torch.complex64, device=dev)
for _ in range(5):
    c.process_ptr(x.data_ptr(), n, n, out.data_ptr(), out.shape[1])
torch.cuda.synchronize()
