"""process_agc: the chain kernel against the sample-by-sample kernel -- bit identity and time (tools/dbg)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import quisk_amd as qh

dev = torch.device("cuda:0")
nch = int(os.environ.get("NCH", "256")); n = 1 << int(os.environ.get("LOG2N", "18")); calls = int(os.environ.get("CALLS", "3"))
rg = float(os.environ.get("RG", "5000")); cpx = int(os.environ.get("CPX", "0"))
torch.manual_seed(1)
s = torch.cuda.current_stream(dev).cuda_stream
def signal():
    x = torch.randn((nch, n + 64), dtype=torch.float64, device=dev)
    x = (x[:, :-8] + x[:, 1:-7] + x[:, 2:-6] + x[:, 3:-5] + x[:, 4:-4])[:, :n] * 2.0 ** 20      # low-passed noise
    lvl = 1.0 + 3.0 * (torch.arange(n, device=dev) // 30000 % 3 == 1)                             # level steps
    x = x * lvl
    return (x + 1j * torch.roll(x, 5, 1)).contiguous()
xs = [signal() for _ in range(calls)]
outs = {}
forms = [int(v) for v in os.environ.get('FORMS', '1,0').split(',')]
for form in forms:
    a = qh.QuiskAgc(nch, 48000, is_cpx=bool(cpx), stream=s)
    a.set_agc(-1, rg); a.debug_form(form)
    y0 = torch.empty((nch, 64), dtype=torch.complex128, device=dev)
    a.process2_ptr(xs[0].data_ptr(), n, y0.data_ptr(), 64, 64)          # the first call only initialises
    ys = []
    ts = []
    for x in xs:
        y = torch.empty_like(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        a.process2_ptr(x.data_ptr(), n, y.data_ptr(), n, n)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ys.append(y)
    outs[form] = torch.cat(ys, 1)
    print("form %d: ms per call %s" % (form, ["%.3f" % (t * 1e3) for t in ts]), flush=True)
d = (outs[forms[1]] - outs[forms[0]]).abs() / outs[forms[0]].abs().clamp_min(1.0)
print("bit-identical:", bool(torch.equal(torch.view_as_real(outs[forms[0]]), torch.view_as_real(outs[forms[1]]))))
print("max rel dev %.3g   samples over 1e-9: %d  over 1e-6: %d  of %d" % (float(d.max()), int((d > 1e-9).sum()), int((d > 1e-6).sum()), d.numel()))
worst = d.max(1).values
print("channels with dev > 1e-9:", int((worst > 1e-9).sum()), " rms out %.3g" % float(outs[forms[0]].abs().pow(2).mean().sqrt()))
if os.environ.get("FIRST"):
    for c in range(min(nch, 4)):
        idx = torch.nonzero(d[c] > 1e-9)
        if idx.numel():
            i = int(idx[0])
            print("ch", c, "first dev at", i, "tile", (i % n) // 512, "in-tile", (i % n) % 512, "rel", float(d[c, i]), "count", idx.numel())
