"""Config 2's 256 channels as ONE engine against TWO engines of 128 channels on two streams, the second half a step behind (its front
kernel beside the first's band kernel): does co-running the two kernels buy anything?  Experiment, not the bench."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import quisk_amd as qh
from quisk_amd import synth
dev = torch.device("cuda", 0)
LOG2 = int(sys.argv[1]) if len(sys.argv) > 1 else 21
n_in = 1 << LOG2; nblk = n_in // 1024; n_out = nblk * 256

def make(nch, first, stream):
    e = qh.RxaEngine(nch, stream=stream.cuda_stream)
    for c in range(nch):
        e.SetRXAShiftRun(c, 1); e.SetRXAShiftFreq(c, synth.shift_freq(first + c)); e.RXANBPSetRun(c, 1); e.SetRXAMode(c, 1)
        e.RXASetPassband(c, 300.0, 3000.0); e.SetRXAAGCMode(c, 0); e.SetRXAAGCFixed(c, 0.0)
    return e

x = torch.randn((256, n_in), dtype=torch.float64, device=dev).to(torch.complex128) * 0.1
y = torch.empty((256, n_out), dtype=torch.complex128, device=dev)
s0, s1 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
one = make(256, 0, s0)
a, b = make(128, 0, s0), make(128, 128, s1)
torch.cuda.synchronize(dev)

def step_one():
    one.process_ptr(x.data_ptr(), n_in, y.data_ptr(), n_out, nblk)
def step_two():
    a.process_ptr(x.data_ptr(), n_in, y.data_ptr(), n_out, nblk)
    b.process_ptr(x[128:].data_ptr(), n_in, y[128:].data_ptr(), n_out, nblk)

def timed(fn, steps=12, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps * 1e3

for rep in range(2):
    print("one engine of 256: %.3f ms   two engines of 128 on two streams: %.3f ms" % (timed(step_one), timed(step_two)), flush=True)
# offset: the second engine's call is enqueued behind an event recorded in the MIDDLE of the first's (after its front kernel: not reachable from
# outside the engine) -- approximate with half-length calls: a, b alternate halves of the step
h = nblk // 2
def step_two_offset():
    for k in range(2):
        a.process_ptr(x[:, k * h * 1024:].data_ptr(), n_in, y[:, k * h * 256:].data_ptr(), n_out, h)
        b.process_ptr(x[128:, k * h * 1024:].data_ptr(), n_in, y[128:, k * h * 256:].data_ptr(), n_out, h)
print("two engines, two half-length calls each: %.3f ms" % timed(step_two_offset), flush=True)
