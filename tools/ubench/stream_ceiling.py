"""What a plain streaming kernel reaches on this box: the practical ceiling the roofline fractions can be read against.
torch reductions / copies over 16 GiB (the size of the headline call's input), best of several."""
import time
import torch

dev = torch.device("cuda:0")
n = 1 << 30
x = torch.empty(n, dtype=torch.complex128, device=dev)
x.real.normal_(); x.imag.normal_()
y = torch.empty(n // 4, dtype=torch.complex128, device=dev)
xr = x.view(torch.float64)


def best(fn, reps=6):
    fn(); torch.cuda.synchronize()
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t0)
    return b

t = best(lambda: xr.sum())
print("read 16 GiB (sum of float64): %.3f ms = %.2f TB/s" % (t * 1e3, 16 * n / t / 1e12))
t = best(lambda: torch.abs(xr).max())
print("read 16 GiB (max |.|):        %.3f ms = %.2f TB/s" % (t * 1e3, 16 * n / t / 1e12))
z = torch.empty(n // 2, dtype=torch.complex128, device=dev)
t = best(lambda: z.copy_(x[: n // 2]))
print("copy 8 GiB -> 8 GiB:          %.3f ms = %.2f TB/s (read + write)" % (t * 1e3, 16 * n / t / 1e12))
t = best(lambda: torch.add(x[0::4], x[1::4], out=y))
print("strided read:                 %.3f ms" % (t * 1e3))
