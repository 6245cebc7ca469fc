"""Where a block's time goes in pan16k_kernel<4> / panfir16k_kernel (config 3): the shader clock at the phase boundaries of workgroup
0's blocks, from a library built with -DQH_PAN_TRACE.  Build it here first:  python tools/dbg/pan_trace.py build [thread] ; then on the
GPU:  QUISKHIP_LIB=quisk_amd/lib/abx/libquiskhip_trace.so python tools/dbg/pan_trace.py [pan|fused]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
TRACE_LIB = os.path.join(ROOT, "quisk_amd", "lib", "abx", "libquiskhip_trace.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    from quisk_amd import build
    # QH_PAN_TRACE = the thread of workgroup 0 that leaves the stamps (0: first lane of the first wavefront; 960: first lane of the last)
    build.build(defines=["QH_PAN_TRACE=" + (sys.argv[2] if len(sys.argv) > 2 else "0")] + sys.argv[3:], out=TRACE_LIB, verbose=False)
    sys.exit(0)
import numpy as np, torch
sys.path.insert(0, os.path.join(ROOT, "tools"))
import quisk_amd as qh
import bench_configs as bc
dev = torch.device("cuda", 0)
L = bc.setup_config3(torch, qh, dev)
sync = lambda: torch.cuda.synchronize(dev)
which = sys.argv[1] if len(sys.argv) > 1 else "pan"
step = {"pan": L.step_pan, "fused": L.step_fused}[which]
t = bc.timed(step, sync, steps=10, warmup=3)
lib = qh.load()
buf = np.zeros(64 * 8, dtype=np.uint64)
assert lib.qh_dbg_pan_trace(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
tr = buf.reshape(64, 8)[:16].astype(np.int64)
span = tr[-1, 5] - tr[0, 6]
print("%s: %.3f ms per step; workgroup 0: 16 blocks in %d clocks of the shader clock (100 MHz counter x ?): per block %.0f" % (which, t * 1e3, span, span / 16.0))
# order of the stamps within a block: 6 (top), 0 (behind the first barrier), 1 (this thread's second round stored), 2 (barrier), 3 (the
# groups' exchange done), 4 (transform done), [7 (FIR's products and fold done)], 5 (|X| and sums done)
order = [6, 0, 1, 2, 3, 4] + ([7] if which == "fused" else []) + [5]
names = {(6, 0): "loads asked for + barrier at the top", (0, 1): "loads arrive, window, radix-4, stores (this thread)",
         (1, 2): "... the rest of the workgroup (barrier)", (2, 3): "second round read out + barrier", (3, 4): "4096-point transform",
         (4, 7): "FIR: products with H', fold", (7, 5): "window in the frequency domain, |X|, sums", (4, 5): "|X| and sums"}
for a, b in zip(order[:-1], order[1:]):
    d = tr[:, b] - tr[:, a]
    print("  %-55s %8.0f (%4.1f %%)" % (names[(a, b)], d.mean(), 100 * d.sum() / span))
e = tr[1:, 6] - tr[:-1, 5]
print("  %-55s %8.0f (%4.1f %%)" % ("loop edge", e.mean(), 100 * e.sum() / span))
