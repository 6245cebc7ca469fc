"""Where a block's time goes in pan16k_kernel<4> (config 3's panadapter): the shader clock at the phase boundaries of workgroup 0's
blocks, from a library built with -DQH_PAN_TRACE (build it here first:  python tools/dbg/pan_trace.py build ; then on the GPU:
QUISKHIP_LIB=quisk_amd/lib/libquiskhip_trace.so python tools/dbg/pan_trace.py)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
TRACE_LIB = os.path.join(ROOT, "quisk_amd", "lib", "libquiskhip_trace.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    from quisk_amd import build
    # QH_PAN_TRACE = the thread of workgroup 0 that leaves the stamps (0: first lane of the first wavefront; 960: first lane of the last)
    build.build(defines=["QH_PAN_TRACE=" + (sys.argv[2] if len(sys.argv) > 2 else "0")], out=TRACE_LIB, verbose=False)
    sys.exit(0)
import numpy as np, torch
sys.path.insert(0, os.path.join(ROOT, "tools"))
import quisk_amd as qh
import bench_configs as bc
dev = torch.device("cuda", 0)
L = bc.setup_config3(torch, qh, dev)
sync = lambda: torch.cuda.synchronize(dev)
which = sys.argv[1] if len(sys.argv) > 1 else "pan"
step = {"pan": L.step_pan, "both": L.step_both}[which]
t = bc.timed(step, sync, steps=10, warmup=3)
lib = qh.load()
buf = np.zeros(64 * 8, dtype=np.uint64)
assert lib.qh_pan_debug_trace(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
tr = buf.reshape(64, 8)[:16, :7].astype(np.int64)
span = tr[-1, 5] - tr[0, 0]
print("%s: %.3f ms per step; workgroup 0: 16 blocks in %d clocks" % (which, t * 1e3, span))
names = ["loads + window + radix-4 (this thread)", "... the rest of the workgroup (barrier)", "exchange between the groups", "4096-point transform", "|X| and sums"]
d = np.diff(tr[:, :6], axis=1)
for k, nm in enumerate(names):
    print("  %-45s %8.0f clocks (%4.1f %%)" % (nm, d[:, k].mean(), 100 * d[:, k].sum() / span))
a = tr[1:, 6] - tr[:-1, 5]
b = tr[1:, 0] - tr[1:, 6]
print("  %-45s %8.0f clocks (%4.1f %%)" % ("loop edge (spill stores / reloads)", a.mean(), 100 * a.sum() / span))
print("  %-45s %8.0f clocks (%4.1f %%)" % ("barrier at the top of the block", b.mean(), 100 * b.sum() / span))
