import sys, numpy as np, ctypes as C
sys.path.insert(0, '.')
import quisk_amd as qh
from quisk_amd import synth
nch, nblk = 3, 256
n_in = nblk * 1024
x = np.stack([synth.make_mode_input_numpy("fm", c, n_in) for c in range(nch)])
L = qh.load()
L.qh_rxa_debug_pll.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
for check_only in (1, 0):
    eng = qh.RxaEngine(nch)
    for c in range(nch):
        eng.SetRXAShiftRun(c, 1); eng.SetRXAShiftFreq(c, synth.shift_freq(c)); eng.RXANBPSetRun(c, 1)
        eng.SetRXAMode(c, 5); eng.SetRXAAGCMode(c, 0); eng.SetRXAAGCFixed(c, 0.0); eng.RXASetPassband(c, -8000.0, 8000.0)
    L.qh_rxa_debug_pll(eng._h, check_only, 0, None, 0)
    prev = 0
    for k in range(4):
        y = eng.process_host(x)
        r = eng.pll_repairs(); print("check_only", check_only, "call", k, "flagged", r - prev); prev = r
        if k == 2:
            for c in range(nch):
                buf = np.zeros(256 * 6)
                L.qh_rxa_debug_pll(eng._h, -1, c, buf.ctypes.data, buf.size)
                e = buf.reshape(256, 6)
                dp = e[1:, 0] - e[:-1, 3]; dp -= np.rint(dp)
                df = e[1:, 1] - e[:-1, 4]; do = e[1:, 2] - e[:-1, 5]
                m = np.maximum(np.abs(dp), np.maximum(np.abs(df), np.abs(do)))
                bad = np.nonzero(m[3:] > 1e-12)[0] + 4
                print(" ch", c, "bad tiles", bad[:20], "n", bad.size)
                for t in bad[:6]:
                    print("   tile", t, "dpt %.3e dfil %.3e dom %.3e" % (dp[t - 1], df[t - 1], do[t - 1]), "warm", e[t, :3], "prev end", e[t - 1, 3:])
