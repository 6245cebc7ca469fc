import numpy as np, sys
sys.path.insert(0,'/root/repo')
from quisk_amd import synth
import math
rate=48000.0
tau_attack=0.001; tau_decay=0.250; n_tau=4
A=int(math.ceil(rate*n_tau*tau_attack))
am=1-math.exp(-1/(rate*tau_attack)); dm=1-math.exp(-1/(rate*tau_decay)); fdm=1-math.exp(-1/(rate*0.005))
fbm=1-math.exp(-1/(rate*0.25)); hbm=1-math.exp(-1/(rate*0.5)); hdm=1-math.exp(-1/(rate*0.1))
out_target=(1-math.exp(-4))*0.9999; max_gain=10000.0; var_gain=1.5
min_volts=out_target/(var_gain*max_gain); hang_level=0.637; pop=5.0
def run(mag, v0, sv0, n0=0, n1=None):
    n=len(mag) if n1 is None else n1
    v=v0; sv=sv0; st=0; hc=0; dt=0; fba=0.1; hba=0.1
    vs=np.zeros(n-n0); att=0
    for j in range(n0,n):
        lo=max(0,j-A+1); rm=mag[lo:j+1].max()
        ao=mag[j-A] if j>=A else 0.0
        fba=fbm*ao+(1-fbm)*fba; hba=hbm*ao+(1-hbm)*hba
        if hc>0: hc-=1
        up=rm>=v
        if up: att+=1
        if st==0:
            if up: v+=(rm-v)*am
            elif v>pop*fba: st=1; v+=(rm-v)*fdm
            elif hba>hang_level: st=2; hc=0; dt=1
            else: st=3; v+=(rm-v)*dm; dt=0
        elif st==1:
            if up: st=0; v+=(rm-v)*am
            elif v>sv: v+=(rm-v)*fdm
            elif hc>0: st=2
            elif dt==0: st=3; v+=(rm-v)*dm
            else: st=4; v+=(rm-v)*hdm
        elif st==2:
            if up: st=0; sv=v; v+=(rm-v)*am
            elif hc==0: st=4; v+=(rm-v)*hdm
        elif st==3:
            if up: st=0; sv=v; v+=(rm-v)*am
            else: v+=(rm-v)*dm
        else:
            if up: st=0; sv=v; v+=(rm-v)*am
            else: v+=(rm-v)*hdm
        if v<min_volts: v=min_volts
        vs[j-n0]=v
    return vs, att
# steady two-tone + noise at 48k after the band filter: emulate: tone amplitude 0.1 + complex noise sigma (in-band share)
rng=np.random.default_rng(1)
n=60000
for sig in (0.0, 0.001, 0.003, 0.01):
    z=0.1*np.exp(2j*np.pi*0.02*np.arange(n))+sig*(rng.standard_normal(n)+1j*rng.standard_normal(n))
    mag=np.abs(z)
    va,att=run(mag,0.1,0.1)
    vb,_=run(mag,0.103,0.1)
    d=np.abs(va-vb)/va
    def first_below(t):
        idx=np.nonzero(d<t)[0]
        return idx[0] if len(idx) else -1
    print("sigma",sig,"attack fraction %.3f"%(att/n),"steps to 1e-6:",first_below(1e-6),"1e-9:",first_below(1e-9),"1e-12:",first_below(1e-12))
