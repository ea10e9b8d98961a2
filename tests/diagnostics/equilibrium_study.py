#!/usr/bin/env python3
"""Are the slowest prior-wide walkers slaved to the spin equilibrium?  (CPU; developer study)
Along the serial oracle's trajectory of three of the slowest prior-wide walkers of tools/adaptive_tilelog.py (and of the Humped
truth) it prints, every 300 grid points: h*lambda of a step over 8 grid intervals, the distance of omega from the root of
omega_dot(omega; Mdisc(t)) = 0 (Newton from the trajectory's value; nan: no root nearby) and d ln(omega) / d ln(t).
Result (round 4): from t ~ 10 s on these walkers track the equilibrium to 1e-3 ... 2e-2, the distance being the first-order lag
(d ln omega_eq / d ln t) / (lambda t) to within 10 %; a start of the sweeps from the bare equilibrium would be no better than the
log-space extrapolation, one from the lag-corrected equilibrium would be good to ~1e-6 at late times -- where the tiles
already converge in two sweeps and what holds the stride down is the indicator, not the guess.
    python tests/diagnostics/equilibrium_study.py"""
import os
import sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import c_oracle as co
tgrid = np.logspace(0,6,10001); cfg = co.cfg_synth()
cases = {"w857":[7.158, 9.24, -2.603, 2.926, -0.938, 2.764], "w698":[5.087, 6.664, -2.576, 2.48, -1.821, 2.644], "w148":[7.085, 7.939, -2.467, 3.201, -1.902, 2.286], "truth":[1,5,-3,2,-1,0]}
for name,p in cases.items():
    p=np.array(p,float); phys=p.copy(); phys[2:]=10.0**phys[2:]
    st,M,W=co.trajectory(cfg,phys,tgrid)
    print(name,'status',st)
    rows=[]
    for i in range(100,10000,300):
        (dM,f),lam=co.rhs(cfg,phys,tgrid[i],M[i],W[i])
        h=tgrid[i]*(1-10**(-6*8/10000))  # stride-8 step
        # Newton for equilibrium from current omega
        w=W[i]; ok=True
        for it in range(30):
            (d_,ff),ll=co.rhs(cfg,phys,tgrid[i],M[i],w)
            if ll==0: ok=False;break
            wn=w-ff/ll
            if not (wn>0): ok=False;break
            if abs(wn-w)<1e-12*w: w=wn;break
            w=wn
        rows.append((i, tgrid[i], h*lam, (w/W[i]-1) if ok else np.nan, f*tgrid[i]/W[i]))
    for r in rows: print("  i=%5d t=%9.2e  h*lam(stride8)=%9.2e  w_eq/w-1=%9.2e  dlnw/dlnt=%9.2e"%r)
