"""How close is the flanger kernel's lock-step schedule to the dependency DAG?  For random flanger clips (the shipped parameter
ranges) count (a) the lock-steps csrc/flanger.hip takes (per 256-sample chunk: G = shortest read-after-write distance, 64/min(G,64)
steps per row of 64) and (b) the steps of an ideal greedy schedule with maximal dependency-free runs.  The kernel is within
~1.15x of (b) on the slowest clips: its duration is the clip's dependency chain, not a scheduling artefact.
    python tools/flanger_hops.py
"""
import numpy as np, math, sys
sys.path.insert(0,'/root/repo')
import torch
from oracle import modulations as omod, util as outil
rng=np.random.default_rng(0)
N=88200; M=44+441
shapes=["cos","rect_cos","inv_rect_cos","tri","saw","rsaw"]
res=[]
for c in range(40):
    rate=math.exp(rng.uniform(math.log(0.5),math.log(3.0))); phase=rng.uniform(0,2*math.pi); sh=shapes[rng.integers(6)]
    lfo=omod.make_mod_signal(882,441.0,rate,phase,sh)
    mod=outil.linear_interpolate_last_dim(lfo.unsqueeze(0),N)[0].numpy()
    mdw=rng.uniform(0,1); width=rng.uniform(0.25,1)
    d=441*width*mod+mdw*44          # delay in samples
    # integer dependency distance: min(w-prev, w-next) in samples ~ floor(d) (and d<1 -> reads slot "M ago")
    dep=np.floor(d).astype(int); dep=np.where(dep<=0, M, dep)   # next = prev+1 -> distance floor(d); if floor(d)==0 next is w itself ("M samples ago")
    # kernel: per 256-chunk G=min(dep) -> steps = sum(ceil(64/min(G,64)))*4 if G<256 else 1 (approx: 4 rows of 64)
    k_steps=0; a_steps=0
    for c0 in range(0,N,256):
        dd=dep[c0:c0+256]; G=dd.min()
        if G>=256: k_steps+=1
        else:
            gr=min(G,64); k_steps+=4*math.ceil(64/gr)
        # adaptive greedy runs across the chunk
        i=0; n=len(dd)
        while i<n:
            # longest L with dd[i+j] > j for all j<L
            j=np.arange(n-i); ok=dd[i:]>j
            L=n-i if ok.all() else int(np.argmin(ok))
            L=max(L,1); L=min(L,256); i+=L; a_steps+=1
    res.append((mdw,width,sh,k_steps,a_steps))
res.sort(key=lambda r:-r[3])
for r in res[:8]: print("mdw %.2f width %.2f %-12s kernel lock-steps %6d adaptive %6d ratio %.2f"%(r[0],r[1],r[2],r[3],r[4],r[3]/r[4]))
print("max kernel",max(r[3] for r in res),"max adaptive",max(r[4] for r in res), "mean kernel", np.mean([r[3] for r in res]), "mean adaptive", np.mean([r[4] for r in res]))
