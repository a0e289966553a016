"""How often does the row cache of the three-blocks-per-CU CM kernels miss?  CPU simulation (FIFO over the byte values of BWT output) for the plain and the
enwik8-calibrated text generator at 44 / 56 / 96 slots.  Test / analysis infrastructure (uses the oracle for LZP + BWT).   python tools/cm_row_cache_sim.py"""
import sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, datagen
from oracle_lib import Oracle
o=Oracle()
def sim(u, slots):
    # FIFO cache of rows keyed by byte value; access sequence = u (each byte's row is needed when it becomes c1)
    res=np.full(256,-1,dtype=np.int64); owner=np.full(slots,-1,dtype=np.int64); hand=0; used=0; miss=0
    prev=-1
    for c in u:
        if c==prev: continue
        prev=c
        if res[c]>=0: continue
        miss+=1
        if used<slots: s=used; used+=1
        else:
            s=hand; hand=(hand+1)%slots
            res[owner[s]]=-1
        owner[s]=c; res[c]=s
    return miss
for noise in (0.0, datagen.ENWIK_NOISE):
    n=8<<20
    t=datagen.text(n, seed=1, chains=1<<12, noise=noise)
    nl,lz=o.lzp_encode(t); src=lz if 0<nl<len(t) else t
    idx,u=o.bwt(src); u=np.frombuffer(u,dtype=np.uint8)
    vals,cnt=np.unique(u,return_counts=True)
    print("noise",noise,"distinct",len(vals),"rows covering 99.9%:", int((np.cumsum(np.sort(cnt)[::-1])/len(u) < 0.999).sum())+1, "repeat", float((u[1:]==u[:-1]).mean()))
    for slots in (44,56,96):
        m=sim(u.tolist(), slots)
        print("   slots",slots,"misses",m,"rate %.4f%%"%(100*m/len(u)))
