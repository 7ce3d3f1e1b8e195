#!/usr/bin/env python3
"""Development aid (CPU): the first pass of K34 in its level form, lane by lane in Python — the middle-key cut of a two-segment tile,
the 12-or-1 probes of a chunk, the level-by-level emission into the ring of 128 descriptors, the rounds of 64 — against the pairs a
brute-force join of the tile gives (mimeo_amd/csrc/k34_fused.hip: do_tile, "level emission").  What it checks is the LOGIC the kernel was
written from before it met a GPU: every pair exactly once, the ring never overwritten, at most 127 descriptors waiting.  The kernel itself
is checked on the GPU (scripts/gpu_k34_ab.py, tests/test_gpu_segments.py)."""
import numpy as np
rng = np.random.default_rng(1)
TILE=4096; QSEG=1280; RING=128
def run_tile(nQ_mean, nT, transitions=True, force_unaligned=False):
    # query offsets
    qcnt = rng.poisson(nQ_mean/TILE, TILE)
    # a heavy key sometimes
    if rng.random()<0.3: qcnt[rng.integers(0,TILE)] += rng.integers(50,300)
    ro = np.concatenate([[0], np.cumsum(qcnt)]); nQ = ro[-1]
    tkeys = np.sort(rng.integers(0,TILE,nT))
    nn = 13 if transitions else 1
    # expected pairs: (target entry e, query entry qi_abs)
    exp=set()
    for e,w in enumerate(tkeys):
        for j in range(nn):
            w2 = w ^ (1<<(j-1)) if j else w
            for qi in range(ro[w2], ro[w2+1]): exp.add((e,qi))
    got=[]
    aligned=False; mid=0
    if nQ>QSEG and not force_unaligned:
        mid=ro[TILE//2]; aligned = mid<=QSEG and nQ-mid<=QSEG
    qs=0
    nrounds=0
    while qs<nQ:
        qe = (nQ if qs else mid) if aligned else min(qs+QSEG,nQ)
        assert qe>qs and qe-qs<=QSEG
        half = 1 if qs else 0
        sQ = np.minimum(np.maximum(ro,qs),qe)-qs
        for ch in range(0,nT,64):
            keys = tkeys[ch:ch+64]; ne=len(keys)
            jlo,jhi=0,nn
            if aligned:
                in_half = any((k>>11)==half for k in keys); in_other=any((k>>11)!=half for k in keys)
                if not in_other: jhi=min(nn,12)
                elif not in_half: jlo=12
            ring=[None]*RING; head=tail=pend=0
            def do_round(n):
                nonlocal head,pend,nrounds
                for lane in range(n):
                    d=ring[(head+lane)%RING]; assert d is not None
                    owner,qi=d>>16,d&0xFFFF
                    got.append((ch+owner, qs+qi))
                    ring[(head+lane)%RING]=None
                head=(head+64)%RING; pend-=n; nrounds+=1
            for j in range(jlo,jhi):
                a=np.zeros(64,int); cnt=np.zeros(64,int)
                for lane in range(ne):
                    w=keys[lane]; w2 = w ^ (1<<(j-1)) if j else w
                    a[lane]=sQ[w2]; cnt[lane]=sQ[w2+1]-sQ[w2]
                k=0
                while True:
                    m=[lane for lane in range(64) if cnt[lane]>k]
                    if not m: break
                    for t,lane in enumerate(m):
                        slot=(tail+t)%RING
                        assert ring[slot] is None, 'ring overwrite'
                        ring[slot]=(lane<<16)|(a[lane]+k)
                    tail=(tail+len(m))%RING; pend+=len(m); k+=1
                    assert pend<=RING
                    if pend>=64: do_round(64)
            if pend: do_round(pend)
            assert pend==0
        qs=qe
    assert len(got)==len(set(got)), 'duplicate pairs'
    assert set(got)==exp, (len(got),len(exp))
    return len(got), aligned
for nQm,nT in [(2441,2441),(2441,300),(1953,120),(1400,700),(2700,500),(300,300),(2560,64),(2441,65)]:
    for tr in (True,False):
        for fu in (False,True):
            n,al=run_tile(nQm,nT,tr,fu)
            print(nQm,nT,tr,fu,'pairs',n,'aligned',al)
print('emulation OK')
