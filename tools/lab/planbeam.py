"""tools/lab/planbeam.py -- beam search over per-block-row choices of the fused sweep's plan (which rows ride the diagonal-block
launch / the row solve's tail: E eager, L lazy with the next-launch rule, D / F fill the launch only, nearest / farthest rows
first, N nothing optional) under the measured launch costs (one-source launch 29 us, two-source 43.5, row solve 9.5 + 0.115 per
tail tile).  Prints the best sweep time found and its plan; run_sweep's lazy rule is within 0.7 % of it.  python tools/lab/planbeam.py"""
import math, sys
nb=32; cap=255; tA=29.0; tB=43.5; t0=27.5; tT=9.5; cT=0.115; bt=96
tiles_of=lambda r,k: (nb-r)+k
def step(state,k,mode):
    done,ddone=list(state[0]),list(state[1])
    used=0; two=0; one=0
    if k>=1 and done[k]<k:
        pk=k-done[k]
        if pk>2: return None
        n=tiles_of(k,k)-1; used+=n
        if pk==2: two+=n
        else: one+=n
        done[k]=k; ddone[k]=k
    for r in range(k+1,nb):
        if k-done[r]>=2:
            if k-done[r]>2: return None
            n=tiles_of(r,k); used+=n; two+=n; done[r]=k; ddone[r]=k
    reserve = 1 if (k+1<nb and k>=1 and ddone[k+1]<k) else 0
    tt=0
    def take_d(r):
        nonlocal used,one
        n=tiles_of(r,k)
        if used+reserve+n<=cap: used+=n; one+=n; done[r]=k; ddone[r]=k; return True
        return False
    def take_t(r):
        nonlocal tt
        n=tiles_of(r,k)
        if r>=k+2 and k>=1 and k+1<nb and tt+n<=bt: tt+=n; done[r]=k; ddone[r]=k; return True
        return False
    behind=[r for r in range(k+1,nb) if k-done[r]==1]
    if mode=='E':
        for r in behind:
            if not take_d(r): take_t(r)
    elif mode=='D':
        for r in behind: take_d(r)
    elif mode=='F':      # far rows first into D
        for r in reversed(behind): take_d(r)
    elif mode=='L':
        nxt=[r for r in range(k+2,nb) if k-done[r]==1]
        load=sum(tiles_of(r,k+1) for r in nxt)+(nb-1)+1
        for r in nxt:
            if load<=cap: break
            if take_d(r) or take_t(r): load-=tiles_of(r,k+1)
    elif mode=='N': pass
    if k+1<nb and k>=1 and ddone[k+1]<k:
        p=k-ddone[k+1]
        if p>2: return None
        used+=1
        if p==2: two+=1
        else: one+=1
        ddone[k+1]=k
    if used>cap: return None
    cD=(t0 if used==0 else (tA if two==0 else tB))
    c=cD+tT+cT*tt
    return (tuple(done),tuple(ddone)),c,(k,mode,one,two,tt)
beam={ (tuple([0]*nb),tuple([0]*nb)): (0.0,[]) }
for k in range(nb):
    nxt={}
    for st,(c,path) in beam.items():
        for mode in "ELDFN":
            r=step(st,k,mode)
            if r is None: continue
            s2,dc,info=r
            if s2 not in nxt or nxt[s2][0]>c+dc: nxt[s2]=(c+dc,path+[info])
    items=sorted(nxt.items(), key=lambda kv: kv[1][0])[:3000]
    beam=dict(items)
    print(k,len(nxt),"best %.0f"%items[0][1][0], file=sys.stderr)
best=min(beam.values(), key=lambda v:v[0])
print("sweep %.0f"%best[0])
for i in best[1]: print(i)
