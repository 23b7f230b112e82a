"""tools/lab/plansim.py -- host model of a tile-by-tile planner of the fused sweep (not built): every
diagonal-block launch carries up to `cap` tiles, each taking one or two of its pending sources; a launch whose tiles all take
one source lasts tA, one with a two-source tile tB; the row solve tT.  Tiles have deadlines (their row's turn).  Prints the
sweep time of greedy launch-type policies for nb block rows.   python tools/lab/plansim.py"""
import math
def sim(nb=32, cap=255, tA=30.0, tB=43.5, tT=9.5, bias=0, verbose=False):
    nxt = {(r,j): (0 if j>=r else j) for r in range(nb) for j in range(nb)}
    total=0.0; log=[]
    for k in range(nb):
        tiles=[]
        for r in range(k, nb):
            for j in range(nb):
                if r==k and j==k: continue
                if j<r and j>=k: continue
                pend=k-nxt[(r,j)]
                if pend<=0: continue
                if r==k: req=pend
                else: req=max(0, pend-(r-k-1))
                tiles.append([r,j,pend,req])
        need2=[t for t in tiles if t[3]>=2]; need1=[t for t in tiles if t[3]==1]
        assert all(t[3]<=2 for t in tiles), (k,[t for t in tiles if t[3]>2][:3])
        # lookahead bias: also go B if the backlog (sum pend) exceeds what A steps could absorb before the end
        backlog=sum(t[2] for t in tiles)
        stepsleft=nb-k
        goB = len(need2)>0 or len(need1)>cap or (bias and backlog > bias*cap)
        chosen=[]
        if not goB:
            chosen=[(t,1) for t in need1]
            rest=sorted([t for t in tiles if t[3]==0], key=lambda t:(t[0]-k-t[2], t[0], t[1]))   # least slack first
            for t in rest:
                if len(chosen)>=cap: break
                chosen.append((t,1))
            cost=tA if chosen else 27.5
        else:
            chosen=[(t,min(2,t[2])) for t in need2+need1]
            rest=sorted([t for t in tiles if t[3]==0], key=lambda t:(-min(2,t[2]), t[0]-k-t[2], t[0], t[1]))
            for t in rest:
                if len(chosen)>=cap: break
                chosen.append((t,min(2,t[2])))
            rounds=max(1,math.ceil(len(chosen)/cap))
            cost=tB*rounds
        for t,s in chosen: nxt[(t[0],t[1])]+=s
        total+=cost+tT
        log.append((k,'B' if goB else 'A',len(chosen),sum(s for _,s in chosen),round(cost)))
    return total,log
if __name__=="__main__":
    for bias in (0,2,3,4,6):
        t,log=sim(bias=bias); print("bias",bias,"sweep %.0f us"%t, "B steps", sum(1 for l in log if l[1]=='B'), "rounds>1", [l for l in log if l[4]>50])
    t,log=sim(bias=0)
    print(log)
