"""Developer: a cost model for loop shapes of K2g (octree_group.hip) on the REAL step sequences of a sample of rays.

Walks the oracle's octree for 4 096 burst rays on the host (the reference's push / pop order, "Octree - alt.cs":159-306; pruning
approximated by the ray's final t), records each ray's sequence of F steps (an interior node: eight children tested) and L steps (eight
list entries pre-culled), and replays those sequences through a wave of eight groups under different loop shapes: R ray slots per group,
and the phases a pass of the loop executes ("FL" = one F and one L round per pass, "FFLL" = two of each, ...).  A phase that any group
uses costs its full instruction count (measured with the counting build, tools/k2g_stats.py: POP + F ~125, POP share + L ~110, loop
control + exact share ~95 per pass).  Output: instructions per ray, passes per ray, groups active per F / L execution.
Round 4, hall, octree 8/16: R=1 FL 1965 (the kernel as first built: 1 490 measured on 36.8 steps per ray where this walk has 49.7),
R=1 FFFLLL 1674, R=2 FFLL 1264, R=3 FFFLLL 1077, R=4 FFFLLL 1014 -- before the cost of keeping R slots' state in LDS.   CPU only.
"""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import hare_amd as H
from oracle import pyoracle as po
m=H.scenes.SCENES['hall']()
ot=po.Topology(m.verts,m.nverts)
og=po.Octree([ot],8,16)
boxes, fc, st, cn, items = og.export()
N=1<<20
start=300000; R=4096
rays=H.scenes.burst_rays(N, m.size)[start:start+R]
ev,_=og.shoot(rays)
def dmax(a,b): return a if (a>b or a!=a) else b
def dmin(a,b): return a if (a<b or a!=a) else b
seqs=[]
for r,e in zip(rays,ev):
    o=r[:3]; d=r[3:]
    inv=[(1.0/d[a] if abs(d[a])>1e-16 else 1e16) for a in range(3)]
    mask=((0 if d[0]>=0 else 1)<<2)|((0 if d[1]>=0 else 1)<<1)|(0 if d[2]>=0 else 1)
    def slab(b):
        t0=[(b[a]-o[a])*inv[a] for a in range(3)]; t1=[(b[a+3]-o[a])*inv[a] for a in range(3)]
        for a in range(3):
            if inv[a]<0: t0[a],t1[a]=t1[a],t0[a]
        return dmax(dmax(t0[0],t0[1]),t0[2]), dmin(dmin(t1[0],t1[1]),t1[2])
    tmn,tmx=slab(boxes[0])
    seq=[]
    if not (tmx<tmn or tmx<0):
        stack=[(0,tmn,tmx)]
        # approximate pruning: use the final t as closestT once we've passed a leaf containing the hit poly (cheap proxy): prune nodes with ca >= t_final after hit found
        tfin=e['t'] if e['hit'] else None; found=False
        while stack:
            ni,a,b=stack.pop()
            if found and tfin is not None and tfin<=a: continue
            if fc[ni]<0:
                seq += ['L']*((cn[ni]+7)//8)
                if tfin is not None and e['poly_id'] in items[st[ni]:st[ni]+cn[ni]]: found=True
                continue
            seq.append('F')
            for k in range(8):
                ci=fc[ni]+(k^mask)
                ca,cb=slab(boxes[ci])
                if cb<ca or cb<0 or ca>b or cb<a: continue
                ca=dmax(ca,a); cb=dmin(cb,b)
                if cb<ca or cb<0: continue
                if fc[ci]<0 and cn[ci]==0: continue
                stack.append((ci,ca,cb))
    seqs.append(seq)

print('mean steps',np.mean([len(s) for s in seqs]),'F frac',np.mean([s.count('F')/max(len(s),1) for s in seqs]))

seqs=[s for s in seqs if s]
COST={'F':75+50,'L':85+25}; CTL=95
def sim(pattern, R, refill_min=2, nwave_rays=len(seqs)):
    G=8
    it=iter(seqs)
    slots=[[None]*R for _ in range(G)]   # each slot: [seq, pos]
    done=0; cost=0; steps=0; execs={'F':[0,0],'L':[0,0]}; iters=0
    pending=True
    def refill():
        nonlocal pending
        for g in range(G):
            for r in range(R):
                if slots[g][r] is None and pending:
                    try: slots[g][r]=[next(it),0]
                    except StopIteration: pending=False
    refill()
    while True:
        if all(s is None for g in slots for s in g):
            if not pending: break
            refill(); continue
        iters+=1; cost+=CTL
        for ph in pattern:
            n=0
            for g in range(G):
                for r in range(R):
                    s=slots[g][r]
                    if s is not None and s[0][s[1]]==ph:
                        s[1]+=1; n+=1; steps+=1
                        if s[1]==len(s[0]): slots[g][r]=None; done+=1
                        break
            if n:
                cost+=COST[ph]; execs[ph][0]+=1; execs[ph][1]+=n
        idle=sum(1 for g in slots for s in g if s is None)
        if idle>=refill_min: refill()
    return cost/done, iters/done, execs['F'][1]/max(execs['F'][0],1), execs['L'][1]/max(execs['L'][0],1)
for R in (1,2,3,4):
    for pat in ('FL','FFLL','FLFL','FFFLLL','FLL','FFL'):
        c,i,fo,lo=sim(pat,R)
        print('R=%d %-7s cost/ray %.0f  iters/ray %.2f  F occ %.2f L occ %.2f'%(R,pat,c,i,fo,lo))
