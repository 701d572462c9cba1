import numpy as np, sys
sys.path.insert(0,'/root/repo')
from hope_amd import tables as T
t=T.all_tables()
tab=np.maximum.accumulate(t['dist_star'][::10],axis=2)   # [120,42,10]
pmax=tab.max(axis=(1,2)); hb=t['hull_base']
d=np.load('/root/repo/gpurun_out/r06/scans_steady.npz')
lid=d['lidar']; nob=d['nobst']
NACT=42
def sim(x, order, MG=4):
    mstep=np.full(NACT,10); groups=0; walks=0; walk_iters=0; fails=0
    for g0 in range(0,len(order),MG):
        grp=order[g0:g0+MG]; mprobe=mstep.copy(); groups+=1
        v=[np.where(mprobe>0, tab[i, np.arange(NACT), np.maximum(mprobe-1,0)], 0.0) for i in grp]
        for gi,i in enumerate(grp):
            xv=x[i]; fail=(mstep>0)&(v[gi]>xv)
            if fail.any():
                walks+=1; fails+=int(fail.sum())
                c=np.where(mstep==mprobe, mstep-1, mstep); itmax=0; newm=mstep.copy()
                for a in np.nonzero(fail)[0]:
                    ca=c[a]; it=0
                    while ca>0:
                        it+=1; stop=False
                        for r in range(4):
                            if ca==0: break
                            if not (tab[i,a,ca-1]>xv): stop=True; break
                            ca-=1
                        if stop: break
                    newm[a]=ca; itmax=max(itmax,it)
                mstep=newm; walk_iters+=itmax
    return mstep,groups,walks,walk_iters
rng=np.random.default_rng(0)
small=np.nonzero(nob<=32)[0]
idx=rng.choice(small,2500,replace=False)
R={k:[] for k in ['nact','g','w','wi','ag','aw','awi','mg','mw','mwi']}
for s in idx:
    x=np.clip(lid[s],0,10)+hb
    act=np.nonzero(x-1e-9<pmax)[0]
    m,g,w,wi=sim(x,list(act)); R['nact'].append(len(act)); R['g'].append(g); R['w'].append(w); R['wi'].append(wi)
    o=sorted(act,key=lambda i:lid[s][i]); m2,g,w,wi=sim(x,o); assert (m2==m).all(); R['ag'].append(g); R['aw'].append(w); R['awi'].append(wi)
    # min-first then ballot order
    if len(act):
        j=act[np.argmin(lid[s][act])]; o=[j]+[i for i in act if i!=j]
    else: o=[]
    m3,g,w,wi=sim(x,o); assert (m3==m).all(); R['mg'].append(g); R['mw'].append(w); R['mwi'].append(wi)
for k,v in R.items(): print(k,np.mean(v).round(2))
