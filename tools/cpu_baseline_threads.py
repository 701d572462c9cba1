#!/usr/bin/env python3
"""Thread scaling of the CPU oracle's batch step (the C call only) on this host: NS scenes (env NS, default 16384) x 3 steps with
1, 8, 32, 64, 128 OpenMP threads -> profiles/rNN_cpu_baseline_threads.txt"""
import numpy as np, sys, time, os
sys.path.insert(0,os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
from hope_amd import tables as T
from hope_amd.scenes import SceneSource, pack_scenes
from oracle import oracle as O
import ctypes as C
t=T.all_tables()
n,mo=int(os.environ.get('NS','16384')),128
src=SceneSource(seed=42)
uniq=[src.draw() for _ in range(1024)]; scenes=[uniq[i%1024] for i in range(n)]
start,dest,bbox,verts,nob,nvert=pack_scenes(scenes,mo)
O.set_tables(hull_base=t['hull_base'], beam_a=t['beam_ab'][:,0], beam_b=t['beam_ab'][:,1], dist_star=t['dist_star'], omp=True)
O.set_tables(hull_base=t['hull_base'], beam_a=t['beam_ab'][:,0], beam_b=t['beam_ab'][:,1], dist_star=t['dist_star'], omp=False)
rng=np.random.default_rng(0)
for nt in [int(x) for x in os.environ.get("NTS","1,8,32,64,128").split(",")]:
    O.lib(True).orc_set_num_threads(nt)
    orc=O.BatchOracle(n,mo,omp=True)
    orc.set_scenes(np.arange(n),start,dest,bbox,verts,nvert,nob)
    orc.reset_obs()
    acts=[rng.uniform(-1,1,(n,2)) for _ in range(3)]
    tc=0; t0=time.perf_counter()
    for a in acts:
        o=orc.out
        t1=time.perf_counter()
        orc.L.orc_batch_step(C.c_int(n), C.c_int(mo), O._p(orc.n_obst), O._p(orc.verts), O._p(orc.nvert), O._p(orc.start), O._p(orc.dest), O._p(orc.bbox), O._p(orc.pose), O._p(orc.t), O._p(orc.accum), O._p(O._f64(a)), C.c_int(1), O._p(o['lidar']), O._p(o['mask']), O._p(o['target']), O._p(o['reward_info']), O._p(o['reward']), O._p(o['status']), O._p(o['rs_found']), O._p(o['rs_ctypes']), O._p(o['rs_lengths']), O._p(o['substeps']))
        tc+=time.perf_counter()-t1
    print(nt,'threads: C only', round(n*3/tc),'steps/s')
