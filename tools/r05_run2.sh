#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests2.txt 2>&1; tail -3 $O/tests2.txt
bash tools/exp_env_ab.sh 3 "v2:" "v1:HOPE_AMD_LIB=$PWD/hope_amd/libhope_env_v1.so" > $O/ab_screen2.txt 2>&1; cat $O/ab_screen2.txt
HOPE_RS_TIMING=1 timeout 400 python tools/rs_timing.py > $O/rs_validate_cycles2.txt 2>/dev/null; cat $O/rs_validate_cycles2.txt
timeout 600 python tools/rs_filter_stats.py --check > $O/rs_filter_stats2.txt 2>/dev/null; cat $O/rs_filter_stats2.txt
python - <<'P'
import time, numpy as np
from hope_amd import _lib as L
lib=L.load_library()
n,mo=8192,128
start, dest, bbox = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 4))
verts = np.ones((n, mo, 4, 2)); nob=np.zeros(n,np.int32)
for th in (1,8,32,64,128,256):
    for lv in (0,1,2):
        lib.hope_scenegen_generate(lv,-1,n//3,1,0,mo,start.ctypes.data,dest.ctypes.data,bbox.ctypes.data,verts.ctypes.data,nob.ctypes.data,None,th)
    t=time.perf_counter()
    for r in range(5):
        for lv in (0,1,2):
            lib.hope_scenegen_generate(lv,-1,n//3,r,0,mo,start.ctypes.data,dest.ctypes.data,bbox.ctypes.data,verts.ctypes.data,nob.ctypes.data,None,th)
    dt=(time.perf_counter()-t)/5
    print(th,'threads', round(3*(n//3)/dt), 'lots/s (pool refill: 3 calls of 2730 lots into preallocated arrays)')
P
