#!/usr/bin/env python3
"""ONE resident sweep per call through the general kernels (front mode 0) and through the fused front end (mode 2; URF_FRONT_TPB = tiles per
block of k_front): why single sweeps keep the general kernels.  python tools/r6_single_sweep.py"""
import os, sys, time, numpy as np
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,"tests"))
import urban_road_filter_amd as u, oracles as O
from hipmem import DevBuf
n=64*2048
p=O.cfg_params("cfg2")
x,y,z=O.cfg_cloud("cfg2",5)
dx,dy,dz=DevBuf.from_numpy(x),DevBuf.from_numpy(y),DevBuf.from_numpy(z); dl=DevBuf(n)
lb,_,_=O.run_b(x,y,z,p)
for mode in (0,2):
    ctx=u.Context(n,1,params=p); ctx.set_front_mode(mode)
    for _ in range(30): ctx.classify_batch_soa(dx,dy,dz,n,1,dl,None)
    ctx.synchronize()
    ok=np.array_equal(dl.to_numpy(np.uint8),lb)
    t0=time.perf_counter()
    for _ in range(500): ctx.classify_batch_soa(dx,dy,dz,n,1,dl,None)
    ctx.synchronize(); ms=(time.perf_counter()-t0)*1e3/500
    ctx.enable_kernel_timing(True); ctx.kernel_timing(); ctx.enable_kernel_timing(True)
    for _ in range(50): ctx.classify_batch_soa(dx,dy,dz,n,1,dl,None)
    k,c=ctx.kernel_timing()
    print("mode",mode,"tpb",os.environ.get("URF_FRONT_TPB"),"fused",ctx.front_scans(),"parity",ok,"%.4f ms"%ms," ".join("%s=%.3f"%(a[2:],v/c) for a,v in k.items()))
    ctx.close()
