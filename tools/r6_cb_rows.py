#!/usr/bin/env python3
"""One sweep per call on the callback path (urf_classify_pc2 through the Python binding), in firing order and row-major.
python tools/r6_cb_rows.py"""
import os, sys, time, numpy as np
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,"tests"))
import urban_road_filter_amd as u, oracles as O
n=64*2048; p=O.cfg_params("cfg2")
x,y,z=O.cfg_cloud("cfg2",5)
rows=tuple(np.ascontiguousarray(a.reshape(-1,64).T.reshape(-1)) for a in (x,y,z))
for name,c in (("firing",(x,y,z)),("rows",rows)):
    with u.Context(n,4,params=p) as ctx:
        for _ in range(5): ctx.classify_xyz(*c)
        t0=time.perf_counter()
        for _ in range(50): ctx.classify_xyz(*c)
        print(name,"sync classify_xyz: %.3f ms per sweep" % ((time.perf_counter()-t0)*1e3/50))
