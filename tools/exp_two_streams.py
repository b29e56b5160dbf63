import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch, numpy as np
import oracles as O, urban_road_filter_amd as u
import bench
S=1024; N=bench.N_PTS
X,Y,Z=bench.gen_batch(S,1)
dev=torch.device("cuda",0)
dx,dy,dz=[torch.from_numpy(a).to(dev) for a in (X,Y,Z)]
dl=torch.empty((S,N),dtype=torch.uint8,device=dev)
p=O.cfg_params("cfg2")
for nctx in (1,2,4):
    per=S//nctx
    ctxs=[u.Context(N,per,device=0,params=p) for _ in range(nctx)]
    streams=[torch.cuda.Stream() for _ in range(nctx)]
    for c,st in zip(ctxs,streams): c.set_stream(st.cuda_stream)
    def step():
        for k,c in enumerate(ctxs):
            lo=k*per
            c.classify_batch_soa(dx[lo:lo+per],dy[lo:lo+per],dz[lo:lo+per],N,per,dl[lo:lo+per],None)
    for _ in range(3): step()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); el=time.perf_counter()-t
    print("ctxs",nctx,"ms/step",el/10*1e3,"scans/s",S*10/el, flush=True)
    for c in ctxs: c.close()
