import sys, time, numpy as np, ctypes as C, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle import oracle_lib as o
L,HQ,H,D=8192,4,1,128
rng=np.random.default_rng(0)
bf=lambda a:(a.astype(np.float32).view(np.uint32)>>16).astype(np.uint16)
q=bf(rng.standard_normal((HQ,L,D))); k=bf(rng.standard_normal((H,L,D))); v=bf(rng.standard_normal((H,L,D)))
y=np.zeros((HQ,L,D),np.uint16); cs=np.zeros((H,L),np.float32); ob=np.zeros((H,L),np.float32)
for th in (16, 32, 64, 128, 256):
    o.set_threads(th)
    t=time.time()
    o.call("cc_prefill_attn", o.ptr(q),o.ptr(k),o.ptr(v),HQ,H,L,D,1,1/np.sqrt(D),o.ptr(y),o.ptr(cs),o.ptr(ob),16,None,0,None)
    print(th, round(time.time()-t,2), flush=True)
