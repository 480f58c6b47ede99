import sys, os, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import torch
from cold_compress_amd import _abi
import cold_compress_amd.attention_utils as au
import test_gpu_recovery as T
DEV="cuda"
H,HQ,S,D=8,32,int(os.environ.get('S','4096')),128
kv,Tn=T._mk(H,S)
gen=torch.Generator(device=DEV).manual_seed(6)
q=torch.randn(1,HQ,1,D,device=DEV,generator=gen).to(torch.bfloat16)
k1=torch.randn(1,H,1,D,device=DEV,generator=gen).to(torch.bfloat16)
p=torch.tensor([Tn],dtype=torch.int32,device=DEV)
kv.decode_step(q,k1,k1,p); torch.cuda.synchronize()
scratch=torch.zeros(64,dtype=torch.int32,device=DEV)
side=torch.cuda.Stream()
fn=_abi.lib()["cc_debug_occupy"]
for nwg,lds,us in [(192,150*1024,1500000),(224,150*1024,1500000),(232,150*1024,1500000),(236,150*1024,1500000)]:
    torch.cuda.synchronize(); t0=time.perf_counter()
    rc=fn(nwg,lds,us,C.c_void_p(scratch.data_ptr()),C.c_void_p(side.cuda_stream)); assert rc==0
    side.synchronize(); t_hog=time.perf_counter()-t0
    p+=1
    torch.cuda.synchronize(); t0=time.perf_counter()
    rc=fn(nwg,lds,us,C.c_void_p(scratch.data_ptr()),C.c_void_p(side.cuda_stream)); assert rc==0
    time.sleep(0.01)
    t1=time.perf_counter()
    kv.decode_step(q,k1,k1,p)
    torch.cuda.current_stream().synchronize(); t_step=time.perf_counter()-t1
    side.synchronize(); t_all=time.perf_counter()-t0
    st=au.single_launch_status(torch.device(DEV))
    print(dict(nwg=nwg,lds=lds,us=us,t_hog=round(t_hog,3),t_step_with_hog=round(t_step,3),t_all=round(t_all,3),status=st,commit=kv.step_commit.tolist()),flush=True)
    if st: au.reset_single_launch_status(torch.device(DEV))
