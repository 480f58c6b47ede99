import sys, os, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from cold_compress_amd import _abi
DEV="cuda"
scratch=torch.zeros(64,dtype=torch.int32,device=DEV)
side=torch.cuda.Stream()
fn=_abi.lib()["cc_debug_occupy"]
main=torch.cuda.current_stream()
x=torch.zeros(8,device=DEV); torch.cuda.synchronize()
for hog_n,hog_lds in [(128,150),(192,150),(240,150),(128,100),(256,150)]:
  for probe_n,probe_lds in [(256,85),(512,45),(64,85)]:
    torch.cuda.synchronize(); t0=time.perf_counter()
    assert fn(hog_n,hog_lds*1024,400000,C.c_void_p(scratch.data_ptr()),C.c_void_p(side.cuda_stream))==0
    time.sleep(0.02)
    t1=time.perf_counter()
    assert fn(probe_n,probe_lds*1024,10,C.c_void_p(scratch.data_ptr()),C.c_void_p(main.cuda_stream))==0
    main.synchronize(); t_probe=time.perf_counter()-t1
    side.synchronize()
    print(dict(hog=(hog_n,hog_lds),probe=(probe_n,probe_lds),t_probe=round(t_probe,4)),flush=True)
