import os, sys
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from compv_amd import capi
from oracle_bindings import synth_frame
ctx=capi.Context(0)
base=ctx.live_allocations()
img=synth_frame(640,480,1)
e=ctx.canny(img,59.0,119.0); ctx.houghsht(e,1.0,60); ctx.houghkht(e,1.0,1.0,10)
after_host=ctx.live_allocations()
dev=torch.device("cuda:0")
for (W,H,F) in ((640,480,19),(4104,72,2),(1280,720,3)):
    frames=np.stack([synth_frame(W,H,10+f) for f in range(F)])
    d_in=torch.from_numpy(frames).to(dev); d_e=torch.empty_like(d_in)
    d_l=torch.zeros((F,8192,5),dtype=torch.int32,device=dev); d_c=torch.zeros(F,dtype=torch.int32,device=dev)
    before=ctx.live_allocations()
    plan=capi.Plan(ctx,W,H,W,F,1.0)
    plan.pipeline(d_in.data_ptr(),59.0,119.0,30,0,d_e.data_ptr(),d_l.data_ptr(),8192,d_c.data_ptr())
    t=plan.pipeline_async(d_in.data_ptr(),59.0,119.0,30,0,d_e.data_ptr(),d_l.data_ptr(),8192,d_c.data_ptr()); plan.wait(t)
    plan.canny(d_in.data_ptr(),708.0,1428.0,d_e.data_ptr(),ksize=5)
    plan.houghkht(d_e.data_ptr(),1.0,1.0,20,threads=12)
    torch.cuda.synchronize()
    mid=ctx.live_allocations()
    plan.close()
    print((W,H,F),"live before plan",before,"with plan",mid,"after close",ctx.live_allocations())
    assert ctx.live_allocations()==before
print("host-path allocations (cached in the context):", after_host-base)
ctx.close()
print("LEAK CHECK OK")
