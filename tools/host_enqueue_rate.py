import os, sys, time
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench
from compv_amd import capi
for (W,H) in ((1920,1080),(3840,2160)):
    F=32; dev=torch.device("cuda:0")
    synth=bench.FrameSynth(torch, dev, W, H)
    blocks=[synth.batch([12345+32*b+f for f in range(F)]) for b in range(2)]
    ctx=capi.Context(0)
    lanes=[]
    for i in range(2):
        lanes.append({"plan":capi.Plan(ctx,W,H,W,F,1.0),"e":torch.empty_like(blocks[0]),"l":torch.zeros((F,1<<16,5),dtype=torch.int32,device=dev),"c":torch.zeros(F,dtype=torch.int32,device=dev),"s":torch.cuda.Stream(device=dev)})
    def enq(q,b): return q["plan"].pipeline_async(blocks[b].data_ptr(),59.0,119.0,100,0,q["e"].data_ptr(),q["l"].data_ptr(),1<<16,q["c"].data_ptr(),q["s"].cuda_stream)
    for k in range(8):
        q=lanes[k%2]; t=enq(q,k%2); q["plan"].wait(t)
    torch.cuda.synchronize()
    # host time of an enqueue when the GPU is idle-ish vs throughput
    N=200; pend=[]; host=0.0
    t0=time.perf_counter()
    for k in range(N):
        q=lanes[k%2]
        h0=time.perf_counter(); t=enq(q,k%2); host+=time.perf_counter()-h0
        pend.append((q,t))
        if len(pend)>4:
            q0,t0_=pend.pop(0); q0["plan"].wait(t0_)
    for q0,t0_ in pend: q0["plan"].wait(t0_)
    torch.cuda.synchronize()
    wall=time.perf_counter()-t0
    print("%dx%d: wall per step %.4f ms, host enqueue time per step %.4f ms" % (W,H,wall/N*1e3, host/N*1e3))
    for q in lanes: q["plan"].close()
    ctx.close()
