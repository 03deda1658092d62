"""Traffic calibration workload for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes: runs the batched Sobel detector
(pass 1 = pure streaming READ of a known byte count with exactly the load pattern of the Canny tile kernel: 8-byte +
two 4-byte loads per lane; pass 2 = the same read + a known 1 B/px WRITE) and the full pipeline on 32 x 4K frames."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from compv_amd import capi
from oracle_bindings import synth_frame
W, H, F = 3840, 2160, 32
frames = np.stack([synth_frame(W, H, 12345 + f) for f in range(F)])
dev = torch.device("cuda:0")
d_in = torch.from_numpy(frames).to(dev); d_out = torch.empty_like(d_in)
d_lines = torch.zeros((F, 1 << 16, 5), dtype=torch.int32, device=dev); d_counts = torch.zeros(F, dtype=torch.int32, device=dev)
ctx = capi.Context(0); plan = capi.Plan(ctx, W, H, W, F, 1.0)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    plan.edge_dete(d_in.data_ptr(), capi.OP_SOBEL, d_out.data_ptr(), st)
    plan.pipeline(d_in.data_ptr(), 59.0, 119.0, 100, 0, d_out.data_ptr(), d_lines.data_ptr(), 1 << 16, d_counts.data_ptr(), st)
torch.cuda.synchronize()
print("known bytes per launch: input", F * W * H, "rows halo factor 66/64 (sobel), 68/64 (canny); output", F * W * H)
