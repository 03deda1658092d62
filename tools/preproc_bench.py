#!/usr/bin/env python
"""Time the pre-processing kernels on 32 resident 4K frames: grayscale (RGB24, RGBA32, YUYV422) and the
Otsu histogram, with HIP events via the plan's timing mode.  Prints ms per launch and the HBM rate on algorithmic bytes
(grayscale: bpp B/px read + 1 B/px written; Otsu: 1 B/px read)."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from compv_amd import capi

W, H, F = 3840, 2160, 32
dev = torch.device("cuda", 0)
ctx = capi.Context(0)
plan = capi.Plan(ctx, W, H, W, F, 1.0)
g = torch.Generator(device="cpu").manual_seed(1)
res = {}
for name, fmt in (("RGB24", capi.FMT_RGB24), ("RGBA32", capi.FMT_RGBA32), ("YUYV422", capi.FMT_YUYV422)):
    bpp = capi.FMT_BYTES[fmt]
    d_in = torch.randint(0, 256, (F, H, W * bpp), dtype=torch.uint8, generator=g).to(dev)
    d_gray = torch.empty((F, H, W), dtype=torch.uint8, device=dev)
    for _ in range(3):
        plan.grayscale(d_in.data_ptr(), fmt, d_gray.data_ptr())
    torch.cuda.synchronize()
    plan.set_timing(1)
    ms = []
    for _ in range(10):
        plan.grayscale(d_in.data_ptr(), fmt, d_gray.data_ptr())
        torch.cuda.synchronize()
        ms += [m for n, m in plan.get_timing() if n == "gray_kernel"]
    plan.set_timing(0)
    t = float(np.median(ms))
    res["gray_" + name] = {"ms": round(t, 4), "GB/s": round(F * W * H * (bpp + 1) / (t * 1e-3) / 1e9, 1)}
    del d_in
d_t = torch.zeros(F, dtype=torch.int32, device=dev)
frames = torch.randint(0, 256, (F, H, W), dtype=torch.uint8, generator=g)
frames[: F // 2] = (frames[: F // 2] // 64) * 64 + 7       # half of the batch: 4 grey levels only (worst case for a naive LDS histogram)
d_gray = frames.to(dev)
for _ in range(3):
    plan.otsu(d_gray.data_ptr(), d_t.data_ptr())
torch.cuda.synchronize()
plan.set_timing(1)
ms = []
for _ in range(10):
    plan.otsu(d_gray.data_ptr(), d_t.data_ptr())
    torch.cuda.synchronize()
    ms += [m for n, m in plan.get_timing() if n == "otsu_kernels"]
plan.set_timing(0)
t = float(np.median(ms))
res["otsu (memset + hist256 + scan)"] = {"ms": round(t, 4), "GB/s": round(F * W * H / (t * 1e-3) / 1e9, 1)}
for K, sigma in ((5, 1.0), (7, 2.0), (15, 4.0)):
    kern = capi.gauss_kernel_fixedpoint(K, sigma)
    d_out = torch.empty_like(d_gray)
    for _ in range(3):
        plan.convlt_fixedpoint(d_gray.data_ptr(), kern, kern, d_out.data_ptr())
    torch.cuda.synchronize()
    plan.set_timing(1)
    ms = []
    for _ in range(10):
        plan.convlt_fixedpoint(d_gray.data_ptr(), kern, kern, d_out.data_ptr())
        torch.cuda.synchronize()
        ms += [m for n, m in plan.get_timing() if n == "convlt_fxp_kernels"]
    plan.set_timing(0)
    t = float(np.median(ms))
    # algorithmic bytes: 1 B/px read + 1 B/px written (the u8 intermediate of the two passes is implementation traffic)
    res["gauss_fxp_K%d (fused hz+vt kernel)" % K] = {"ms": round(t, 4), "GB/s": round(F * W * H * 2 / (t * 1e-3) / 1e9, 1)}
print(json.dumps(res))
plan.close(); ctx.close()
