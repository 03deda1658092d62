"""How much of the KHT linking stage could run concurrently if the walks of different 8-connected components were given to different workers
(the two-phase design of the round-3 verdict: label components on the GPU, walk them in parallel)?  Components never interact -- a walk cannot leave
its component -- so the linker run on ONE component alone produces exactly that component's strings, and its time is the sequential floor of any such
design.  CPU only (compvhip_houghkht_link_u8 needs no device).   python tools/kht_components.py [W H]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from scipy import ndimage
from compv_amd import capi
from oracle_bindings import Oracle, synth_frame

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
orc = Oracle()
rc, edges = orc.canny(synth_frame(W, H, 12345), 59.0, 119.0)
lab, n = ndimage.label(edges != 0, structure=np.ones((3, 3), int))
sizes = np.bincount(lab.ravel())[1:]
big = int(np.argmax(sizes)) + 1
giant = np.where(lab == big, 255, 0).astype(np.uint8)
rest = np.where((lab != big) & (lab != 0), 255, 0).astype(np.uint8)


import ctypes as C
lib = capi.load()
sz = C.c_size_t
lib.compvhip_houghkht_link_u8.argtypes = [C.c_void_p, sz, sz, sz, sz, C.c_void_p, sz, C.c_void_p, C.c_void_p, sz, C.c_void_p]
cap = int((edges != 0).sum())
xy = np.zeros((cap, 2), np.int32); ends = np.zeros(cap, np.uint32)


def best(e, reps=9):
    """the C entry point alone (buffers preallocated): pack + seed scan + walks"""
    e = np.ascontiguousarray(e)
    npts, nstr = sz(0), sz(0)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        rc = lib.compvhip_houghkht_link_u8(e.ctypes.data, W, H, W, 10, xy.ctypes.data, cap, C.byref(npts), ends.ctypes.data, cap, C.byref(nstr))
        ts.append(time.perf_counter() - t0)
        assert rc == 0
    return min(ts) * 1e3, (xy[:npts.value].copy(), ends[:nstr.value].copy())


t_all, (p_all, s_all) = best(edges)
t_giant, (p_g, s_g) = best(giant)
t_rest, (p_r, s_r) = best(rest)
t_empty, _ = best(np.zeros_like(edges))
print("%dx%d Canny(59,119): %d edge pixels in %d 8-connected components; the largest holds %d (%.1f %%)" % (W, H, int(sizes.sum()), n, int(sizes.max()), 100.0 * sizes.max() / sizes.sum()))
print("linker (pack + scan + walks, ms on this host): whole map %.2f | giant component alone %.2f | all other components %.2f | empty map (pack + scan only) %.2f" % (t_all, t_giant, t_rest, t_empty))
print("strings: whole %d = giant %d + rest %d; points %d = %d + %d" % (len(s_all), len(s_g), len(s_r), len(p_all), len(p_g), len(p_r)))
print("sequential floor of a component-parallel linker: %.2f of %.2f ms = %.0f %% (best case gain %.0f %%)" % (t_giant, t_all, 100 * t_giant / t_all, 100 * (1 - t_giant / t_all)))
