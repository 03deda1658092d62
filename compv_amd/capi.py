"""ctypes binding of the C ABI in include/compv_hip.h (compv_amd/lib/libcompv_hip.so).

This is plumbing for tests and bench.py: the product is the shared library itself, which a CompV build binds
directly from C++.  There is NO CPU fallback: importing works without a GPU (so that the
symbol-export test can run), but creating a context without a GPU, or loading without the built library, fails loudly.
"""
import ctypes as C
import importlib.util
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcompv_hip.so")

OK = 0
E_NOT_IMPLEMENTED = -1
E_NOT_INITIALIZED = -2
E_INVALID_STATE = -3
E_INVALID_PARAMETER = -4
E_OUT_OF_MEMORY = -5
E_OUT_OF_BOUND = -6
E_HIP = -7

OP_SOBEL, OP_SCHARR, OP_PREWITT = 0, 2, 3
THRESHOLD_COMPARE_TO_GRADIENT, THRESHOLD_PERCENT_OF_MEAN, THRESHOLD_OTSU = 0, 1, 2
(FMT_RGBA32, FMT_ARGB32, FMT_BGRA32, FMT_RGB24, FMT_BGR24, FMT_RGB565LE, FMT_RGB565BE, FMT_BGR565LE, FMT_BGR565BE,
 FMT_YUYV422, FMT_UYVY422, FMT_Y) = range(12)
FMT_BYTES = [4, 4, 4, 3, 3, 2, 2, 2, 2, 2, 2, 1]

# every symbol include/compv_hip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "compvhip_device_count", "compvhip_ctx_create", "compvhip_ctx_destroy", "compvhip_last_error",
    "compvhip_live_allocations", "compvhip_edge_dete_u8", "compvhip_canny_u8", "compvhip_houghsht_u8",
    "compvhip_houghsht_dims", "compvhip_houghsht_vote_grid", "compvhip_plan_create", "compvhip_plan_destroy", "compvhip_plan_canny",
    "compvhip_plan_houghsht", "compvhip_plan_pipeline", "compvhip_plan_acc", "compvhip_plan_edge_counts",
    "compvhip_plan_set_timing", "compvhip_plan_get_timing", "compvhip_plan_acc_export", "compvhip_plan_edge_dete",
    "compvhip_houghkht_u8", "compvhip_grayscale_u8", "compvhip_otsu_u8", "compvhip_plan_grayscale", "compvhip_plan_otsu",
    "compvhip_gauss_kernel_fixedpoint", "compvhip_convlt1_fixedpoint_u8", "compvhip_plan_convlt1_fixedpoint", "compvhip_plan_to_cartesian",
    "compvhip_plan_pipeline_async", "compvhip_plan_wait", "compvhip_houghsht_to_cartesian", "compvhip_houghkht_to_cartesian",
    "compvhip_houghkht_kernels_u8", "compvhip_houghkht_stage_ms", "compvhip_convlt1_8u16s16s", "compvhip_convlt1_16s16s16s",
    "compvhip_plan_pipeline_ex", "compvhip_plan_houghkht", "compvhip_plan_houghkht_stage_ms", "compvhip_houghkht_link_u8",
    "compvhip_houghkht_dims", "compvhip_host_cpu_budget",
]


class Line(C.Structure):
    _fields_ = [("rho", C.c_float), ("theta", C.c_float), ("strength", C.c_int32), ("row", C.c_int32), ("col", C.c_int32)]


LINE_DTYPE = np.dtype([("rho", "<f4"), ("theta", "<f4"), ("strength", "<i4"), ("row", "<i4"), ("col", "<i4")])


class PipelineOpts(C.Structure):
    """compvhip_pipeline_opts (include/compv_hip.h)"""
    _fields_ = [("tLow", C.c_float), ("tHigh", C.c_float), ("threshold", C.c_int), ("maxLines", C.c_int), ("ksize", C.c_int),
                ("thresholdType", C.c_int), ("pixfmt", C.c_int), ("d_gray", C.c_void_p), ("d_otsu", C.c_void_p), ("d_cart", C.c_void_p)]


class CompvHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("compvhip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64.so (same soname as /opt/rocm's).  Device pointers are only
    interchangeable inside ONE HIP runtime, so when torch is installed its copy is loaded first and the loader
    resolves libcompv_hip.so's DT_NEEDED libamdhip64.so.7 to it (no torch import needed, no dependency on torch)."""
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec and spec.submodule_search_locations:
        p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(p):
            C.CDLL(p, mode=C.RTLD_GLOBAL)


def load():
    """Load the HIP library; raises if it was not built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("compv_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the HIP extension is mandatory; there is no CPU fallback)" % LIB_PATH)
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    sz, vp, i32 = C.c_size_t, C.c_void_p, C.c_int
    L.compvhip_device_count.restype = i32
    L.compvhip_ctx_create.argtypes = [C.POINTER(vp), i32]
    L.compvhip_ctx_destroy.argtypes = [vp]
    L.compvhip_ctx_destroy.restype = None
    L.compvhip_last_error.argtypes = [vp]
    L.compvhip_last_error.restype = C.c_char_p
    L.compvhip_live_allocations.argtypes = [vp]
    L.compvhip_live_allocations.restype = C.c_long
    L.compvhip_edge_dete_u8.argtypes = [vp, vp, sz, sz, sz, i32, vp, sz]
    L.compvhip_canny_u8.argtypes = [vp, vp, sz, sz, sz, C.c_float, C.c_float, i32, i32, vp, sz]
    L.compvhip_houghsht_u8.argtypes = [vp, vp, sz, sz, sz, C.c_float, C.c_float, i32, i32, vp, sz, C.POINTER(sz), vp, sz]
    L.compvhip_houghkht_kernels_u8.argtypes = [vp, vp, sz, sz, sz, C.c_double, sz, vp, sz, C.POINTER(sz), C.POINTER(C.c_double)]
    L.compvhip_houghkht_stage_ms.argtypes = [vp, vp]
    L.compvhip_houghkht_u8.argtypes = [vp, vp, sz, sz, sz, C.c_float, C.c_float, i32, i32, C.c_double, sz, C.c_double, vp, sz, C.POINTER(sz),
                                       C.POINTER(C.c_double)]
    L.compvhip_houghsht_dims.argtypes = [sz, sz, C.c_float, C.POINTER(sz), C.POINTER(sz), C.POINTER(C.c_float)]
    L.compvhip_houghsht_vote_grid.argtypes = [sz, sz, C.c_float, sz, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.compvhip_plan_create.argtypes = [vp, sz, sz, sz, sz, C.c_float, C.POINTER(vp)]
    L.compvhip_plan_destroy.argtypes = [vp]
    L.compvhip_plan_destroy.restype = None
    L.compvhip_plan_canny.argtypes = [vp, vp, C.c_float, C.c_float, i32, i32, vp, vp]
    L.compvhip_plan_houghsht.argtypes = [vp, vp, i32, i32, vp, sz, vp, vp]
    L.compvhip_plan_edge_dete.argtypes = [vp, vp, i32, vp, vp]
    L.compvhip_grayscale_u8.argtypes = [vp, vp, i32, sz, sz, sz, vp, sz]
    L.compvhip_otsu_u8.argtypes = [vp, vp, sz, sz, sz, C.POINTER(C.c_double)]
    L.compvhip_plan_grayscale.argtypes = [vp, vp, i32, vp, vp]
    L.compvhip_plan_otsu.argtypes = [vp, vp, vp, vp]
    L.compvhip_plan_to_cartesian.argtypes = [vp, vp, vp, sz, vp, vp]
    L.compvhip_gauss_kernel_fixedpoint.argtypes = [sz, C.c_float, vp]
    L.compvhip_convlt1_fixedpoint_u8.argtypes = [vp, vp, sz, sz, sz, vp, vp, sz, vp, sz]
    L.compvhip_plan_convlt1_fixedpoint.argtypes = [vp, vp, vp, vp, sz, vp, vp]
    L.compvhip_convlt1_8u16s16s.argtypes = [vp, vp, sz, sz, sz, vp, vp, sz, vp, sz]
    L.compvhip_convlt1_16s16s16s.argtypes = [vp, vp, sz, sz, sz, vp, vp, sz, vp, sz]
    L.compvhip_plan_pipeline.argtypes = [vp, vp, C.c_float, C.c_float, i32, i32, vp, vp, sz, vp, vp]
    L.compvhip_plan_pipeline_async.argtypes = [vp, vp, C.c_float, C.c_float, i32, i32, vp, vp, sz, vp, vp, C.POINTER(i32)]
    L.compvhip_plan_wait.argtypes = [vp, i32]
    L.compvhip_plan_houghkht.argtypes = [vp, vp, C.c_float, C.c_float, i32, i32, C.c_double, sz, C.c_double, vp, sz, vp, vp, i32]
    L.compvhip_plan_houghkht_stage_ms.argtypes = [vp, vp, C.POINTER(C.c_double), C.POINTER(i32)]
    L.compvhip_plan_pipeline_ex.argtypes = [vp, vp, C.POINTER(PipelineOpts), vp, vp, sz, vp, vp, C.POINTER(i32)]
    L.compvhip_plan_acc.argtypes = [vp, sz, C.POINTER(vp), C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]
    L.compvhip_plan_acc_export.argtypes = [vp, sz, vp, sz, vp]
    L.compvhip_plan_edge_counts.argtypes = [vp, C.POINTER(vp)]
    L.compvhip_plan_set_timing.argtypes = [vp, i32]
    L.compvhip_plan_get_timing.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_float), i32]
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One GPU context (compvhip_ctx)."""

    def __init__(self, device=-1):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.compvhip_ctx_create(C.byref(h), device)
        if rc != OK:
            raise CompvHipError(rc, "compvhip_ctx_create failed (no usable GPU? the HIP path is mandatory)")
        self.h = h

    def close(self):
        if self.h:
            self.lib.compvhip_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self):
        return self.lib.compvhip_last_error(self.h).decode()

    def live_allocations(self):
        return self.lib.compvhip_live_allocations(self.h)

    def _chk(self, rc):
        if rc != OK:
            raise CompvHipError(rc, self.last_error())

    # ---- host entry points (numpy in / numpy out) ----
    def edge_dete(self, img, op=OP_SOBEL):
        H, W = img.shape
        out = np.empty((H, W), np.uint8)
        self._chk(self.lib.compvhip_edge_dete_u8(self.h, _ptr(img), W, H, img.strides[0], op, _ptr(out), W))
        return out

    def grayscale(self, packed, pixfmt, W):
        """packed: (H, S*bpp) uint8 array, rows of S samples; returns the (H, W) luma plane (CompVImage::convertGrayscale)."""
        H = packed.shape[0]
        bpp = FMT_BYTES[pixfmt]
        S = packed.strides[0] // bpp
        out = np.empty((H, W), np.uint8)
        self._chk(self.lib.compvhip_grayscale_u8(self.h, _ptr(packed), pixfmt, W, H, S, _ptr(out), W))
        return out

    def convlt_fixedpoint(self, img, vt, hz):
        """CompVMathConvlt::convlt1FixedPoint (vt/hz: uint16 Q16 weights)."""
        H, W = img.shape
        vt = np.ascontiguousarray(vt, np.uint16); hz = np.ascontiguousarray(hz, np.uint16)
        out = np.empty((H, W), np.uint8)
        self._chk(self.lib.compvhip_convlt1_fixedpoint_u8(self.h, _ptr(img), W, H, img.strides[0], _ptr(vt), _ptr(hz), len(vt), _ptr(out), W))
        return out

    def convlt1_i16(self, img, vt, hz):
        """CompVMathConvlt::convlt1<u8|s16, s16, s16>: separable integer correlation, int16 out (img: uint8 or int16, C-contiguous rows)."""
        H, W = img.shape
        vt = np.ascontiguousarray(vt, np.int16); hz = np.ascontiguousarray(hz, np.int16)
        assert len(vt) == len(hz)
        out = np.zeros((H, W), np.int16)
        fn = self.lib.compvhip_convlt1_8u16s16s if img.dtype == np.uint8 else self.lib.compvhip_convlt1_16s16s16s
        assert img.dtype in (np.uint8, np.int16)
        self._chk(fn(self.h, _ptr(img), W, H, img.strides[0] // img.itemsize, _ptr(vt), _ptr(hz), len(vt), _ptr(out), W))
        return out

    def otsu(self, img):
        """CompVImage::thresholdOtsu: the Otsu level (a double holding an integer, like the reference)."""
        H, W = img.shape
        t = C.c_double(0)
        self._chk(self.lib.compvhip_otsu_u8(self.h, _ptr(img), W, H, img.strides[0], C.byref(t)))
        return t.value

    def canny(self, img, tLow, tHigh, ksize=3, threshold_type=THRESHOLD_COMPARE_TO_GRADIENT, out=None):
        H, W = img.shape
        if out is None:
            out = np.empty((H, W), np.uint8)
        self._chk(self.lib.compvhip_canny_u8(self.h, _ptr(img), W, H, img.strides[0], tLow, tHigh, ksize, threshold_type,
                                             _ptr(out), out.strides[0]))
        return out

    def houghsht_dims(self, W, H, theta_deg=1.0):
        R, T, st = C.c_size_t(), C.c_size_t(), C.c_float()
        rc = self.lib.compvhip_houghsht_dims(W, H, theta_deg, C.byref(R), C.byref(T), C.byref(st))
        self._chk(rc)
        return R.value, T.value, st.value

    def houghsht(self, edges, theta_deg=1.0, threshold=100, max_lines=0, rho=1.0, cap=1 << 16, want_acc=False):
        H, W = edges.shape
        lines = np.zeros(cap, LINE_DTYPE)
        n = C.c_size_t(0)
        acc = None
        accp, accs = None, 0
        if want_acc:
            R, T, _ = self.houghsht_dims(W, H, theta_deg)
            acc = np.zeros((R, T), np.int32)
            accp, accs = _ptr(acc), T
        rc = self.lib.compvhip_houghsht_u8(self.h, _ptr(edges), W, H, edges.strides[0], rho, theta_deg, threshold, max_lines,
                                           _ptr(lines), cap, C.byref(n), accp, accs)
        if rc == E_OUT_OF_BOUND and n.value > cap:
            return self.houghsht(edges, theta_deg, threshold, max_lines, rho, cap=n.value, want_acc=want_acc)
        self._chk(rc)
        lines = lines[:n.value]
        return (lines, acc) if want_acc else lines


    def houghkht(self, edges, rho=1.0, theta_deg=1.0, threshold=1, max_lines=0, min_dev=2.0, min_size=10, min_height=0.002, cap=1 << 14):
        """Returns (lines, GS); lines['row'] / ['col'] hold the rho / theta indices."""
        H, W = edges.shape
        lines = np.zeros(cap, LINE_DTYPE)
        n = C.c_size_t(0)
        gs = C.c_double(1.0)
        rc = self.lib.compvhip_houghkht_u8(self.h, _ptr(edges), W, H, edges.strides[0], rho, theta_deg, threshold, max_lines, min_dev, min_size,
                                           min_height, _ptr(lines), cap, C.byref(n), C.byref(gs))
        if rc == E_OUT_OF_BOUND and n.value > cap:
            return self.houghkht(edges, rho, theta_deg, threshold, max_lines, min_dev, min_size, min_height, cap=n.value)
        self._chk(rc)
        return lines[:n.value], gs.value


    def houghkht_kernels(self, edges, min_dev=2.0, min_size=10, cap=1 << 16):
        """Stage inspection: (kernels[n, 7] float64 in CompVHoughKhtKernel field order, before the height pruning; hmax)."""
        H, W = edges.shape
        out = np.zeros((cap, 7), np.float64)
        n = C.c_size_t(0)
        hmax = C.c_double(0.0)
        rc = self.lib.compvhip_houghkht_kernels_u8(self.h, _ptr(edges), W, H, edges.strides[0], min_dev, min_size, _ptr(out), cap, C.byref(n), C.byref(hmax))
        if rc == E_OUT_OF_BOUND and n.value > cap:
            return self.houghkht_kernels(edges, min_dev, min_size, cap=n.value)
        self._chk(rc)
        return out[:n.value], hmax.value

    def houghkht_stage_ms(self):
        """Milliseconds of the six stages of the last houghkht() call: link, subdivide, statistics, prune, vote + peaks, sort + sweep."""
        ms = np.zeros(6, np.float64)
        self._chk(self.lib.compvhip_houghkht_stage_ms(self.h, _ptr(ms)))
        return ms


class Plan:
    """Batched device-resident pipeline (compvhip_plan). Pointers are raw device addresses (e.g. torch .data_ptr())."""

    def __init__(self, ctx, W, H, S, frames, theta_deg=1.0):
        self.ctx = ctx
        self.lib = ctx.lib
        h = C.c_void_p()
        ctx._chk(self.lib.compvhip_plan_create(ctx.h, W, H, S, frames, theta_deg, C.byref(h)))
        self.h = h
        self.W, self.H, self.S, self.frames = W, H, S, frames
        self.timing_mode = 0

    def close(self):
        if self.h:
            self.lib.compvhip_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def canny(self, d_in, tLow, tHigh, d_edges, ksize=3, threshold_type=THRESHOLD_COMPARE_TO_GRADIENT, stream=0):
        self.ctx._chk(self.lib.compvhip_plan_canny(self.h, d_in, tLow, tHigh, ksize, threshold_type, d_edges, stream))

    def edge_dete(self, d_in, op, d_out, stream=0):
        self.ctx._chk(self.lib.compvhip_plan_edge_dete(self.h, d_in, op, d_out, stream))

    def grayscale(self, d_in, pixfmt, d_gray, stream=0):
        self.ctx._chk(self.lib.compvhip_plan_grayscale(self.h, d_in, pixfmt, d_gray, stream))

    def convlt_fixedpoint(self, d_in, vt, hz, d_out, stream=0):
        vt = np.ascontiguousarray(vt, np.uint16); hz = np.ascontiguousarray(hz, np.uint16)
        self.ctx._chk(self.lib.compvhip_plan_convlt1_fixedpoint(self.h, d_in, _ptr(vt), _ptr(hz), len(vt), d_out, stream))

    def to_cartesian(self, d_lines, d_counts, line_cap, d_cart, stream=0):
        self.ctx._chk(self.lib.compvhip_plan_to_cartesian(self.h, d_lines, d_counts, line_cap, d_cart, stream))

    def otsu(self, d_gray, d_thresholds, stream=0):
        self.ctx._chk(self.lib.compvhip_plan_otsu(self.h, d_gray, d_thresholds, stream))

    def houghsht(self, d_edges, threshold, max_lines, d_lines, line_cap, d_counts, stream=0):
        self.ctx._chk(self.lib.compvhip_plan_houghsht(self.h, d_edges, threshold, max_lines, d_lines, line_cap, d_counts, stream))

    def pipeline(self, d_in, tLow, tHigh, threshold, max_lines, d_edges, d_lines, line_cap, d_counts, stream=0):
        self.ctx._chk(self.lib.compvhip_plan_pipeline(self.h, d_in, tLow, tHigh, threshold, max_lines, d_edges, d_lines, line_cap,
                                                      d_counts, stream))

    def pipeline_async(self, d_in, tLow, tHigh, threshold, max_lines, d_edges, d_lines, line_cap, d_counts, stream=0):
        """Enqueue one step without waiting for its hysteresis flag; returns the ticket for wait()."""
        t = C.c_int(-1)
        self.ctx._chk(self.lib.compvhip_plan_pipeline_async(self.h, d_in, tLow, tHigh, threshold, max_lines, d_edges, d_lines, line_cap,
                                                            d_counts, stream, C.byref(t)))
        return t.value

    def pipeline_ex(self, d_in, tLow, tHigh, threshold, max_lines, d_edges, d_lines, line_cap, d_counts, ksize=3,
                    threshold_type=THRESHOLD_COMPARE_TO_GRADIENT, pixfmt=FMT_Y, d_gray=0, d_otsu=0, d_cart=0, stream=0, asynchronous=False):
        """[grayscale ->] Canny (any kernel size / threshold mode) -> SHT [-> toCartesian] as one enqueue; returns the ticket when asynchronous."""
        o = PipelineOpts(tLow, tHigh, threshold, max_lines, ksize, threshold_type, pixfmt, d_gray or None, d_otsu or None, d_cart or None)
        t = C.c_int(-1)
        self.ctx._chk(self.lib.compvhip_plan_pipeline_ex(self.h, d_in, C.byref(o), d_edges, d_lines, line_cap, d_counts, stream,
                                                         C.byref(t) if asynchronous else None))
        return t.value if asynchronous else None

    def houghkht(self, d_edges, rho=1.0, theta_deg=1.0, threshold=1, max_lines=0, min_dev=2.0, min_size=10, min_height=0.002, cap=1 << 14, threads=0):
        """CompVHoughKht::process on the plan's device edge maps; returns ([lines of frame f as a LINE_DTYPE array], [GS of frame f or None])."""
        F = self.frames
        lines = np.zeros((F, cap), LINE_DTYPE)
        counts = np.zeros(F, np.uint64)
        gs = np.full(F, np.nan, np.float64)
        self.ctx._chk(self.lib.compvhip_plan_houghkht(self.h, d_edges, rho, theta_deg, threshold, max_lines, min_dev, min_size, min_height,
                                                      _ptr(lines), cap, _ptr(counts), _ptr(gs), threads))
        return [lines[f][:int(counts[f])] for f in range(F)], [None if np.isnan(g) else float(g) for g in gs]

    def houghkht_stage_ms(self):
        ms = (C.c_double * 6)(); wall = C.c_double(0); th = C.c_int(0)
        self.ctx._chk(self.lib.compvhip_plan_houghkht_stage_ms(self.h, ms, C.byref(wall), C.byref(th)))
        names = ["link", "subdivide", "statistics", "prune_gmin", "vote_peaks", "sort_sweep"]
        F = max(1, self.frames)
        host = ms[0] + ms[3] + ms[5]
        return {"stages": {n: round(ms[i] / F, 4) for i, n in enumerate(names)}, "wall_ms": wall.value, "threads": th.value,
                "host_share": round(host / max(sum(ms), 1e-9), 3)}

    def wait(self, ticket):
        self.ctx._chk(self.lib.compvhip_plan_wait(self.h, ticket))

    def acc(self, frame):
        p, R, T, pitch = C.c_void_p(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        self.ctx._chk(self.lib.compvhip_plan_acc(self.h, frame, C.byref(p), C.byref(R), C.byref(T), C.byref(pitch)))
        return p.value, R.value, T.value, pitch.value

    def acc_export(self, frame, d_out, out_stride, stream=0):
        self.ctx._chk(self.lib.compvhip_plan_acc_export(self.h, frame, d_out, out_stride, stream))

    def edge_counts_ptr(self):
        p = C.c_void_p()
        self.ctx._chk(self.lib.compvhip_plan_edge_counts(self.h, C.byref(p)))
        return p.value

    def set_timing(self, mode=1):
        """0/False = off, 1/True = HIP events around every kernel, 2 = only around the two roofline kernels."""
        self.ctx._chk(self.lib.compvhip_plan_set_timing(self.h, int(mode)))
        self.timing_mode = int(mode)

    def get_timing(self, cap=256):
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        n = self.lib.compvhip_plan_get_timing(self.h, names, ms, cap)
        return [(names[i].decode(), ms[i]) for i in range(max(n, 0))]


def houghkht_link(edges, min_size=10):
    """The host stage of KHT alone (compvhip_houghkht_link_u8, no device): (points[n, 2] int32 as (x, y), string end indices)."""
    lib = load()
    e = np.ascontiguousarray(edges, dtype=np.uint8)
    H, W = e.shape
    sz = C.c_size_t
    lib.compvhip_houghkht_link_u8.argtypes = [C.c_void_p, sz, sz, sz, sz, C.c_void_p, sz, C.c_void_p, C.c_void_p, sz, C.c_void_p]
    n_set = int((e != 0).sum())
    xy = np.zeros((max(n_set, 1), 2), np.int32)
    ends = np.zeros(max(n_set, 1), np.uint32)
    npts, nstr = sz(0), sz(0)
    rc = lib.compvhip_houghkht_link_u8(_ptr(e), W, H, e.strides[0], min_size, _ptr(xy), len(xy), C.byref(npts), _ptr(ends), len(ends), C.byref(nstr))
    if rc:
        raise CompvHipError(rc, "compvhip_houghkht_link_u8")
    return xy[:npts.value].copy(), ends[:nstr.value].copy()


def host_cpu_budget():
    """CPUs this process may really use at once: min(hardware threads, affinity mask, cgroup quota) (compvhip_host_cpu_budget)."""
    lib = load()
    lib.compvhip_host_cpu_budget.restype = C.c_int
    return int(lib.compvhip_host_cpu_budget())


def houghsht_vote_grid(W, H, theta_deg=1.0, frames=1):
    """(nx, ny, window rows) of the voting kernel's image-tile grid for a plan of `frames` W x H frames (compvhip_houghsht_vote_grid)."""
    lib = load()
    nx = C.c_int(0); ny = C.c_int(0); rw = C.c_int(0)
    rc = lib.compvhip_houghsht_vote_grid(W, H, theta_deg, frames, C.byref(nx), C.byref(ny), C.byref(rw))
    if rc:
        raise CompvHipError(rc, "compvhip_houghsht_vote_grid")
    return nx.value, ny.value, rw.value


def houghkht_dims(W, H, rho=1.0, theta_deg=1.0):
    """(T, rhoN) of the KHT vote map for a W x H image (compvhip_houghkht_dims)."""
    lib = load()
    r = C.c_size_t(0); t = C.c_size_t(0)
    lib.compvhip_houghkht_dims.argtypes = [C.c_size_t, C.c_size_t, C.c_float, C.c_float, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    rc = lib.compvhip_houghkht_dims(W, H, rho, theta_deg, C.byref(r), C.byref(t))
    if rc:
        raise CompvHipError(rc, "compvhip_houghkht_dims")
    return int(t.value), int(r.value)


def to_cartesian(W, H, lines, kht=False):
    """CompVHoughSht / CompVHoughKht::toCartesian for (rho, theta) pairs: (n, 4) float32 a.x, a.y, b.x, b.y (host arithmetic, no GPU needed)."""
    L = load()
    n = len(lines)
    buf = np.zeros(max(n, 1), LINE_DTYPE)
    for i, l in enumerate(lines):
        buf[i]["rho"] = l[0]; buf[i]["theta"] = l[1]
    out = np.zeros((max(n, 1), 4), np.float32)
    fn = L.compvhip_houghkht_to_cartesian if kht else L.compvhip_houghsht_to_cartesian
    fn.argtypes = [C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    rc = fn(W, H, _ptr(buf), n, _ptr(out))
    if rc != OK:
        raise CompvHipError(rc, "invalid toCartesian parameters")
    return out[:n]


def gauss_kernel_fixedpoint(size, sigma):
    """CompVMathGauss::kernelDim1FixedPoint (host arithmetic, no GPU needed)."""
    k = np.zeros(size, np.uint16)
    rc = load().compvhip_gauss_kernel_fixedpoint(size, sigma, _ptr(k))
    if rc != OK:
        raise CompvHipError(rc, "invalid Gaussian kernel parameters")
    return k
