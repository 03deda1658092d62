"""ctypes bindings for the TEST-ONLY checkers:

* ``Oracle``  -> oracle/liboracle.so   (C restatement of the CompV hot path, oracle/compv_oracle.c)
* ``RefShim`` -> oracle/_ref/libcompv_refshim.so (the real CompV library compiled from /root/reference by
  oracle/build_ref.sh; present in the build container and, as a prebuilt .so, on the GPU box)

Nothing in compv_amd/ imports this module.
"""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


def synth_frame(W, H, seed=12345):
    """SURVEY.md 8(d) synthetic frame, vectorised (LCG jump-ahead by doubling). Bit-identical to
    orc_synth_frame() in oracle/compv_oracle.c (checked by tests/test_oracle.py)."""
    n = W * H
    M = np.uint64(0xFFFFFFFF)
    a = np.uint64(1664525)
    c = np.uint64(1013904223)
    ap = np.array([1], dtype=np.uint64)  # a^k
    cp = np.array([0], dtype=np.uint64)  # c_k with s_k = a^k s_0 + c_k
    while len(ap) <= n:
        aL = (ap[-1] * a) & M
        cL = (cp[-1] * a + c) & M
        ap, cp = np.concatenate([ap, (ap * aL) & M]), np.concatenate([cp, (ap * cL + cp) & M])
    s = ((ap[1:n + 1] * np.uint64(seed & 0xFFFFFFFF)) + cp[1:n + 1]) & M
    s = s.astype(np.uint32).reshape(H, W)
    i = np.arange(W, dtype=np.int64)[None, :]
    j = np.arange(H, dtype=np.int64)[:, None]
    v = 40 + (((i // 64 + j // 64) & 1) * 150) + (s >> 28).astype(np.int64)
    v = np.where(((i + 2 * j) % 257) < 3, 255, v)
    return v.astype(np.uint8)


def libstdcxx_version():
    """Newest GLIBCXX symbol version of the libstdc++ this process loads, e.g. 'GLIBCXX_3.4.30'.  The order of equal-strength Hough lines
    is "what this runtime's std::sort (introsort) does" -- in the reference and in the host entry points that reproduce it -- so the
    fixtures that pin that order record the runtime they were generated with (tests/golden/golden_sht_order.json)."""
    import re
    C.CDLL("libstdc++.so.6")
    path = None
    with open("/proc/self/maps") as f:
        for line in f:
            if "libstdc++.so" in line:
                path = line.split()[-1]
                break
    if not path:
        return None
    vers = set(re.findall(rb"GLIBCXX_3\.4\.(\d+)", open(path, "rb").read()))
    return "GLIBCXX_3.4.%d" % max(int(v) for v in vers) if vers else None


def md5_rows(a):
    """MD5 over the valid bytes of each row (the reference's compv_tests_md5 convention,
    tests/tests_common.cxx:98-117)."""
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OrcLine(C.Structure):
    _fields_ = [("rho", C.c_float), ("theta", C.c_float), ("strength", C.c_int64), ("row", C.c_int32), ("col", C.c_int32)]


class KhtLine(C.Structure):
    _fields_ = [("rho", C.c_float), ("theta", C.c_float), ("strength", C.c_int32), ("rho_index", C.c_int32), ("theta_index", C.c_int32)]


class RefLine(C.Structure):
    _fields_ = [("rho", C.c_float), ("theta", C.c_float), ("strength", C.c_longlong)]


def build_oracle():
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("compv_oracle.c", "kht_oracle.c")]
    cxx = os.path.join(ORACLE_DIR, "kht_sort.cpp")
    deps = srcs + [cxx] + [os.path.join(ORACLE_DIR, f) for f in ("compv_oracle.h", "kht_oracle.h")]
    out = os.path.join(ORACLE_DIR, "liboracle.so")
    if (not os.path.exists(out)) or any(os.path.getmtime(out) < os.path.getmtime(d) for d in deps):
        # -ffp-contract=off: the reference's KHT is plain IEEE double arithmetic (no FMA contraction)
        objs = []
        for s in srcs:
            o = s[:-2] + ".o"
            subprocess.check_call(["gcc", "-O2", "-fPIC", "-std=gnu11", "-ffp-contract=off", "-c", s, "-o", o])
            objs.append(o)
        o = cxx[:-4] + ".o"
        subprocess.check_call(["g++", "-O2", "-fPIC", "-std=c++11", "-c", cxx, "-o", o])
        objs.append(o)
        subprocess.check_call(["g++", "-shared", "-o", out] + objs + ["-lm"])
    return out


class Oracle:
    def __init__(self):
        self.lib = C.CDLL(build_oracle())
        L = self.lib
        sz = C.c_size_t
        L.orc_synth_frame.argtypes = [C.c_void_p, sz, sz, sz, C.c_uint32]
        L.orc_synth_frame.restype = None
        L.orc_convlt1_8u16s16s.argtypes = [C.c_void_p, sz, sz, sz, C.c_void_p, C.c_void_p, sz, C.c_void_p]
        L.orc_convlt1_16s16s16s.argtypes = [C.c_void_p, sz, sz, sz, C.c_void_p, C.c_void_p, sz, C.c_void_p]
        L.orc_gradient.argtypes = [C.c_void_p, sz, sz, sz, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_edge_dete.argtypes = [C.c_void_p, sz, sz, sz, C.c_int, C.c_void_p, sz, C.c_void_p]
        L.orc_canny_thresholds.argtypes = [C.c_float, C.c_float, C.c_int, C.c_uint32, sz, sz, C.c_void_p, C.c_void_p]
        L.orc_canny_coverage.argtypes = [sz, C.c_void_p, C.c_void_p]
        L.orc_canny_coverage.restype = None
        L.orc_canny.argtypes = [C.c_void_p, sz, sz, sz, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, sz, C.c_void_p]
        L.orc_sht_dims.argtypes = [sz, sz, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_sht_tables.argtypes = [C.c_float, sz, C.c_void_p, C.c_void_p]
        L.orc_sht_acc.argtypes = [C.c_void_p, sz, sz, sz, C.c_void_p, C.c_void_p, sz, C.c_void_p, sz]
        L.orc_sht_lines.argtypes = [C.c_void_p, sz, sz, sz, C.c_int32, C.c_int32, C.c_float, C.c_int, C.c_void_p, sz, C.c_void_p]
        L.orc_sht.argtypes = [C.c_void_p, sz, sz, sz, C.c_float, C.c_int32, C.c_int, C.c_void_p, sz, C.c_void_p]

    def kht(self, edges, rho=1.0, theta_deg=1.0, threshold=1, max_lines=0, min_dev=2.0, min_size=10, min_height=0.002, cap=1 << 16):
        """CompVHoughKht::process restated: returns ([(rho, theta, strength, rho_index, theta_index)], GS)."""
        H, W = edges.shape
        L = self.lib
        L.orc_kht.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_float, C.c_float, C.c_int32, C.c_int, C.c_double, C.c_size_t,
                              C.c_double, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        buf = (KhtLine * cap)(); n = C.c_size_t(0); gs = C.c_double(0)
        r = L.orc_kht(_p(edges), W, H, edges.strides[0], rho, theta_deg, threshold, max_lines, min_dev, min_size, min_height, buf, cap, C.byref(n), C.byref(gs))
        assert r == 0, r
        return [(buf[i].rho, buf[i].theta, buf[i].strength, buf[i].rho_index, buf[i].theta_index) for i in range(min(n.value, cap))], gs.value

    def kht_link(self, edges, min_size=10):
        """linking_AppendixA restated (oracle/kht_oracle.c::orc_kht_link): (points[n, 2] int32 as (x, y), string end indices)."""
        H, W = edges.shape
        L = self.lib
        sz = C.c_size_t

        class Pos(C.Structure):
            _fields_ = [("y", C.c_int), ("x", C.c_int), ("cy", C.c_double), ("cx", C.c_double)]

        class Range(C.Structure):
            _fields_ = [("begin", sz), ("end", sz)]
        L.orc_kht_link.argtypes = [C.c_void_p, sz, sz, sz, sz, C.POINTER(C.c_void_p), C.POINTER(sz), C.POINTER(C.c_void_p), C.POINTER(sz)]
        L.orc_free.argtypes = [C.c_void_p]
        poss = C.c_void_p(); strings = C.c_void_p()
        npos = sz(0); ns = sz(0)
        assert L.orc_kht_link(_p(edges), W, H, edges.strides[0], min_size, C.byref(poss), C.byref(npos), C.byref(strings), C.byref(ns)) == 0
        try:
            pa = C.cast(poss, C.POINTER(Pos)); sa = C.cast(strings, C.POINTER(Range))
            ends = np.array([sa[i].end for i in range(ns.value)], np.uint32)
            n = int(ends[-1]) if ns.value else 0
            xy = np.array([(pa[i].x, pa[i].y) for i in range(n)], np.int32).reshape(n, 2)
            if ns.value:
                assert sa[0].begin == 0 and all(sa[i].begin == sa[i - 1].end for i in range(1, ns.value))   # strings are contiguous
            return xy, ends
        finally:
            for q in (poss, strings):
                if q.value:
                    L.orc_free(q)

    def kht_kernels(self, edges, min_dev=2.0, min_size=10):
        """linking_AppendixA + clusters_find + voting_Algorithm2_Kernels restated: (kernels[n, 7] float64 in CompVHoughKhtKernel field
        order, before the height pruning; hmax)."""
        H, W = edges.shape
        L = self.lib
        sz = C.c_size_t
        L.orc_kht_link.argtypes = [C.c_void_p, sz, sz, sz, sz, C.POINTER(C.c_void_p), C.POINTER(sz), C.POINTER(C.c_void_p), C.POINTER(sz)]
        L.orc_kht_clusters.argtypes = [C.c_void_p, C.c_void_p, sz, sz, C.c_double, C.POINTER(C.c_void_p), C.POINTER(sz)]
        L.orc_kht_kernels.argtypes = [C.c_void_p, C.c_void_p, sz, C.c_void_p, C.POINTER(C.c_double)]
        L.orc_free.argtypes = [C.c_void_p]
        poss = C.c_void_p(); strings = C.c_void_p(); clusters = C.c_void_p()
        npos = sz(0); ns = sz(0); nc = sz(0)
        assert L.orc_kht_link(_p(edges), W, H, edges.strides[0], min_size, C.byref(poss), C.byref(npos), C.byref(strings), C.byref(ns)) == 0
        try:
            if ns.value == 0:
                return np.zeros((0, 7)), 0.0
            assert L.orc_kht_clusters(poss, strings, ns.value, min_size, min_dev, C.byref(clusters), C.byref(nc)) == 0
            out = np.zeros((nc.value, 7), np.float64)
            hmax = C.c_double(0.0)
            if nc.value:
                assert L.orc_kht_kernels(poss, clusters, nc.value, _p(out), C.byref(hmax)) == 0
            return out, hmax.value
        finally:
            for q in (poss, strings, clusters):
                if q.value:
                    L.orc_free(q)

    def synth(self, W, H, seed=12345):
        out = np.zeros((H, W), np.uint8)
        self.lib.orc_synth_frame(_p(out), W, H, W, seed)
        return out

    def sht_to_cartesian(self, W, H, lines, kht=False):
        """lines: iterable of (rho, theta, ...) -> (n, 4) float32 array a.x, a.y, b.x, b.y (CompVHoughSht / CompVHoughKht::toCartesian)."""
        n = len(lines)
        buf = (OrcLine * max(n, 1))()
        for i, l in enumerate(lines):
            buf[i].rho = l[0]; buf[i].theta = l[1]
        out = np.zeros((max(n, 1), 4), np.float32)
        fn = self.lib.orc_kht_to_cartesian if kht else self.lib.orc_sht_to_cartesian
        fn.argtypes = [C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        fn.restype = None
        fn(W, H, buf, n, _p(out))
        return out[:n]

    def kht_to_cartesian(self, W, H, lines):
        return self.sht_to_cartesian(W, H, lines, kht=True)

    # ---- optional Gaussian pre-blur ----
    def gauss_kernel_f32(self, size, sigma):
        k = np.zeros(size, np.float32)
        L = self.lib
        L.orc_gauss_kernel_f32.argtypes = [C.c_size_t, C.c_float, C.c_void_p]
        assert L.orc_gauss_kernel_f32(size, sigma, _p(k)) == 0
        return k

    def gauss_kernel_fxp(self, size, sigma):
        k = np.zeros(size, np.uint16)
        L = self.lib
        L.orc_gauss_kernel_fixedpoint.argtypes = [C.c_size_t, C.c_float, C.c_void_p]
        assert L.orc_gauss_kernel_fixedpoint(size, sigma, _p(k)) == 0
        return k

    def convlt_fxp(self, img, vt, hz):
        H, W = img.shape
        S = img.strides[0]
        out = np.zeros((H, S), np.uint8)
        vt = np.ascontiguousarray(vt, np.uint16); hz = np.ascontiguousarray(hz, np.uint16)
        L = self.lib
        L.orc_convlt1_fixedpoint.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        r = L.orc_convlt1_fixedpoint(_p(img), W, H, S, _p(vt), _p(hz), len(vt), _p(out))
        return r, out[:, :W]

    # ---- caller-side pre-processing ----
    def fmt_bytes(self, fmt):
        return self.lib.orc_fmt_bytes(fmt)

    def grayscale(self, packed, fmt, W):
        """packed: (H, S*bpp) uint8 rows of S samples; returns (H, W) luma."""
        H = packed.shape[0]
        bpp = self.fmt_bytes(fmt)
        S = packed.shape[1] // bpp
        out = np.zeros((H, W), np.uint8)
        L = self.lib
        L.orc_grayscale.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t]
        assert L.orc_grayscale(_p(packed), fmt, W, H, S, _p(out), W) == 0
        return out

    def hist256(self, img):
        H, W = img.shape
        h = np.zeros(256, np.uint32)
        L = self.lib
        L.orc_hist256.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
        L.orc_hist256.restype = None
        L.orc_hist256(_p(img), W, H, img.strides[0], _p(h))
        return h

    def otsu(self, img):
        H, W = img.shape
        L = self.lib
        L.orc_otsu.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t]
        return L.orc_otsu(_p(img), W, H, img.strides[0])

    def otsu_canny_thresholds(self, t, flow=0.5, fhigh=1.0):
        L = self.lib
        L.orc_otsu_canny_thresholds.argtypes = [C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.orc_otsu_canny_thresholds.restype = None
        lo = C.c_int(0); hi = C.c_int(0)
        L.orc_otsu_canny_thresholds(t, flow, fhigh, C.byref(lo), C.byref(hi))
        return lo.value, hi.value

    def convlt_8u(self, img, vt, hz):
        H, W = img.shape
        vt = np.ascontiguousarray(vt, np.int16); hz = np.ascontiguousarray(hz, np.int16)
        out = np.zeros((H, img.strides[0]), np.int16)
        r = self.lib.orc_convlt1_8u16s16s(_p(img), W, H, img.strides[0], _p(vt), _p(hz), len(vt), _p(out))
        return r, out[:, :W]

    def convlt_16s(self, img, vt, hz):
        H, W = img.shape
        vt = np.ascontiguousarray(vt, np.int16); hz = np.ascontiguousarray(hz, np.int16)
        out = np.zeros((H, img.strides[0] // 2), np.int16)
        r = self.lib.orc_convlt1_16s16s16s(_p(img), W, H, img.strides[0] // 2, _p(vt), _p(hz), len(vt), _p(out))
        return r, out[:, :W]

    def gradient(self, img, op=0):
        H, W = img.shape
        S = img.strides[0]
        gx = np.zeros((H, S), np.int16); gy = np.zeros((H, S), np.int16); g = np.zeros((H, S), np.uint16)
        r = self.lib.orc_gradient(_p(img), W, H, S, op, _p(gx), _p(gy), _p(g))
        assert r == 0, r
        return gx[:, :W], gy[:, :W], g[:, :W]

    def edge_dete(self, img, op=0):
        H, W = img.shape
        out = np.zeros((H, W), np.uint8)
        gmax = C.c_uint16(0)
        r = self.lib.orc_edge_dete(_p(img), W, H, img.strides[0], op, _p(out), W, C.byref(gmax))
        assert r == 0, r
        return out, gmax.value

    def canny_thresholds(self, fLow, fHigh, typ=0, total=0, W=1, H=1):
        lo = C.c_uint16(0); hi = C.c_uint16(0)
        r = self.lib.orc_canny_thresholds(fLow, fHigh, typ, total, W, H, C.byref(lo), C.byref(hi))
        return r, lo.value, hi.value

    def canny_coverage(self, W):
        a = C.c_size_t(0); b = C.c_size_t(0)
        self.lib.orc_canny_coverage(W, C.byref(a), C.byref(b))
        return a.value, b.value

    def canny(self, img, fLow, fHigh, ksize=3, typ=0, want_gnms=False):
        H, W = img.shape
        out = np.zeros((H, W), np.uint8)
        gn = np.zeros((H, img.strides[0]), np.uint16) if want_gnms else None
        r = self.lib.orc_canny(_p(img), W, H, img.strides[0], fLow, fHigh, ksize, typ, _p(out), W, _p(gn) if want_gnms else None)
        if want_gnms:
            return r, out, gn[:, :W]
        return r, out

    def sht_dims(self, W, H, theta_deg=1.0):
        R = C.c_size_t(0); T = C.c_size_t(0); th = C.c_float(0)
        r = self.lib.orc_sht_dims(W, H, theta_deg, C.byref(R), C.byref(T), C.byref(th))
        assert r == 0
        return R.value, T.value, th.value

    def sht_tables(self, theta_deg, T):
        s = np.zeros(T, np.int32); c = np.zeros(T, np.int32)
        self.lib.orc_sht_tables(theta_deg, T, _p(s), _p(c))
        return s, c

    def sht_acc(self, edges, theta_deg=1.0):
        H, W = edges.shape
        R, T, _ = self.sht_dims(W, H, theta_deg)
        s, c = self.sht_tables(theta_deg, T)
        acc = np.zeros((R, T), np.int32)
        self.lib.orc_sht_acc(_p(edges), W, H, edges.strides[0], _p(s), _p(c), T, _p(acc), T)
        return acc

    def sht_lines_from_acc(self, acc, W, H, theta_deg, threshold, max_lines=0):
        R, T = acc.shape
        _, _, th = self.sht_dims(W, H, theta_deg)
        cap = max(1, int((acc > threshold).sum()))
        buf = (OrcLine * cap)()
        n = C.c_size_t(0)
        r = self.lib.orc_sht_lines(_p(acc), R, T, acc.strides[0] // 4, threshold, W + H, th, max_lines, buf, cap, C.byref(n))
        assert r == 0, r
        return [(buf[i].rho, buf[i].theta, buf[i].strength, buf[i].row, buf[i].col) for i in range(min(n.value, cap))]

    def sht_lines_from_acc_reference_order(self, acc, W, H, theta_deg, threshold, max_lines=0):
        """The line list in the order the reference returns it (its unstable std::sort on the strength alone, applied to the
        (row, col) emission order, then the first max_lines): all lines -> orc_sht_reference_order -> truncation."""
        R, T = acc.shape
        _, _, th = self.sht_dims(W, H, theta_deg)
        cap = max(1, int((acc > threshold).sum()))
        buf = (OrcLine * cap)()
        n = C.c_size_t(0)
        r = self.lib.orc_sht_lines(_p(acc), R, T, acc.strides[0] // 4, threshold, W + H, th, 0, buf, cap, C.byref(n))
        assert r == 0 and n.value <= cap, (r, n.value, cap)
        self.lib.orc_sht_reference_order.argtypes = [C.c_void_p, C.c_size_t]
        self.lib.orc_sht_reference_order(buf, n.value)
        keep = n.value if max_lines <= 0 else min(n.value, max_lines)
        return [(buf[i].rho, buf[i].theta, buf[i].strength, buf[i].row, buf[i].col) for i in range(keep)]

    def sht(self, edges, theta_deg=1.0, threshold=100, max_lines=0, reference_order=False):
        acc = self.sht_acc(edges, theta_deg)
        H, W = edges.shape
        if reference_order:
            return self.sht_lines_from_acc_reference_order(acc, W, H, theta_deg, threshold, max_lines)
        return self.sht_lines_from_acc(acc, W, H, theta_deg, threshold, max_lines)


def refshim_path():
    return os.path.join(ORACLE_DIR, "_ref", "libcompv_refshim.so")


def have_refshim():
    return os.path.exists(refshim_path())


class RefShim:
    """The real CompV CPU library (intrinsics path). threads: 1 or -1 (all cores)."""

    def __init__(self, threads=1):
        self.lib = C.CDLL(refshim_path())
        L = self.lib
        sz = C.c_size_t
        L.refshim_init.argtypes = [C.c_int]
        L.refshim_sobel.argtypes = [C.c_void_p, sz, sz, sz, C.c_void_p, sz]
        L.refshim_canny.argtypes = [C.c_void_p, sz, sz, sz, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, sz]
        if hasattr(L, "refshim_edge_dete"):
            L.refshim_edge_dete.argtypes = [C.c_void_p, sz, sz, sz, C.c_int, C.c_void_p, sz]
        L.refshim_sht.argtypes = [C.c_void_p, sz, sz, sz, C.c_float, sz, C.c_int, C.c_void_p, sz, C.c_void_p]
        L.refshim_kht.argtypes = [C.c_void_p, sz, sz, sz, C.c_float, C.c_float, sz, C.c_int, C.c_void_p, sz, C.c_void_p, C.c_void_p]
        L.refshim_sht_acc.argtypes = [C.c_void_p, sz, sz, sz, C.c_void_p, C.c_void_p, sz, C.c_void_p, sz]
        L.refshim_convlt1_8u16s16s.argtypes = [C.c_void_p, sz, sz, sz, C.c_void_p, C.c_void_p, sz, C.c_void_p]
        L.refshim_convlt1_16s16s16s.argtypes = [C.c_void_p, sz, sz, sz, C.c_void_p, C.c_void_p, sz, C.c_void_p]
        L.refshim_bench_pipeline.argtypes = [C.c_void_p, sz, sz, sz, sz, C.c_float, C.c_float, C.c_float, sz, C.c_int, C.c_void_p, C.c_void_p]
        L.refshim_bench_pipeline.restype = C.c_double
        assert L.refshim_init(threads) == 0
        self.threads = L.refshim_threads()
        self.avx2 = bool(L.refshim_has_avx2())

    def reinit(self, threads):
        assert self.lib.refshim_init(threads) == 0
        self.threads = self.lib.refshim_threads()

    def sht_to_cartesian(self, W, H, lines, kht=False):
        n = len(lines)
        buf = (RefLine * max(n, 1))()
        for i, l in enumerate(lines):
            buf[i].rho = l[0]; buf[i].theta = l[1]; buf[i].strength = 1
        out = np.zeros((max(n, 1), 4), np.float32)
        fn = self.lib.refshim_kht_to_cartesian if kht else self.lib.refshim_sht_to_cartesian
        fn.argtypes = [C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        assert fn(W, H, buf, n, _p(out)) == 0
        return out[:n]

    def kht_to_cartesian(self, W, H, lines):
        return self.sht_to_cartesian(W, H, lines, kht=True)

    def gauss_kernel_f32(self, size, sigma):
        k = np.zeros(size, np.float32)
        L = self.lib
        L.refshim_gauss_kernel_f32.argtypes = [C.c_size_t, C.c_float, C.c_void_p]
        assert L.refshim_gauss_kernel_f32(size, sigma, _p(k)) == 0
        return k

    def gauss_kernel_fxp(self, size, sigma):
        k = np.zeros(size, np.uint16)
        L = self.lib
        L.refshim_gauss_kernel_fixedpoint.argtypes = [C.c_size_t, C.c_float, C.c_void_p]
        assert L.refshim_gauss_kernel_fixedpoint(size, sigma, _p(k)) == 0
        return k

    def convlt_fxp(self, img, vt, hz):
        H, W = img.shape
        S = img.strides[0]
        out = np.zeros((H, S), np.uint8)
        vt = np.ascontiguousarray(vt, np.uint16); hz = np.ascontiguousarray(hz, np.uint16)
        L = self.lib
        L.refshim_convlt1_fixedpoint.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        r = L.refshim_convlt1_fixedpoint(_p(img), W, H, S, _p(vt), _p(hz), len(vt), _p(out))
        return r, out[:, :W]

    def grayscale(self, packed, fmt, W, bpp):
        H = packed.shape[0]
        S = packed.shape[1] // bpp
        out = np.zeros((H, W), np.uint8)
        L = self.lib
        L.refshim_grayscale.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t]
        r = L.refshim_grayscale(_p(packed), fmt, W, H, S, _p(out), W)
        assert r == 0, r
        return out

    def otsu(self, img):
        H, W = img.shape
        L = self.lib
        L.refshim_otsu.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
        t = C.c_double(0)
        assert L.refshim_otsu(_p(img), W, H, img.strides[0], C.byref(t)) == 0
        return t.value

    def sobel(self, img):
        H, W = img.shape
        out = np.zeros((H, W), np.uint8)
        assert self.lib.refshim_sobel(_p(img), W, H, img.strides[0], _p(out), W) == 0
        return out

    def edge_dete(self, img, op=0):
        """op: 0 Sobel, 2 Scharr, 3 Prewitt (include/compv_hip.h ids)"""
        H, W = img.shape
        out = np.zeros((H, W), np.uint8)
        assert self.lib.refshim_edge_dete(_p(img), W, H, img.strides[0], op, _p(out), W) == 0
        return out

    def canny(self, img, fLow, fHigh, ksize=3, typ=0):
        H, W = img.shape
        out = np.zeros((H, W), np.uint8)
        r = self.lib.refshim_canny(_p(img), W, H, img.strides[0], fLow, fHigh, ksize, typ, _p(out), W)
        return r, out

    def _lines(self, buf, n):
        return [(buf[i].rho, buf[i].theta, buf[i].strength) for i in range(n)]

    def sht(self, edges, theta_deg=1.0, threshold=100, max_lines=0, cap=1 << 20):
        H, W = edges.shape
        buf = (RefLine * cap)(); n = C.c_size_t(0)
        r = self.lib.refshim_sht(_p(edges), W, H, edges.strides[0], theta_deg, threshold, max_lines, buf, cap, C.byref(n))
        assert r == 0, r
        assert n.value <= cap
        return self._lines(buf, n.value)

    def kht(self, edges, rho=1.0, theta_deg=1.0, threshold=1, max_lines=0, cap=1 << 18):
        H, W = edges.shape
        buf = (RefLine * cap)(); n = C.c_size_t(0); gs = C.c_double(0)
        r = self.lib.refshim_kht(_p(edges), W, H, edges.strides[0], rho, theta_deg, threshold, max_lines, buf, cap, C.byref(n), C.byref(gs))
        assert r == 0, r
        return self._lines(buf, min(n.value, cap)), gs.value

    def sht_acc(self, edges, sinQ, cosQ, R):
        H, W = edges.shape
        T = len(sinQ)
        stride = (T + 15) & ~15
        acc = np.zeros((R, stride), np.int32)
        r = self.lib.refshim_sht_acc(_p(edges), W, H, edges.strides[0], _p(sinQ), _p(cosQ), T, _p(acc), stride)
        assert r == 0, r
        return acc[:, :T].copy()

    def convlt_8u(self, img, vt, hz, S=None):
        H, W = img.shape
        vt = np.ascontiguousarray(vt, np.int16); hz = np.ascontiguousarray(hz, np.int16)
        out = np.zeros((H, img.strides[0]), np.int16)
        r = self.lib.refshim_convlt1_8u16s16s(_p(img), W, H, img.strides[0], _p(vt), _p(hz), len(vt), _p(out))
        return r, out[:, :W]

    def convlt_16s(self, img, vt, hz):
        H, W = img.shape
        vt = np.ascontiguousarray(vt, np.int16); hz = np.ascontiguousarray(hz, np.int16)
        out = np.zeros((H, img.strides[0] // 2), np.int16)
        r = self.lib.refshim_convlt1_16s16s16s(_p(img), W, H, img.strides[0] // 2, _p(vt), _p(hz), len(vt), _p(out))
        return r, out[:, :W]

    def bench_kht(self, edge_maps, rho=1.0, theta_deg=1.0, threshold=1):
        """wall ms of CompVHoughKht::process over the (n, H, W) edge maps, and the number of lines found"""
        n, H, W = edge_maps.shape
        sz = C.c_size_t
        self.lib.refshim_bench_kht.argtypes = [C.c_void_p, sz, sz, sz, sz, C.c_float, C.c_float, sz, C.c_void_p]
        self.lib.refshim_bench_kht.restype = C.c_double
        l = C.c_longlong(0)
        ms = self.lib.refshim_bench_kht(_p(edge_maps), W, H, W, n, rho, theta_deg, threshold, C.byref(l))
        return ms, l.value

    def bench_pipeline(self, frames, fLow, fHigh, theta_deg, threshold, stages=3):
        n, H, W = frames.shape
        e = C.c_longlong(0); l = C.c_longlong(0)
        ms = self.lib.refshim_bench_pipeline(_p(frames), W, H, W, n, fLow, fHigh, theta_deg, threshold, stages, C.byref(e), C.byref(l))
        return ms, e.value, l.value
