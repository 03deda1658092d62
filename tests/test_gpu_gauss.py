"""GPU parity of the optional Gaussian pre-blur through the C ABI: compvhip_convlt1_fixedpoint_u8 and the plan
variant against the oracle (pinned to the compiled reference and to the reference's own golden by tests/test_gauss_oracle.py)."""
import numpy as np
import pytest

from oracle_bindings import synth_frame
from test_gauss_oracle import CASE0_KERNEL_LITERAL, case0_input, md5_rows

pytestmark = pytest.mark.gpu


def test_reference_known_answer_case0_on_gpu(hip_ctx):
    """unittests/math_convlt.cxx:21 (case 0): the reference's own MD5 for the fixed-point convolution, computed by the HIP path."""
    kern = np.array(CASE0_KERNEL_LITERAL[:7], np.uint16)
    out = hip_ctx.convlt_fixedpoint(case0_input(), kern, kern)
    assert md5_rows(out) == "2678b73a89681f12fb474dd8102fc37c"


@pytest.mark.parametrize("W,H,S", [(16, 16, 64), (64, 16, 64), (333, 77, 384), (641, 130, 704), (1920, 70, 1920), (2049, 65, 2112)])
@pytest.mark.parametrize("size", [3, 5, 7, 9, 11, 13, 15])
def test_convlt_fixedpoint_matches_oracle(hip_ctx, oracle, W, H, S, size):
    rng = np.random.default_rng(W * 31 + size)
    buf = np.zeros((H, S), np.uint8)
    buf[:, :W] = rng.integers(0, 256, (H, W), dtype=np.uint8)
    img = buf[:, :W]
    for kern in (oracle.gauss_kernel_fxp(size, 0.3 * size), rng.integers(0, 65536 // size, size).astype(np.uint16), np.full(size, 65535, np.uint16)):
        vt, hz = kern, kern[::-1].copy()
        rc, exp = oracle.convlt_fxp(img, vt, hz)
        assert rc == 0
        got = hip_ctx.convlt_fixedpoint(img, vt, hz)
        assert (got == exp).all(), int((got != exp).sum())


def test_convlt_fixedpoint_errors(hip_ctx):
    from compv_amd import capi
    img = np.zeros((32, 32), np.uint8)
    with pytest.raises(capi.CompvHipError) as e:
        hip_ctx.convlt_fixedpoint(img, [1, 2, 3, 4], [1, 2, 3, 4])           # even size (compv_math_convlt.h:100)
    assert e.value.code == capi.E_INVALID_PARAMETER
    with pytest.raises(capi.CompvHipError) as e:
        hip_ctx.convlt_fixedpoint(np.zeros((4, 32), np.uint8), [1] * 5, [1] * 5)   # H < kernel
    assert e.value.code == capi.E_INVALID_PARAMETER
    with pytest.raises(capi.CompvHipError) as e:
        hip_ctx.convlt_fixedpoint(img, [1] * 17, [1] * 17)                   # > 15 taps
    assert e.value.code == capi.E_NOT_IMPLEMENTED


def test_plan_blur_then_canny_batch(hip_ctx, oracle):
    """Gaussian(5, sigma 1) -> Canny on a 4-frame batch on the device, in place, against the oracle chain."""
    import torch
    from compv_amd import capi
    W, H, S, F = 1282, 720, 1344, 4
    dev = torch.device("cuda", 0)
    frames = np.zeros((F, H, S), np.uint8)
    kern = capi.gauss_kernel_fixedpoint(5, 1.0)
    assert kern.tolist() == oracle.gauss_kernel_fxp(5, 1.0).tolist()
    exp = []
    for f in range(F):
        frames[f, :, :W] = synth_frame(W, H, 4321 + f)
        rc, b = oracle.convlt_fxp(frames[f][:, :W], kern, kern)
        rc2, e = oracle.canny(np.ascontiguousarray(b), 30.0, 70.0)
        assert rc == 0 and rc2 == 0
        exp.append((b, e))
    d = torch.from_numpy(frames).to(dev)
    d_edges = torch.empty_like(d)
    plan = capi.Plan(hip_ctx, W, H, S, F, 1.0)
    try:
        d_blur = torch.zeros_like(d)                                           # the kernels write the W valid columns only: keep the stride padding defined
        plan.convlt_fixedpoint(d.data_ptr(), kern, kern, d_blur.data_ptr())   # out of place: fused single-kernel path
        plan.convlt_fixedpoint(d.data_ptr(), kern, kern, d.data_ptr())        # in place: two passes through the plan's scratch
        torch.cuda.synchronize()
        assert torch.equal(d_blur, d)
        plan.canny(d.data_ptr(), 30.0, 70.0, d_edges.data_ptr())
        torch.cuda.synchronize()
        b = d.cpu().numpy(); e = d_edges.cpu().numpy()
        for f in range(F):
            assert (b[f][:, :W] == exp[f][0]).all(), f
            assert (e[f][:, :W] == exp[f][1]).all(), f
    finally:
        plan.close()


# ---------------------------------------------------------------------------------------------------------------
# integer separable correlation: the reference's own known-answer vectors on the GPU
# ---------------------------------------------------------------------------------------------------------------
def test_convlt1_int16_reference_known_answers_on_the_gpu(hip_ctx):
    """unittests/math_convlt.cxx:24-25, cases 5 and 6 (1285x720, stride 1344, k = 7, data (i*j)+53 with alternating signs for the int16
    case): MD5 of the int16 result -- the only reference-held golden vectors on the hot path's operators -- from the HIP kernels."""
    from test_oracle import _convlt_inputs
    from oracle_bindings import md5_rows
    d8, d16, k = _convlt_inputs()
    assert md5_rows(hip_ctx.convlt1_i16(d8, k, k)) == "7f1116ade2a1cdb37842084c781ee05e"
    assert md5_rows(hip_ctx.convlt1_i16(d16, k, k)) == "cad2f4d2fd66e171997f39804e667699"


@pytest.mark.parametrize("W,H,K", [(9, 9, 3), (64, 33, 5), (257, 65, 1), (301, 200, 15), (1282, 70, 9)])
def test_convlt1_int16_matches_oracle(hip_ctx, oracle, W, H, K):
    """Generic odd kernel sizes, both input types, saturation included (weights up to +-32767 overflow int16 on purpose), and the gradient
    of the path as two calls: gx = (vt smoothing, hz derivative), gy swapped (canny_dete.cxx:237-241)."""
    from compv_amd import capi
    rng = np.random.default_rng(W * 7 + K)
    img8 = rng.integers(0, 256, (H, W), dtype=np.uint8)
    img16 = rng.integers(-32768, 32768, (H, W), dtype=np.int16)
    for scale in (3, 32767):
        vt = rng.integers(-scale, scale + 1, K).astype(np.int16)
        hz = rng.integers(-scale, scale + 1, K).astype(np.int16)
        rc, exp = oracle.convlt_8u(img8, vt, hz)
        assert rc == 0 and (hip_ctx.convlt1_i16(img8, vt, hz) == exp).all()
        rc, exp = oracle.convlt_16s(img16, vt, hz)
        assert rc == 0 and (hip_ctx.convlt1_i16(img16, vt, hz) == exp).all()
    if K == 3:
        gx_exp, gy_exp, _ = oracle.gradient(img8, 0)
        assert (hip_ctx.convlt1_i16(img8, [1, 2, 1], [-1, 0, 1]) == gx_exp).all()
        assert (hip_ctx.convlt1_i16(img8, [-1, 0, 1], [1, 2, 1]) == gy_exp).all()
    with pytest.raises(capi.CompvHipError) as e:
        hip_ctx.convlt1_i16(img8[:2], [1, 2, 1], [1, 2, 1])          # H < k (compv_math_convlt.h:100)
    assert e.value.code == capi.E_INVALID_PARAMETER
    with pytest.raises(capi.CompvHipError) as e:
        hip_ctx.convlt1_i16(img8, [1, 2], [1, 2])                    # even kernel size
    assert e.value.code == capi.E_INVALID_PARAMETER
