"""The fast-math element-wise functions of the kernels (`__expf`, `__sinf`, `__fdividef`, Abramowitz-Stegun erf) swept against
float64 libm over the input ranges the real checkpoint produces and well beyond: |x * alpha| up to 1e3 for
Snake with alpha in [0.05, 20], +-30 for the GEMM epilogue activations.  Run with -m gpu on an MI355X."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def lib():
    assert torch.cuda.is_available(), 'needs a GPU'
    from cv2amd import lib as L
    l = L.lib()
    l.cv2_dbg_act.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p]
    l.cv2_dbg_pre.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_float, C.c_void_p]
    return l


def _grid(lo, hi, n, seed):
    """dense uniform grid + random points + the neighbourhood of 0"""
    g = torch.Generator().manual_seed(seed)
    x = torch.cat([torch.linspace(lo, hi, n), (torch.rand(n, generator=g) * (hi - lo) + lo),
                   torch.linspace(-1e-3, 1e-3, 2001), torch.tensor([0.0, -0.0, lo, hi])])
    return x.float()


def _run_act(lib, x, act, slope=0.0):
    from cv2amd import lib as L
    xd = x.cuda().contiguous()
    out = torch.empty_like(xd)
    L.check(lib.cv2_dbg_act(xd.data_ptr(), out.data_ptr(), xd.numel(), act, slope, L.stream_ptr()))
    torch.cuda.synchronize()
    return out.cpu().double()


def _run_pre(lib, x, pre, alpha, slope=0.0):
    from cv2amd import lib as L
    xd = x.cuda().contiguous()
    out = torch.empty_like(xd)
    L.check(lib.cv2_dbg_pre(xd.data_ptr(), out.data_ptr(), xd.numel(), pre, alpha, slope, L.stream_ptr()))
    torch.cuda.synchronize()
    return out.cpu().double()


def test_gelu_silu_mish_against_libm(lib):
    """Epilogue activations of the flow GEMMs.  Bound: 2e-6 absolute + 2e-6 relative — two orders below the bf16 rounding (2^-9)
    of the value when it is stored as the next GEMM's operand."""
    x = _grid(-30.0, 30.0, 200001, 1)
    xd = x.double()
    ref = {1: 0.5 * xd * (1.0 + torch.special.erf(xd / math.sqrt(2.0))),
           2: xd * torch.sigmoid(xd),
           3: xd * torch.tanh(torch.nn.functional.softplus(xd, threshold=20.0))}
    worst = {}
    for act, r in ref.items():
        got = _run_act(lib, x, act)
        assert torch.isfinite(got).all()
        err = (got - r).abs()
        tol = 2e-6 + 2e-6 * r.abs()
        worst[act] = (err / tol).max().item()
        i = int((err / tol).argmax())
        assert (err <= tol).all(), f'act {act}: x={x[i].item():.6g} got {got[i].item():.9g} want {r[i].item():.9g}'
    # leaky ReLU is exact
    got = _run_act(lib, x, 4, 0.01)
    assert torch.equal(got.float(), torch.where(x > 0, x, x * 0.01))


@pytest.mark.parametrize('alpha', [0.05, 0.3, 1.0, 3.7, 20.0])
def test_snake_against_libm_up_to_1e3_radians(lib, alpha):
    """Snake pre-activation of the HiFT ResBlocks with the hardware sine.  v_sin_f32 reduces its argument in revolutions, so the
    absolute error of sin grows with |alpha x| * 2^-24; the Snake term divides sin^2 by alpha.  Bound: (4e-6 + |alpha x| * 2.4e-7) / alpha
    absolute — for the |alpha x| < 100 of the real checkpoint (HiFT ranges) below 3e-5 / alpha, against activations
    of order 1."""
    xmax = 1000.0 / alpha
    x = _grid(-xmax, xmax, 400001, 2)
    xd = x.double()
    a32 = float(np.float32(alpha))
    ref = xd + torch.sin(xd * a32) ** 2 / (a32 + 1e-9)
    got = _run_pre(lib, x, 1, alpha)
    assert torch.isfinite(got).all()
    err = (got - ref).abs()
    tol = (4e-6 + (xd * a32).abs() * 2.4e-7) / a32 + 1e-6 * ref.abs()
    i = int((err / tol).argmax())
    assert (err <= tol).all(), f'alpha {alpha}: x={x[i].item():.6g} (alpha x = {x[i].item() * alpha:.4g}) err {err[i].item():.3e} tol {tol[i].item():.3e}'
    # inside the range the real checkpoint reaches the error stays at round-off
    m = (xd * a32).abs() < 100.0
    assert (err[m] * a32).max().item() < 3e-5


def test_leaky_relu_pre_activation_is_exact(lib):
    x = _grid(-50.0, 50.0, 10001, 3)
    got = _run_pre(lib, x, 2, 1.0, 0.1)
    assert torch.equal(got.float(), torch.where(x > 0, x, x * 0.1))
