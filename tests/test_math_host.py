"""CPU: the bounded-range FP64 math used inside the HIP kernels (csrc/nlc_math.h), built with g++,
against numpy/libm on dense grids.  The device build differs only in v_rcp_f64-based division."""

import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    out = tmp_path_factory.mktemp("mathhost") / "libmath_host.so"
    subprocess.check_call(
        ["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(out), os.path.join(HERE, "helpers", "math_host.cpp")]
    )
    return ctypes.CDLL(str(out))


def call(lib, name, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    f = getattr(lib, name)
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
    f(x.ctypes.data, y.ctypes.data, x.size)
    return y


def ulp_err(y, ref):
    return np.abs(y - ref) / np.spacing(np.abs(ref) + 1e-300)


def test_tanh(lib):
    x = np.concatenate([np.linspace(-40, 40, 400001), np.logspace(-300, 1, 20001), -np.logspace(-12, 1, 2001), [0.0]])
    assert ulp_err(call(lib, "nlc_t_tanh", x), np.tanh(x)).max() <= 4


def test_tanh_pair_fast(lib):
    """Hidden-layer activation of the rollout kernels (round 3): absolute error <= 1.5e-13 everywhere (degree-9 Horner exp,
    one-constant reduction), exact sign symmetry, 1 within an ulp at saturation, finite for huge arguments."""
    x = np.concatenate([np.linspace(-40, 40, 400001), np.logspace(-300, 3, 20001), -np.logspace(-12, 3, 2001), [0.0, 1e300, -1e300]])
    y = call(lib, "nlc_t_tanh_pair_fast", x)
    assert np.abs(y - np.tanh(x)).max() <= 1.5e-13
    assert np.all(np.abs(y) <= 1.0 + 3e-16) and np.all(np.isfinite(y))
    assert np.array_equal(call(lib, "nlc_t_tanh_pair_fast", -x), -y)


def test_sigmoid(lib):
    x = np.concatenate([np.linspace(-800, 800, 400001), np.logspace(-12, 2, 2001)])
    ref = np.where(x >= 0, 1 / (1 + np.exp(-np.abs(x))), np.exp(-np.abs(x)) / (1 + np.exp(-np.abs(x))))
    y = call(lib, "nlc_t_sigmoid", x)
    ok = np.abs(x) < 700
    assert ulp_err(y[ok], ref[ok]).max() <= 4
    assert np.all(np.isfinite(y)) and np.all(y >= 0) and np.all(y <= 1)


def test_exp(lib):
    x = np.linspace(-700, 700, 400001)
    assert ulp_err(call(lib, "nlc_t_exp", x), np.exp(x)).max() <= 2


def test_sincos(lib):
    x = np.concatenate([np.linspace(-2 * np.pi, 2 * np.pi, 800001), np.logspace(-300, 0, 3001)])
    s, c = call(lib, "nlc_t_sin", x), call(lib, "nlc_t_cos", x)
    # absolute accuracy 1e-16-ish everywhere (near the zeros relative error is limited by the 2-term pi/2)
    assert np.abs(s - np.sin(x)).max() < 2.5e-16
    assert np.abs(c - np.cos(x)).max() < 2.5e-16
    small = np.abs(x) < 0.7
    assert ulp_err(s[small], np.sin(x[small])).max() <= 2


def test_tan(lib):
    x = np.concatenate([np.linspace(0, np.pi / 2, 400001), np.logspace(-300, -1, 2001)])
    y, ref = call(lib, "nlc_t_tan", x), np.tan(x)
    ok = x < 1.5707
    assert ulp_err(y[ok], ref[ok]).max() <= 4
    assert np.all(np.abs(y[~ok] - ref[~ok]) <= 1e-11 * np.abs(ref[~ok]))


def test_tan_pi4_plus(lib):
    x = np.concatenate([np.linspace(0, np.pi / 2, 400001), np.logspace(-18, -1, 2001)])
    ref = np.tan(x.astype(np.longdouble)).astype(np.float64)  # tan of that very double, 80-bit
    y = call(lib, "nlc_t_tan_pi4", x)
    # (cos a + sin a)/(cos a - sin a) amplifies the 1e-16 absolute rounding of sin/cos near the zero and the pole:
    # absolute error O(1e-16 (1 + tan^2)), the same conditioning tan has w.r.t. a one-ulp change of its argument
    ok = x < 1.57
    assert np.all(np.abs(y[ok] - ref[ok]) <= 4e-16 * (1.0 + ref[ok] ** 2))
    assert np.isfinite(y).all() and y.min() > -3e-16 and y[400000] > 1e15


def test_cos_quadrant(lib):
    x = np.linspace(-2 * np.pi, 2 * np.pi, 700001)
    j0 = (np.arange(x.size) % 7) - 3
    pi_l = np.longdouble("3.14159265358979323846264338327950288")
    ref = np.cos(x.astype(np.longdouble) + j0 * pi_l / 2).astype(np.float64)
    assert np.abs(call(lib, "nlc_t_cosq", x) - ref).max() < 3e-16


def test_table_driven_tanh_sigmoid(lib):
    x = np.concatenate([np.linspace(-40, 40, 400001), np.logspace(-300, 1, 20001), -np.logspace(-12, 1, 2001), [0.0]])
    # the rounded table entry leaves an ABSOLUTE error of ~1e-16 on e^{-2a}-1, so for 0.003 < |x| < 0.03 the
    # result is exact to 2e-16 absolute rather than to a few ulp (it feeds matmuls: same as any rounding of O(1) data)
    y, ref = call(lib, "nlc_t_tanh_t", x), np.tanh(x)
    assert np.all(np.abs(y - ref) <= 2.5e-16 + 4 * np.spacing(np.abs(ref)))
    assert ulp_err(y[np.abs(x) > 0.05], ref[np.abs(x) > 0.05]).max() <= 6
    assert ulp_err(y[np.abs(x) < 0.002], ref[np.abs(x) < 0.002]).max() <= 4
    xs = np.concatenate([np.linspace(-800, 800, 400001), np.logspace(-12, 2, 2001)])
    ref = np.where(xs >= 0, 1 / (1 + np.exp(-np.abs(xs))), np.exp(-np.abs(xs)) / (1 + np.exp(-np.abs(xs))))
    y = call(lib, "nlc_t_sigmoid_t", xs)
    ok = np.abs(xs) < 700
    assert ulp_err(y[ok], ref[ok]).max() <= 4
    assert np.all(np.isfinite(y)) and np.all(y >= 0) and np.all(y <= 1)


def test_ilt_short_forms(lib):
    """The instruction-count-trimmed trig of the stand-alone Fourier ILT kernel: absolute accuracy ~2e-16 for the
    cosine (one reduction by pi, quarter-turn parity in the reduction), tan as num/den within a few ulp away from the
    pole and tracking the conditioning of the argument's own rounding next to it."""
    x = np.concatenate([np.linspace(-7.0, 7.0, 800001), np.linspace(-1e3, 1e3, 20001), [0.0, np.pi, -np.pi]])
    mm = (np.arange(x.size) & 1).astype(np.float64)
    y = call(lib, "nlc_t_cos_mpio2", x)
    xl = x.astype(np.longdouble)
    ref = np.where(mm == 0, np.cos(xl), -np.sin(xl))  # cos(x + pi/2) = -sin(x)
    bound = 3e-16 + 1.2e-16 * np.abs(x) / np.pi  # 2-term pi/2: the reduction error grows with the quotient
    assert np.all(np.abs(y - ref.astype(np.float64)) <= bound)
    ys = call(lib, "nlc_t_sin_mpio2", x)
    ref_s = np.where(mm == 0, np.sin(xl), np.cos(xl))  # sin(x + pi/2) = cos(x)
    assert np.all(np.abs(ys - ref_s.astype(np.float64)) <= bound)
    xt = np.linspace(0.0, np.pi / 2, 400001)[:-1]
    t = call(lib, "nlc_t_tan_short", xt)
    ref_t = np.tan(xt)
    # relative error bounded by a few ulp plus the pole's conditioning: d tan / tan = dx (1 + tan^2)/tan
    cond = (1 + ref_t ** 2) / np.maximum(ref_t, 1e-300) * 2.3e-16
    assert np.all(np.abs(t - ref_t) <= (6e-16 + cond) * np.maximum(np.abs(ref_t), 1.0))


def test_ilt_row_forms(lib):
    """Round 6, the row-per-lane Fourier ILT kernel: tan(pi/4 + a) = num/den from the Cephes rational (|a| <= pi/4) within a few
    ulp away from the pole and tracking the argument's own conditioning next to it; cos(x + m pi/2) by one reduction by pi with
    the cosine (m = 0) or sine (m = 1) polynomial: absolute accuracy ~2e-16."""
    a = np.linspace(-np.pi / 4, np.pi / 4, 400001)[1:-1]
    t = call(lib, "nlc_t_tan_rat", a)
    al = a.astype(np.longdouble)
    ref = np.tan(al + np.longdouble(np.pi) / 4 + np.longdouble(6.123233995736766e-17) / 2).astype(np.float64)  # (pi/4 to ~1e-33)
    cond = (1 + ref ** 2) / np.maximum(ref, 1e-300) * 1.2e-16  # the rounding of a itself
    assert np.all(np.abs(t - ref) <= (6e-16 + cond) * np.maximum(np.abs(ref), 1.0))
    x = np.concatenate([np.linspace(-7.0, 7.0, 800001), np.linspace(-1e3, 1e3, 20001), [0.0, np.pi, -np.pi]])
    mm = (np.arange(x.size) & 1).astype(np.float64)
    y = call(lib, "nlc_t_cos_kpio2", x)
    xl = x.astype(np.longdouble)
    want = np.where(mm == 0, np.cos(xl), -np.sin(xl)).astype(np.float64)
    bound = 3e-16 + 1.2e-16 * np.abs(x) / np.pi
    assert np.all(np.abs(y - want) <= bound)


def test_sincos_reduced(lib):
    """Backward of the row-per-lane Fourier ILT kernel: sin x and cos x from ONE reduction by pi, absolute accuracy ~2e-16."""
    x = np.ascontiguousarray(np.concatenate([np.linspace(-7.0, 7.0, 400001), np.linspace(-1e3, 1e3, 20001), [0.0, np.pi, -np.pi]]))
    sn, cs = np.empty_like(x), np.empty_like(x)
    f = lib.nlc_t_sincos_reduced
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
    f(x.ctypes.data, sn.ctypes.data, cs.ctypes.data, x.size)
    xl = x.astype(np.longdouble)
    bound = 3e-16 + 1.2e-16 * np.abs(x) / np.pi
    assert np.all(np.abs(sn - np.sin(xl).astype(np.float64)) <= bound)
    assert np.all(np.abs(cs - np.cos(xl).astype(np.float64)) <= bound)
