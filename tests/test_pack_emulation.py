"""CPU: lane-by-lane emulation of the MFMA kernels' dataflow on the fragment-packed weights (csrc/nlc_pack.h),
against the oracle.  Covers the packing order, the layer-3 slot permutation, the ILT coefficient matrix and the
accumulator-register-as-next-B-fragment chaining for every (d, S) the reference envs use."""

import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import ilt as oilt
from oracle import nl_model as onl

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    out = tmp_path_factory.mktemp("packhost") / "libpack_host.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", str(out), os.path.join(HERE, "helpers", "pack_host.cpp")])
    return ctypes.CDLL(str(out))


def P(t):
    return ctypes.c_void_p(np.ascontiguousarray(t.numpy() if torch.is_tensor(t) else t, dtype=np.float64).ctypes.data)


def arr(t):
    return np.ascontiguousarray(t.numpy() if torch.is_tensor(t) else t, dtype=np.float64)


@pytest.mark.parametrize("env,S", [("oderl-cartpole", 17), ("oderl-pendulum", 17), ("oderl-acrobot", 17), ("oderl-cartpole", 33),
                                   ("oderl-acrobot", 33), ("oderl-pendulum", 9)])
def test_mlp_ilt_dataflow(lib, env, S):
    st = onl.ENV_STATS[env]
    d, nu = st["d"], st["nu"]
    sd = onl.make_synthetic_state_dict(1, d, nu, 128, S, st["state_std"], [1.0], tame=True)
    torch.manual_seed(d + S)
    p = torch.randn(16, d + 2, dtype=torch.float64)
    tn = 0.125
    pre = "laplace_rep_func.linear_tanh_stack."
    keep = [arr(sd[pre + k]) for k in ("0.weight", "0.bias", "2.weight", "2.bias", "4.weight", "4.bias")]
    alpha, tol, scale = oilt.ilt_options("fourier")
    out = np.zeros((16, d))
    pa = arr(p)
    lib.emu_mlp_ilt(*[ctypes.c_void_p(a.ctypes.data) for a in keep], d, S, 128, ctypes.c_void_p(pa.ctypes.data),
                    ctypes.c_double(tn), ctypes.c_double(alpha), ctypes.c_double(np.log(tol)), ctypes.c_double(scale),
                    ctypes.c_void_p(out.ctypes.data))
    ref = oilt.laplace_reconstruct(lambda i: onl.rep_func(sd, i, d, S), p, torch.full((16, 1), tn, dtype=torch.float64),
                                   recon_dim=d, ilt_reconstruction_terms=S).view(16, d)
    np.testing.assert_allclose(out, ref.numpy(), rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("nu,B", [(1, 4), (2, 4), (1, 5), (3, 3)])
def test_gru_chunked_dataflow(lib, nu, B):
    sd = onl.make_synthetic_state_dict(2, 4, nu, 128, 17)
    torch.manual_seed(nu * 10 + B)
    win = torch.randn(16, B, nu, dtype=torch.float64)
    pre = "action_encoder.gru."
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l1", "weight_hh_l1", "bias_ih_l1", "bias_hh_l1"]
    keep = [arr(sd[pre + n]) for n in names] + [arr(sd["action_encoder.linear_out.weight"]), arr(sd["action_encoder.linear_out.bias"])]
    out = np.zeros((16, 2))
    w = arr(win)
    lib.emu_gru(*[ctypes.c_void_p(a.ctypes.data) for a in keep], 64, nu, B, ctypes.c_void_p(w.ctypes.data),
                ctypes.c_void_p(out.ctypes.data))
    np.testing.assert_allclose(out, onl.gru_encoder(sd, win).numpy(), rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize("nu,B", [(1, 4), (2, 4), (1, 1), (3, 3)])
def test_gru_int8_sliced_dataflow(lib, nu, B):
    """csrc/kernels_gru_i8.hip's dataflow on the CPU: the 36-block weight stream of pack_gru_i8_stream (order, digit fragments,
    recombination factors, biases), the digit cut of the states, the i8 MFMA's operand / accumulator maps as
    tools/i8gemm_check.hip measured them, merged and level-by-level recombination -- against the oracle's float64 GRU."""
    sd = onl.make_synthetic_state_dict(2, 4, nu, 128, 17)
    torch.manual_seed(nu * 10 + B)
    win = torch.randn(16, B, nu, dtype=torch.float64)
    pre = "action_encoder.gru."
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l1", "weight_hh_l1", "bias_ih_l1", "bias_hh_l1"]
    keep = [arr(sd[pre + n]) for n in names] + [arr(sd["action_encoder.linear_out.weight"]), arr(sd["action_encoder.linear_out.bias"])]
    out = np.zeros((16, 2))
    w = arr(win)
    lib.emu_gru_i8(*[ctypes.c_void_p(a.ctypes.data) for a in keep], 64, nu, B, ctypes.c_void_p(w.ctypes.data),
                   ctypes.c_void_p(out.ctypes.data))
    np.testing.assert_allclose(out, onl.gru_encoder(sd, win).numpy(), rtol=1e-11, atol=1e-13)


def test_int8_digit_cut_and_row_exponents(lib):
    """nlc_pack.h: seven signed digits of rint(x 2^54) reassemble to the value rounded at 2^-55 (exactly for values on the grid), the
    top digit stays inside int8 for |x| <= 1, and a row's exponent is the smallest e with max |w| <= 2^e."""
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-1, 1, 4000), np.array([0.0, 1.0, -1.0, 0.5, -0.5, 2.0 ** -54, -(2.0 ** -54), 2.0 ** -55 * 1.5, 1 - 2.0 ** -53,
                                                            -1 + 2.0 ** -53, 2.0 ** -30, 1e-300, 0.999999, 2.0 ** -24 + 2.0 ** -54])])
    back = np.zeros_like(x)
    top = np.zeros(len(x), dtype=np.int32)
    lib.emu_i8_roundtrip(ctypes.c_void_p(x.ctypes.data), len(x), ctypes.c_void_p(back.ctypes.data), ctypes.c_void_p(top.ctypes.data))
    assert np.all(np.abs(back - x) <= 2.0 ** -55)
    grid = np.round(x * 2.0 ** 54) / 2.0 ** 54
    assert np.array_equal(back, grid)
    assert np.all(np.abs(top) <= 65)
    W = np.array([[0.0, 0.0, 0.0], [0.3, -0.9, 0.1], [1.0, 0.2, -0.5], [1.0000001, 0.0, 0.0], [2.0 ** -7, -(2.0 ** -9), 0.0], [3e4, 1.0, -2.0]])
    e = np.zeros(len(W), dtype=np.int32)
    lib.emu_i8_row_exponents(ctypes.c_void_p(W.ctypes.data), len(W), W.shape[1], ctypes.c_void_p(e.ctypes.data))
    assert e.tolist() == [0, 0, 0, 1, -7, 15]
    mx = np.abs(W).max(axis=1)
    assert np.all(mx <= 2.0 ** e) and np.all((mx == 0) | (mx > 2.0 ** (e - 1.0)))
