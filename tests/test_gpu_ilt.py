"""GPU parity tests (run with ``-m gpu`` on an MI355X), through the C ABI of libnlc_hip.so via the drop-in Python mirror:
the stand-alone ILT entry points (a9): Fourier / de Hoog / fixed Talbot / Stehfest reconstruction, backward kernels, query points.  Helpers and tolerances: tests/gpu_common.py.
"""

import glob
import os

import numpy as np
import pytest
import torch

from gpu_common import *  # noqa: F401,F403
from gpu_common import GOLD, TOL, T64, load_sd, build_model

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("d,S", [(5, 17), (3, 17), (6, 33), (4, 9), (5, 32)])
def test_ilt_fourier_vs_oracle(nlc, d, S):
    from oracle import ilt as oilt

    torch.manual_seed(d * 100 + S)
    N = 1537  # ragged: not a multiple of the block tile
    theta = (torch.rand(N, d, S, dtype=torch.float64) * 2 - 1) * np.pi
    phi = (torch.rand(N, d, S, dtype=torch.float64) * 2 - 1) * np.pi / 2 * 0.999
    t = torch.rand(N, dtype=torch.float64) * 2 + 0.05
    for opts in (None, dict(scale=3.0, alpha=1e-2)):
        ref = oilt.ilt_from_sphere(theta, phi, t, "fourier", opts)
        got = nlc.ilt_reconstruct(theta.cuda(), phi.cuda(), t.cuda(), "fourier", opts).cpu()
        scale = ref.abs().max()
        np.testing.assert_allclose(got.numpy() / scale, ref.numpy() / scale, rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("d,S,N", [(5, 17, 250_037), (3, 33, 200_003), (6, 9, 120_001)])
def test_ilt_fourier_many_tiles_per_block(nlc, d, S, N):
    """More tiles than the persistent grid has blocks: every block streams several tiles through the continuous
    cross-tile load pipeline (successor-tile prefetch, last whole tile without a successor, ragged tail tile)."""
    from oracle import ilt as oilt

    g = torch.Generator().manual_seed(N)
    theta = (torch.rand(N, d, S, dtype=torch.float64, generator=g) * 2 - 1) * np.pi
    phi = (torch.rand(N, d, S, dtype=torch.float64, generator=g) * 2 - 1) * np.pi / 2 * 0.999
    t = torch.rand(N, dtype=torch.float64, generator=g) * 2 + 0.05
    ref = oilt.ilt_from_sphere(theta, phi, t, "fourier", None)
    got = nlc.ilt_reconstruct(theta.cuda(), phi.cuda(), t.cuda(), "fourier", None).cpu()
    scale = ref.abs().max()
    np.testing.assert_allclose(got.numpy() / scale, ref.numpy() / scale, rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("d,S,N", [(5, 17, 1537), (3, 33, 700), (6, 9, 2049), (2, 17, 1), (5, 17, 90_001), (1, 3, 1),
                                   (1, 2, 2)])
def test_ilt_fourier_backward_vs_autograd_of_oracle(nlc, d, S, N):
    """nlc_ilt_reconstruct_backward against torch autograd through the CPU restatement (float64)."""
    from oracle import ilt as oilt

    g = torch.Generator().manual_seed(7 * N + S)
    theta = ((torch.rand(N, d, S, dtype=torch.float64, generator=g) * 2 - 1) * np.pi).requires_grad_()
    phi = ((torch.rand(N, d, S, dtype=torch.float64, generator=g) * 2 - 1) * np.pi / 2 * 0.99).requires_grad_()
    t = torch.rand(N, dtype=torch.float64, generator=g) * 2 + 0.05
    gx = torch.randn(N, d, dtype=torch.float64, generator=g)
    for opts in (None, dict(scale=3.0, alpha=1e-2)):
        ref = oilt.ilt_from_sphere(theta, phi, t, "fourier", opts)
        rt, rp = torch.autograd.grad(ref, (theta, phi), gx)
        th_d = theta.detach().cuda().requires_grad_()
        ph_d = phi.detach().cuda().requires_grad_()
        got = nlc.ilt_reconstruct(th_d, ph_d, t.cuda(), "fourier", opts)
        np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-9,
                                   atol=1e-11 * float(ref.detach().abs().max()))
        gt, gp = torch.autograd.grad(got, (th_d, ph_d), gx.cuda())
        for a, b in ((gt, rt), (gp, rp)):
            sc = float(b.abs().max())
            np.testing.assert_allclose(a.cpu().numpy() / sc, b.numpy() / sc, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("algo,S", [("fixed_tablot", 11), ("fixed_tablot", 24), ("stehfest", 8), ("stehfest", 16)])
def test_ilt_linear_backward_vs_autograd_of_oracle(nlc, algo, S):
    """Round 3: HIP backward of the two linear ILT algorithms (ilt_linear_bwd_kernel behind the same autograd Function as the
    Fourier one) against torch autograd through the CPU restatement; ragged N."""
    from oracle import ilt as oilt

    N, d = 777, 5
    g = torch.Generator().manual_seed(S)
    theta = ((torch.rand(N, d, S, dtype=torch.float64, generator=g) * 2 - 1) * np.pi).requires_grad_()
    phi = ((torch.rand(N, d, S, dtype=torch.float64, generator=g) * 2 - 1) * np.pi / 2 * 0.9).requires_grad_()
    t = torch.rand(N, dtype=torch.float64, generator=g) * 2 + 0.05
    gx = torch.randn(N, d, dtype=torch.float64, generator=g)
    ref = oilt.ilt_from_sphere(theta, phi, t, algo)
    rt, rp = torch.autograd.grad(ref, (theta, phi), gx)
    th_d, ph_d = theta.detach().cuda().requires_grad_(), phi.detach().cuda().requires_grad_()
    got = nlc.ilt_reconstruct(th_d, ph_d, t.cuda(), algo)
    assert got.requires_grad
    sc = float(ref.detach().abs().max())
    np.testing.assert_allclose(got.detach().cpu().numpy() / sc, ref.detach().numpy() / sc, rtol=1e-9, atol=1e-11)
    gt, gp = torch.autograd.grad(got, (th_d, ph_d), gx.cuda())
    for a, b in ((gt, rt), (gp, rp)):
        sc = float(b.abs().max())
        np.testing.assert_allclose(a.cpu().numpy() / sc, b.numpy() / sc, rtol=1e-9, atol=1e-12)


def test_laplace_reconstruct_trains_rep_func_through_hip_ilt(nlc):
    """Gradients reach the representation function's weights AND the latent p through laplace_reconstruct
    (the training path of w_nl.py:137-144), equal to autograd through the CPU restatement."""
    from oracle import ilt as oilt

    torch.manual_seed(3)
    B, P, d, S = 37, 7, 5, 17
    lin = torch.nn.Linear(2 * S + P, 2 * d * S).double()

    def make_rep(mod):
        def rep(i):
            out = mod(i.reshape(-1, 2 * S + P)).view(-1, 2 * d, S)
            return torch.tanh(out[:, :d, :]) * np.pi, torch.tanh(out[:, d:, :]) * np.pi / 2
        return rep

    p = torch.randn(B, P, dtype=torch.float64)
    t = torch.tensor([0.1, 0.25, 0.7], dtype=torch.float64)
    p_ref = p.clone().requires_grad_()
    ref = oilt.laplace_reconstruct(make_rep(lin), p_ref, t, recon_dim=d, ilt_algorithm="fourier",
                                   ilt_reconstruction_terms=S)
    w = torch.randn_like(ref)
    (ref * w).sum().backward()
    ref_grads = [lin.weight.grad.clone(), lin.bias.grad.clone(), p_ref.grad.clone()]
    lin_d = torch.nn.Linear(2 * S + P, 2 * d * S).double().cuda()
    lin_d.load_state_dict(lin.state_dict())
    p_d = p.cuda().requires_grad_()
    got = nlc.laplace_reconstruct(make_rep(lin_d), p_d, t.cuda(), recon_dim=d, ilt_algorithm="fourier",
                                  ilt_reconstruction_terms=S)
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-9, atol=1e-11)
    (got * w.cuda()).sum().backward()
    for a, b in zip((lin_d.weight.grad, lin_d.bias.grad, p_d.grad), ref_grads):
        sc = float(b.abs().max())
        np.testing.assert_allclose(a.cpu().numpy() / sc, b.numpy() / sc, rtol=1e-8, atol=1e-11)
    # (the two linear algorithms have a HIP backward as well since round 3: test_ilt_linear_backward_vs_autograd_of_oracle)
    x = nlc.ilt_reconstruct(torch.zeros(2, 1, 16, dtype=torch.float64, device="cuda", requires_grad=True),
                            torch.zeros(2, 1, 16, dtype=torch.float64, device="cuda"),
                            torch.full((2,), 0.1, dtype=torch.float64, device="cuda"), "stehfest")
    assert x.requires_grad


def test_ilt_fourier_full_bench_size_vs_oracle(nlc):
    """The stand-alone kernel at the bench's N = K*T = 655 360 points (d = 5, S = 17), forward and backward, against the
    CPU restatement on the same inputs (forward on every point; the gradient check on a checksum <g, x> = sum g x)."""
    from oracle import ilt as oilt

    N, d, S = 16384 * 40, 5, 17
    g = torch.Generator().manual_seed(99)
    theta = (torch.rand(N, d, S, dtype=torch.float64, generator=g) * 2 - 1) * np.pi
    phi = (torch.rand(N, d, S, dtype=torch.float64, generator=g) * 2 - 1) * np.pi / 2 * 0.99
    t = torch.full((N,), 0.125, dtype=torch.float64)
    gx = torch.randn(N, d, dtype=torch.float64, generator=g)
    ref = oilt.ilt_from_sphere(theta, phi, t, "fourier", None)
    th_d, ph_d = theta.cuda().requires_grad_(), phi.cuda().requires_grad_()
    got = nlc.ilt_reconstruct(th_d, ph_d, t.cuda(), "fourier", None)
    scale = float(ref.abs().max())
    np.testing.assert_allclose(got.detach().cpu().numpy() / scale, ref.numpy() / scale, rtol=1e-9, atol=1e-11)
    gt, gp = torch.autograd.grad(got, (th_d, ph_d), gx.cuda())
    # directional derivative of <gx, x> along a random direction, by central differences of the ORACLE
    dth = torch.randn(theta.shape, dtype=torch.float64, generator=g)
    dph = torch.randn(phi.shape, dtype=torch.float64, generator=g) * 0.1
    eps = 1e-6
    fp = (oilt.ilt_from_sphere(theta + eps * dth, phi + eps * dph, t, "fourier", None) * gx).sum()
    fm = (oilt.ilt_from_sphere(theta - eps * dth, phi - eps * dph, t, "fourier", None) * gx).sum()
    fd = float((fp - fm) / (2 * eps))
    an = float((gt.cpu() * dth).sum() + (gp.cpu() * dph).sum())
    assert abs(fd - an) <= 1e-6 * max(abs(fd), abs(an), 1.0), (fd, an)


def test_ilt_empty_and_single(nlc):
    z = nlc.ilt_reconstruct(torch.zeros(0, 5, 17).double().cuda(), torch.zeros(0, 5, 17).double().cuda(),
                            torch.zeros(0).double().cuda())
    assert z.shape == (0, 5)
    one = nlc.ilt_reconstruct(torch.zeros(1, 1, 17).double().cuda(), torch.zeros(1, 1, 17).double().cuda(),
                              torch.full((1,), 0.125).double().cuda())
    assert one.shape == (1, 1) and torch.isfinite(one).all()


@pytest.mark.parametrize("S", [33, 17, 9, 3, 5, 13, 21, 27, 31])
def test_ilt_dehoog_vs_oracle(nlc, S):
    from oracle import ilt as oilt

    torch.manual_seed(S)
    N, d = 700, 5
    # smooth F(s) (a rational transform sampled at the query points + small noise) keeps the QD table
    # well conditioned; the oracle and the kernel follow the same mpmath recurrences
    t = torch.rand(N, dtype=torch.float64) * 2 + 0.05
    alpha, tol, scale = oilt.ilt_options("dehoog")
    sr, si, _, _ = oilt.query_points(t, S, alpha, tol, scale)
    s = torch.complex(sr, si).unsqueeze(1)
    a = (torch.rand(N, d, 1, dtype=torch.float64) + 0.5)
    w = (torch.rand(N, d, 1, dtype=torch.float64) * 3 + 0.5)
    F = (s + a) / ((s + a) ** 2 + w**2)
    theta, phi = oilt.complex_to_sphere(F.real, F.imag)
    ref = oilt.ilt_from_sphere(theta, phi, t, "dehoog")
    got = nlc.ilt_reconstruct(theta.cuda(), phi.cuda(), t.cuda(), "dehoog").cpu()
    exact = torch.exp(-a.squeeze(-1) * t.view(-1, 1)) * torch.cos(w.squeeze(-1) * t.view(-1, 1))
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-6, atol=1e-8)
    if S == 33:
        np.testing.assert_allclose(got.numpy(), exact.numpy(), rtol=1e-5, atol=1e-6)


def test_ilt_known_answers_golden(nlc):
    """G4: analytic pairs / mpmath de Hoog, F sampled at the query points -> sphere -> HIP ILT."""
    from oracle import ilt as oilt

    g = np.load(f"{GOLD}/g4_ilt_known.npz")
    ts = T64(g["ts"])
    for name in ("exp_decay", "cosine", "sine_damped", "ramp"):
        th, ph = oilt.complex_to_sphere(T64(g[f"{name}_dehoog33_Fre"]), T64(g[f"{name}_dehoog33_Fim"]))
        got = nlc.ilt_reconstruct(th.unsqueeze(1).cuda(), ph.unsqueeze(1).cuda(), ts.cuda(), "dehoog").cpu()
        np.testing.assert_allclose(got.numpy()[:, 0], g[f"{name}_mp_dehoog"], rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(got.numpy()[:, 0], g[f"{name}_exact"], rtol=1e-6, atol=1e-7)


def test_rep_inputs_vs_oracle(nlc):
    from oracle import ilt as oilt

    torch.manual_seed(3)
    p = torch.randn(37, 7, dtype=torch.float64)
    for t in (torch.rand(37, 3, dtype=torch.float64) + 0.05, torch.rand(4, dtype=torch.float64) + 0.05):
        for algo, S in (("fourier", 17), ("dehoog", 33)):
            ref, _ = oilt.rep_func_inputs(p, t, S, algo)
            got, _ = nlc.rep_func_inputs(p.cuda(), t.cuda(), S, algo)
            np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=1e-12, atol=1e-13)


def test_laplace_reconstruct_generic_rep_func(nlc):
    """Full torchlaplace-style call with an arbitrary torch representation function."""
    from oracle import ilt as oilt

    torch.manual_seed(5)
    d, S, P, B = 3, 17, 6, 50
    lin = torch.nn.Linear(2 * S + P, 2 * d * S).double()

    def rep(i):
        out = lin.to(i.device)(i.view(-1, 2 * S + P)).view(-1, 2 * d, S)
        return torch.tanh(out[:, :d]) * torch.pi, torch.tanh(out[:, d:]) * torch.pi / 2

    p = torch.randn(B, P, dtype=torch.float64)
    t = torch.rand(B, 4, dtype=torch.float64) + 0.1
    with torch.no_grad():
        ref = oilt.laplace_reconstruct(rep, p, t, recon_dim=d, ilt_reconstruction_terms=S)
        got = nlc.laplace_reconstruct(rep, p.cuda(), t.cuda(), recon_dim=d, ilt_reconstruction_terms=S).cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), **TOL)


@pytest.mark.parametrize("algo,S", [("fixed_tablot", 17), ("fixed_tablot", 33), ("fixed_tablot", 8), ("stehfest", 16), ("stehfest", 12)])
def test_ilt_linear_algorithms_vs_oracle(nlc, algo, S):
    """fixed_tablot / stehfest (the other closed-form values of the reference's nl_ilt_algorithm knob, config.py:36):
    HIP rep-func inputs (query points on the algorithm's own contour, sphere projection) and HIP reconstruction vs the
    oracle's restatement of mpmath's FixedTalbot / Stehfest, and a full laplace_reconstruct through a torch rep func."""
    from oracle import ilt as oilt

    torch.manual_seed(S)
    N, d, P = 333, 3, 5
    t = torch.rand(N, dtype=torch.float64) * 2 + 0.05
    p = torch.randn(N, P, dtype=torch.float64)
    ref_in, _ = oilt.rep_func_inputs(p, t.view(N, 1), S, algo)
    got_in, _ = nlc.laplace.rep_func_inputs(p.cuda(), t.view(N, 1).cuda(), S, algo)
    np.testing.assert_allclose(got_in.cpu().numpy(), ref_in.numpy(), rtol=1e-12, atol=1e-12)
    theta = (torch.rand(N, d, S, dtype=torch.float64) * 2 - 1) * np.pi
    phi = (torch.rand(N, d, S, dtype=torch.float64) * 2 - 1) * np.pi / 2 * 0.9
    ref = oilt.ilt_from_sphere(theta, phi, t, algo)
    got = nlc.ilt_reconstruct(theta.cuda(), phi.cuda(), t.cuda(), algo).cpu()
    scale = float(ref.abs().max())
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-9, atol=1e-11 * scale)
    lin = torch.nn.Linear(2 * S + P, 2 * d * S).double()

    def rep(i):
        o = lin(i.reshape(-1, i.shape[-1])).view(-1, 2 * d, S)
        return torch.tanh(o[:, :d]) * np.pi, torch.tanh(o[:, d:]) * np.pi / 2

    with torch.no_grad():
        ref2 = oilt.laplace_reconstruct(rep, p, t.view(N, 1), recon_dim=d, ilt_algorithm=algo, ilt_reconstruction_terms=S)
        lin = lin.cuda()
        got2 = nlc.laplace_reconstruct(rep, p.cuda(), t.view(N, 1).cuda(), recon_dim=d, ilt_algorithm=algo,
                                       ilt_reconstruction_terms=S).cpu()
    scale2 = float(ref2.abs().max())
    np.testing.assert_allclose(got2.numpy(), ref2.numpy(), rtol=1e-8, atol=1e-10 * scale2)


@pytest.mark.parametrize("S,N,d", [(3, 5, 1), (5, 70, 3), (17, 203, 3), (33, 129, 5)])
def test_dehoog_autograd_path(nlc, monkeypatch, S, N, d):
    """Training through a de Hoog model (the reference trains through whatever ilt_algorithm is configured,
    train_utils.py:388-407): with grad-requiring theta / phi, ilt_reconstruct runs the same HIP forward kernel and, in
    backward, ilt_dehoog_bwd_kernel -- reverse mode through the quotient-difference table.  Gradients against autograd
    through the CPU restatement, on generic (random) Laplace terms: a rational F of low degree makes the table degenerate
    (e -> rounding noise), where no two roundings of the algorithm agree on a derivative."""
    from oracle import ilt as oilt

    g = torch.Generator().manual_seed(100 + S)
    t = torch.rand(N, dtype=torch.float64, generator=g) * 2 + 0.05
    theta = (torch.rand(N, d, S, dtype=torch.float64, generator=g) * 2 - 1) * 3.0
    phi = (torch.rand(N, d, S, dtype=torch.float64, generator=g) * 2 - 1) * 1.2
    w = torch.randn(N, d, dtype=torch.float64, generator=g)
    tho, pho = theta.clone().requires_grad_(), phi.clone().requires_grad_()
    ref_inplace = oilt.ilt_from_sphere(theta, phi, t, "dehoog")
    monkeypatch.setitem(oilt.LINE_INTEGRATE, "dehoog", dehoog_line_integrate_functional)
    ref = oilt.ilt_from_sphere(tho, pho, t, "dehoog")
    np.testing.assert_allclose(ref.detach().numpy(), ref_inplace.numpy(), rtol=1e-7, atol=1e-9)  # (vectorised vs per-entry complex ops)
    (ref * w).sum().backward()
    hip = nlc.ilt_reconstruct(theta.cuda(), phi.cuda(), t.cuda(), "dehoog")
    th, ph = theta.cuda().requires_grad_(), phi.cuda().requires_grad_()
    x = nlc.ilt_reconstruct(th, ph, t.cuda(), "dehoog")
    assert x.requires_grad
    np.testing.assert_array_equal(x.detach().cpu().numpy(), hip.cpu().numpy())
    # rows whose table is well conditioned in the oracle itself (a near-zero e somewhere amplifies rounding differences
    # of value AND gradient alike): judged by the forward agreement
    ok = ((x.detach().cpu() - ref.detach()).abs() <= 1e-9 * (1.0 + ref.detach().abs())).all(dim=1)
    assert ok.float().mean() > 0.9
    gth, gph = torch.autograd.grad(x, (th, ph), w.cuda())
    assert torch.isfinite(gth).all() and torch.isfinite(gph).all()
    for got_, ref_ in ((gth, tho.grad), (gph, pho.grad)):
        got_, ref_ = got_.cpu()[ok], ref_[ok]
        sc = ref_.abs().amax(dim=(1, 2), keepdim=True) + 1e-300
        np.testing.assert_allclose((got_ / sc).numpy(), (ref_ / sc).numpy(), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("S,N,d", [(33, 13200, 5), (17, 26300, 5)])
def test_dehoog_backward_when_a_wavefront_walks_more_than_one_block(nlc, S, N, d):
    """The backward kernel's grid is persistent (at most 1024 / 2048 slabs of tape): from 1025 / 2049 blocks of 64 rows on, a
    wavefront walks a SECOND block.  Round 5 found that path broken by a compiler placement (a spill reload behind a loop that
    leaves EXEC empty: kernels_dehoog_bwd.hip) -- no test had more than 1024 blocks.  The rows of a big batch must give the
    bits the same rows give in a small one (one block per wavefront), first, last and a ragged final block included."""
    g = torch.Generator(device="cuda").manual_seed(S)
    theta = (torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * 3.0
    phi = (torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * 1.2
    t = torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * 2 + 0.05
    w = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
    assert (N * d + 63) // 64 > (1024 if S > 17 else 2048) and (N * d) % 64 != 0

    def grads(sl):
        th, ph = theta[sl].clone().requires_grad_(), phi[sl].clone().requires_grad_()
        return torch.autograd.grad(nlc.ilt_reconstruct(th, ph, t[sl], "dehoog"), (th, ph), w[sl])

    big = grads(slice(0, N))
    assert torch.isfinite(big[0]).all() and torch.isfinite(big[1]).all()
    for lo, hi in ((0, 300), (N // 2 - 7, N // 2 + 250), (N - 333, N)):
        small = grads(slice(lo, hi))
        assert torch.equal(big[0][lo:hi], small[0]) and torch.equal(big[1][lo:hi], small[1]), (lo, hi)


def test_dehoog_backward_owns_its_scratch_across_streams(nlc):
    """The QD tape of ilt_dehoog_bwd_kernel is stream-ordered scratch of each call (no ctx state): backward calls issued
    back to back on two streams give the bits of a lone call."""
    g = torch.Generator(device="cuda").manual_seed(1)
    N, d, S = 700, 5, 33
    theta = ((torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * 3.0).requires_grad_()
    phi = ((torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * 1.2).requires_grad_()
    t = torch.full((N,), 0.125, dtype=torch.float64, device="cuda")
    gx = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
    ref = torch.autograd.grad(nlc.ilt_reconstruct(theta, phi, t, "dehoog"), (theta, phi), gx)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for _ in range(10):
        for st in (s1, s2):
            with torch.cuda.stream(st):
                outs.append(torch.autograd.grad(nlc.ilt_reconstruct(theta, phi, t, "dehoog"), (theta, phi), gx))
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o[0], ref[0]) and torch.equal(o[1], ref[1])


def test_ilt_single_point_wide(nlc):
    from oracle import ilt as oilt

    torch.manual_seed(5)
    for algo, S in (("fourier", 33), ("dehoog", 33), ("dehoog", 9)):
        theta = (torch.rand(1, 6, S, dtype=torch.float64) * 2 - 1) * np.pi * 0.3
        phi = (torch.rand(1, 6, S, dtype=torch.float64) * 2 - 1) * 0.4
        t = torch.tensor([0.7], dtype=torch.float64)
        ref = oilt.ilt_from_sphere(theta, phi, t, algo)
        got = nlc.ilt_reconstruct(theta.cuda(), phi.cuda(), t.cuda(), algo).cpu()
        if algo == "fourier":
            np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-9, atol=1e-9 * float(ref.abs().max()))
        else:
            # random (non-smooth) F makes the QD table ill-conditioned: both sides are finite and agree loosely
            assert torch.isfinite(got).all() == torch.isfinite(ref).all()


@pytest.mark.parametrize("S", list(range(3, 34)))
def test_ilt_fourier_row_kernels_every_term_count(nlc, S):
    """Round 6: the row-per-lane kernels of the Fourier ILT (direct global -> LDS tile loads; odd S <= 17: two tiles in flight per
    wavefront, odd S > 17: one; even S keeps the term-per-lane stream -- a 64-row tile's row-wise reads would conflict).  Forward at scale 2 (compile-time quarter turns) and at another
    scale (the per-term phase / weight table), and the backward, on a population that gives every wavefront several tiles AND a
    ragged last tile, with a per-row t -- against the oracle and autograd through it."""
    from oracle import ilt as oilt

    d, N = 3, 64 * 2048 // 3 + 29  # rows = N d: more tiles than the grid has wavefronts at S <= 17, not a multiple of 64
    g = torch.Generator().manual_seed(S)
    theta = (torch.rand(N, d, S, dtype=torch.float64, generator=g) * 2 - 1) * np.pi
    phi = (torch.rand(N, d, S, dtype=torch.float64, generator=g) * 2 - 1) * np.pi / 2 * 0.999
    t = torch.rand(N, dtype=torch.float64, generator=g) * 2 + 0.05
    for opts in (None, dict(scale=3.0, alpha=1e-2)):
        ref = oilt.ilt_from_sphere(theta, phi, t, "fourier", opts)
        got = nlc.ilt_reconstruct(theta.cuda(), phi.cuda(), t.cuda(), "fourier", opts).cpu()
        scale = ref.abs().max()
        np.testing.assert_allclose(got.numpy() / scale, ref.numpy() / scale, rtol=1e-9, atol=1e-11)
    # backward on a slice (autograd through the CPU oracle is the slow side)
    n = 64 * 40 // d + 7
    th = theta[:n].clone().requires_grad_()
    ph = (phi[:n] * 0.99).clone().requires_grad_()
    gx = torch.randn(n, d, dtype=torch.float64, generator=g)
    ref_gt, ref_gp = torch.autograd.grad(oilt.ilt_from_sphere(th, ph, t[:n], "fourier", None), (th, ph), gx)
    thc, phc = th.detach().cuda().requires_grad_(), ph.detach().cuda().requires_grad_()
    got_gt, got_gp = torch.autograd.grad(nlc.ilt_reconstruct(thc, phc, t[:n].cuda(), "fourier", None), (thc, phc), gx.cuda())
    for got_g, ref_g in ((got_gt, ref_gt), (got_gp, ref_gp)):
        sc = float(ref_g.abs().max())
        np.testing.assert_allclose(got_g.cpu().numpy() / sc, ref_g.numpy() / sc, rtol=1e-9, atol=1e-11)


def test_ilt_fourier_unaligned_and_even_inputs_keep_the_stream_kernel(nlc):
    """The row kernels need 16-byte aligned arrays and an odd term count; a view that starts 8 bytes into an allocation and an
    even term count take the term-per-lane stream instead -- same results either way."""
    from oracle import ilt as oilt

    d, S, N = 5, 17, 4099
    g = torch.Generator().manual_seed(3)
    flat_t = (torch.rand(N * d * S + 1, dtype=torch.float64, generator=g) * 2 - 1) * np.pi
    flat_p = (torch.rand(N * d * S + 1, dtype=torch.float64, generator=g) * 2 - 1) * np.pi / 2 * 0.999
    t = torch.rand(N, dtype=torch.float64, generator=g) * 2 + 0.05
    ft, fp = flat_t.cuda(), flat_p.cuda()
    theta_u, phi_u = ft[1:].view(N, d, S), fp[1:].view(N, d, S)  # data_ptr % 16 == 8
    assert theta_u.data_ptr() % 16 == 8 and theta_u.is_contiguous()
    ref = oilt.ilt_from_sphere(flat_t[1:].view(N, d, S), flat_p[1:].view(N, d, S), t, "fourier", None)
    got_u = nlc.ilt_reconstruct(theta_u, phi_u, t.cuda(), "fourier", None).cpu()
    got_a = nlc.ilt_reconstruct(theta_u.clone(), phi_u.clone(), t.cuda(), "fourier", None).cpu()  # fresh allocations: aligned
    scale = ref.abs().max()
    np.testing.assert_allclose(got_u.numpy() / scale, ref.numpy() / scale, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(got_a.numpy() / scale, ref.numpy() / scale, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(got_a.numpy(), got_u.numpy(), rtol=1e-12, atol=1e-13 * float(scale))
