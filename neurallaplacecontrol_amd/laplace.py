"""Drop-in for ``torchlaplace.laplace_reconstruct`` (reference call sites ``w_nl.py:137-144``,
``w_latent_ode.py:88-94``) on MI355X.

Stages, all on the device:
  1. HIP ``nlc_ilt_rep_inputs``  -- contour evaluation s_k(t), Riemann-sphere projection, concat with p
  2. the caller's ``laplace_rep_func`` (any torch callable; PyTorch-ROCm)
  3. HIP ``nlc_ilt_reconstruct`` -- sphere->complex and the Fourier / de Hoog line integral
     (Fourier: differentiable, ``nlc_ilt_reconstruct_backward`` is its autograd backward, so a representation
     function trains through it as through torchlaplace)

**Parity unpinned vs upstream torchlaplace** (the package is absent offline): defaults ``alpha``,
``tol``, ``scale`` and the ``[theta_s | phi_s | p]`` input order follow SURVEY.md §A.3 and can be
overridden through ``options``.
"""

import ctypes as C

import torch

from . import _lib

_default_ctx = {}


def default_ctx(index):
    """Shared ctx for stateless ILT calls on device `index`."""
    if index not in _default_ctx:
        _default_ctx[index] = _lib.Ctx(index)
    return _default_ctx[index]


def compute_device(*tensors):
    """CUDA device to compute on: the first CUDA tensor's, else torch's current CUDA device."""
    if not torch.cuda.is_available():
        raise RuntimeError("neurallaplacecontrol_amd needs an AMD MI355X (no HIP device visible); there is no CPU path")
    for t in tensors:
        if torch.is_tensor(t) and t.is_cuda:
            return t.device
    return torch.device("cuda", torch.cuda.current_device())


def _prep(t, dev):
    return t.detach().to(device=dev, dtype=torch.float64).contiguous()


def rep_func_inputs(p, t, ilt_reconstruction_terms, ilt_algorithm="fourier", options=None, ctx=None):
    """(B, Tt, 2S+P) rows ``[theta_s | phi_s | p]`` and the (B, Tt) time grid, on the device."""
    dev = compute_device(p, t)
    desc = _lib.ilt_desc(ilt_algorithm, ilt_reconstruction_terms, options)
    p_d = _prep(p, dev)
    t_d = _prep(torch.as_tensor(t), dev)
    if p_d.dim() != 2:
        raise ValueError("p must be (batch, latent_dim)")
    B, P = p_d.shape
    if t_d.dim() == 0:
        t_d = t_d.view(1)
    if t_d.dim() == 1:
        batched, Tt = 0, t_d.shape[0]
    elif t_d.dim() == 2 and t_d.shape[0] == B:
        batched, Tt = 1, t_d.shape[1]
    else:
        raise ValueError("t must be (time,) or (batch, time)")
    S = desc.terms
    out = torch.empty((B, Tt, 2 * S + P), dtype=torch.float64, device=dev)
    ctx = ctx or default_ctx(dev.index)
    with torch.cuda.device(dev):
        ctx.use_torch_stream()
        ctx.check(
            ctx.lib.nlc_ilt_rep_inputs(
                ctx.h, C.byref(desc), _lib.ptr(p_d), _lib.ptr(t_d), batched, B, Tt, P, _lib.ptr(out)
            )
        )
    t2 = t_d if batched else t_d.view(1, -1).expand(B, -1)
    return out, t2


def _ilt_forward(theta_d, phi_d, t_d, desc, ctx):
    N, d, _ = theta_d.shape
    x = torch.empty((N, d), dtype=torch.float64, device=theta_d.device)
    with torch.cuda.device(theta_d.device):
        ctx.use_torch_stream()
        ctx.check(
            ctx.lib.nlc_ilt_reconstruct(
                ctx.h, C.byref(desc), _lib.ptr(theta_d), _lib.ptr(phi_d), _lib.ptr(t_d), N, d, _lib.ptr(x)
            )
        )
    return x


class _IltFn(torch.autograd.Function):
    """``nlc_ilt_reconstruct`` with ``nlc_ilt_reconstruct_backward`` as its vector-Jacobian product, so the
    representation function trains through the HIP line integral as it does through torchlaplace
    (``train_utils.py:388-407`` -> ``w_nl.py:137-144``).  No gradient with respect to t."""

    @staticmethod
    def forward(fctx, theta, phi, t_d, desc, ctx):
        theta_d, phi_d = theta.detach().contiguous(), phi.detach().contiguous()
        fctx.save_for_backward(theta_d, phi_d, t_d)
        fctx.desc, fctx.ctx = desc, ctx
        return _ilt_forward(theta_d, phi_d, t_d, desc, ctx)

    @staticmethod
    def backward(fctx, grad_x):
        theta_d, phi_d, t_d = fctx.saved_tensors
        desc, ctx = fctx.desc, fctx.ctx
        N, d, _ = theta_d.shape
        g = grad_x.detach().to(dtype=torch.float64).contiguous()
        g_theta, g_phi = torch.empty_like(theta_d), torch.empty_like(phi_d)
        with torch.cuda.device(theta_d.device):
            ctx.use_torch_stream()
            ctx.check(
                ctx.lib.nlc_ilt_reconstruct_backward(
                    ctx.h, C.byref(desc), _lib.ptr(theta_d), _lib.ptr(phi_d), _lib.ptr(t_d), _lib.ptr(g), N, d,
                    _lib.ptr(g_theta), _lib.ptr(g_phi),
                )
            )
        return g_theta, g_phi, None, None, None


def ilt_reconstruct(theta, phi, t, ilt_algorithm="fourier", options=None, ctx=None):
    """theta, phi: (N, d, S) representation-function outputs, t: (N,) -> x (N, d).

    Differentiable with respect to theta / phi for every algorithm: HIP forward and HIP backward kernels behind one
    autograd Function (de Hoog: reverse mode through the quotient-difference table, ``kernels_dehoog_bwd.hip``)."""
    dev = compute_device(theta, phi, t)
    needs_grad = torch.is_grad_enabled() and (
        (torch.is_tensor(theta) and theta.requires_grad) or (torch.is_tensor(phi) and phi.requires_grad)
    )
    t_d = _prep(t, dev).reshape(-1)
    if theta.dim() != 3 or theta.shape != phi.shape:
        raise ValueError("theta and phi must both be (N, d, S)")
    N, d, S = theta.shape
    if t_d.numel() != N:
        raise ValueError("t must have one entry per row of theta/phi")
    desc = _lib.ilt_desc(ilt_algorithm, S, options)
    ctx = ctx or default_ctx(dev.index)
    if needs_grad:
        # HIP forward + HIP backward kernels behind one autograd Function
        return _IltFn.apply(
            theta.to(device=dev, dtype=torch.float64), phi.to(device=dev, dtype=torch.float64), t_d, desc, ctx
        )
    return _ilt_forward(_prep(theta, dev), _prep(phi, dev), t_d, desc, ctx)


def laplace_reconstruct(
    laplace_rep_func,
    p,
    t,
    recon_dim=None,
    ilt_algorithm="fourier",
    use_sphere_projection=True,
    ilt_reconstruction_terms=33,
    options=None,
    compute_deriv=False,
    x0=None,
):
    """Reconstruct x(t) from the learned Laplace representation; returns (batch, time, recon_dim).

    Same call signature as ``torchlaplace.laplace_reconstruct`` as used at ``w_nl.py:137-144``.
    The result lives on `p`'s device (computation always happens on the GPU).
    """
    if not use_sphere_projection:
        raise NotImplementedError("use_sphere_projection=False is not implemented on the HIP path")
    if compute_deriv or x0 is not None:
        raise NotImplementedError("compute_deriv / x0 are not implemented on the HIP path")
    out_device = p.device
    S = int(ilt_reconstruction_terms)
    if recon_dim is None:
        recon_dim = p.shape[1]
    inp, t2 = rep_func_inputs(p, t, S, ilt_algorithm, options)
    B, Tt = t2.shape
    if torch.is_grad_enabled() and p.requires_grad:
        # training: the query-point columns are constants, the latent columns must stay in the autograd graph
        p_dev = p.to(device=inp.device, dtype=torch.float64)
        inp = torch.cat((inp[..., : 2 * S], p_dev.unsqueeze(1).expand(B, Tt, p_dev.shape[1])), dim=2)
    theta, phi = laplace_rep_func(inp)
    theta = theta.reshape(B * Tt, recon_dim, S)
    phi = phi.reshape(B * Tt, recon_dim, S)
    x = ilt_reconstruct(theta, phi, t2.reshape(-1), ilt_algorithm, options)
    return x.view(B, Tt, recon_dim).to(out_device)
