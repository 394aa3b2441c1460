"""Oracle: inverse Laplace transform reconstruction (stage a9 of SURVEY.md §8).

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.

Restates ``torchlaplace.laplace_reconstruct`` as called from the reference at
``w_nl.py:137-144`` (and ``w_latent_ode.py:88-94``).  The package itself is an
un-vendored, unpinned PyPI dependency (``requirements.txt:17``) that is absent
here, so this file follows the published algorithm:

* query points + Fourier-series ILT: Neural Laplace paper (arXiv 2206.04843),
  eq. for the Riemann-sphere maps ``u``/``v`` and the Fourier ILT;
* de Hoog-Knight-Stokes: mpmath 1.3.0 ``calculus/inverselaplace.py:426-432``
  (abscissa) and ``:476-531`` (QD table, continued fraction, remainder).

**Parity unpinned vs upstream torchlaplace**: the defaults ``alpha``, ``tol``,
``scale`` and the rep-func input order ``[theta_s | phi_s | p]`` are recalled,
not verifiable offline.  They are parameters here and in the
HIP path, with the same defaults on both sides.
"""

import math

import torch

# recalled torchlaplace defaults
ILT_DEFAULTS = {
    "fourier": dict(alpha=1.0e-3, scale=2.0),
    "dehoog": dict(alpha=1.0e-10, scale=2.0),
    # the two other closed-form algorithms torchlaplace offers (reference knob config.py:36 nl_ilt_algorithm): no
    # abscissa parameters; nodes and weights restated from mpmath 1.3.0 calculus/inverselaplace.py (FixedTalbot :18-187,
    # Stehfest :192-352), parity unpinned vs upstream like the rest of stage a9
    "fixed_tablot": dict(alpha=0.0, scale=1.0),
    "stehfest": dict(alpha=0.0, scale=1.0),
}
LINEAR_ALGOS = ("fixed_tablot", "stehfest")


def talbot_tables(S):
    """Fixed Talbot (Abate & Valko 2004) with M = S nodes, r = 2M/5 (mpmath FixedTalbot.calc_laplace_parameter):
    s_k = delta_k / t,  delta_0 = r,  delta_k = r theta_k (cot theta_k + i),  theta_k = k pi / M;
    x(t) = (1/t) Re sum_k W_k F(s_k),  W_0 = (2/5) e^{r} / 2,
    W_k = (2/5) e^{delta_k} (1 + i theta_k (1 + cot^2 theta_k) - i cot theta_k)   (calc_time_domain_solution).
    Returns (delta_re, delta_im, W_re, W_im), each (S,) float64."""
    M = S
    r = 2.0 * M / 5.0
    k = torch.arange(M, dtype=torch.float64)
    theta = k * math.pi / M
    cot = torch.zeros(M, dtype=torch.float64)
    cot[1:] = 1.0 / torch.tan(theta[1:])
    d_re = r * theta * cot
    d_im = r * theta
    d_re[0], d_im[0] = r, 0.0
    delta = torch.complex(d_re, d_im)
    fac = torch.complex(torch.ones(M, dtype=torch.float64), theta * (1.0 + cot * cot) - cot)
    W = 0.4 * torch.exp(delta) * fac
    W[0] = 0.4 * math.exp(r) / 2.0
    return d_re, d_im, W.real.clone(), W.imag.clone()


def stehfest_tables(S):
    """Gaver-Stehfest with M = S (even) real nodes s_k = k ln2 / t, k = 1..M, and Salzer weights V_k
    (mpmath Stehfest._coeff): x(t) = (ln2 / t) sum_k V_k Re F(s_k).  Returns (node_re, node_im, W_re, W_im)."""
    if S % 2 or S < 2:
        raise ValueError("stehfest needs an even number of terms")
    M, M2 = S, S // 2
    V = []
    for k in range(1, M + 1):
        z = 0
        for j in range((k + 1) // 2, min(k, M2) + 1):
            z += (j**M2 * math.factorial(2 * j)) / (
                math.factorial(M2 - j) * math.factorial(j) * math.factorial(j - 1) * math.factorial(k - j) * math.factorial(2 * j - k)
            )
        V.append((-1) ** (k + M2) * z)
    V = torch.tensor(V, dtype=torch.float64)
    node = torch.arange(1, M + 1, dtype=torch.float64) * math.log(2.0)
    return node, torch.zeros(M, dtype=torch.float64), math.log(2.0) * V, torch.zeros(M, dtype=torch.float64)


def linear_tables(algo, S):
    return talbot_tables(S) if algo == "fixed_tablot" else stehfest_tables(S)


def ilt_options(algo, options=None):
    """Resolve (alpha, tol, scale) for an algorithm; tol defaults to 10*alpha."""
    if algo not in ILT_DEFAULTS:
        raise ValueError(f"unsupported ilt_algorithm {algo!r} (oracle restates fourier, dehoog, fixed_tablot, stehfest)")
    o = dict(ILT_DEFAULTS[algo])
    if options:
        o.update(options)
    if o.get("tol") is None:
        o["tol"] = 10.0 * o["alpha"]
    return o["alpha"], o["tol"], o["scale"]


def query_points(t, S, alpha, tol, scale):
    """s_k(t) = gamma + i*pi*k/T,  T = scale*t,  gamma = alpha - ln(tol)/(scale*T).

    t: (...,) float64 -> (s_real (...,S), s_imag (...,S), T (...,), gamma (...,))
    (mpmath inverselaplace.py:426-432; shared by Fourier and de Hoog.)
    """
    t = t.to(torch.float64)
    T = scale * t
    gamma = alpha - math.log(tol) / (scale * T)
    k = torch.arange(S, dtype=torch.float64)
    s_real = gamma.unsqueeze(-1).expand(*t.shape, S).clone()
    s_imag = math.pi * k / T.unsqueeze(-1)
    return s_real, s_imag, T, gamma


def complex_to_sphere(s_real, s_imag):
    """Riemann-sphere projection u: theta = atan2(Im, Re), phi = asin((|s|^2-1)/(|s|^2+1))."""
    a2 = s_real * s_real + s_imag * s_imag
    theta = torch.atan2(s_imag, s_real)
    phi = torch.asin((a2 - 1.0) / (a2 + 1.0))
    return theta, phi


def sphere_to_complex(theta, phi):
    """Inverse map v: F = tan(phi/2 + pi/4) * (cos theta + i sin theta)."""
    r = torch.tan(phi / 2.0 + math.pi / 4.0)
    return r * torch.cos(theta), r * torch.sin(theta)


def fourier_line_integrate(f_real, f_imag, t, T, gamma):
    """x(t) = e^{gamma t}/T * [ Re F_0 / 2 + sum_{k>=1} Re(F_k e^{i pi k t/T}) ].

    f_*: (..., S) with leading dims broadcastable against t's (...); returns (...)
    """
    S = f_real.shape[-1]
    k = torch.arange(S, dtype=torch.float64)
    ang = math.pi * k * (t / T).unsqueeze(-1)
    re = f_real * torch.cos(ang) - f_imag * torch.sin(ang)
    total = 0.5 * re[..., 0] + re[..., 1:].sum(-1)
    return torch.exp(gamma * t) / T * total


def _cdiv(ar, ai, br, bi):
    den = br * br + bi * bi
    return (ar * br + ai * bi) / den, (ai * br - ar * bi) / den


def dehoog_line_integrate(f_real, f_imag, t, T, gamma):
    """de Hoog-Knight-Stokes accelerated Fourier ILT (mpmath inverselaplace.py:476-531).

    f_*: (..., S) with S = 2M+1; t, T, gamma broadcastable to (...); returns (...).
    Uses torch complex128 for brevity; arithmetic order follows mpmath.
    """
    S = f_real.shape[-1]
    if S % 2 != 1 or S < 3:
        raise ValueError("de Hoog needs an odd number of terms 2M+1 >= 3")
    M = (S - 1) // 2
    fp = torch.complex(f_real, f_imag)
    lead = fp.shape[:-1]
    e = torch.zeros(*lead, S, M + 1, dtype=torch.complex128)
    q = torch.zeros(*lead, 2 * M, M, dtype=torch.complex128)
    q[..., 0, 0] = fp[..., 1] / (fp[..., 0] / 2.0)
    for i in range(1, 2 * M):
        q[..., i, 0] = fp[..., i + 1] / fp[..., i]
    for r in range(1, M + 1):
        mr = 2 * (M - r) + 1
        e[..., 0:mr, r] = q[..., 1 : mr + 1, r - 1] - q[..., 0:mr, r - 1] + e[..., 1 : mr + 1, r - 1]
        if r != M:
            rq = r + 1
            mrq = 2 * (M - rq) + 1 + 2
            for i in range(mrq):
                q[..., i, rq - 1] = q[..., i + 1, rq - 2] * e[..., i + 1, rq - 1] / e[..., i, rq - 1]
    d = torch.zeros(*lead, S, dtype=torch.complex128)
    d[..., 0] = fp[..., 0] / 2.0
    for r in range(1, M + 1):
        d[..., 2 * r - 1] = -q[..., 0, r - 1]
        d[..., 2 * r] = -e[..., 0, r]
    ang = math.pi * (t / T)
    z = torch.complex(torch.cos(ang), torch.sin(ang))
    A_prev = torch.zeros(lead, dtype=torch.complex128)
    A_cur = d[..., 0].clone()
    B_prev = torch.ones(lead, dtype=torch.complex128)
    B_cur = torch.ones(lead, dtype=torch.complex128)
    for i in range(1, 2 * M):
        A_next = A_cur + d[..., i] * A_prev * z
        B_next = B_cur + d[..., i] * B_prev * z
        A_prev, A_cur = A_cur, A_next
        B_prev, B_cur = B_cur, B_next
    brem = (1.0 + (d[..., 2 * M - 1] - d[..., 2 * M]) * z) / 2.0
    rem = brem * (torch.sqrt(1.0 + d[..., 2 * M] * z / brem) - 1.0)
    A_np = A_cur + rem * A_prev
    B_np = B_cur + rem * B_prev
    return torch.exp(gamma * t) / T * (A_np / B_np).real


LINE_INTEGRATE = {"fourier": fourier_line_integrate, "dehoog": dehoog_line_integrate}


def ilt_from_sphere(theta, phi, t, algo="fourier", options=None):
    """theta, phi: (N, d, S) rep-func outputs; t: (N,) -> x (N, d)."""
    alpha, tol, scale = ilt_options(algo, options)
    t = t.to(torch.float64)
    if algo in LINEAR_ALGOS:
        _, _, wr, wi = linear_tables(algo, theta.shape[-1])
        fr, fi = sphere_to_complex(theta, phi)
        return (fr * wr - fi * wi).sum(-1) / t.view(-1, 1)
    T = scale * t
    gamma = alpha - math.log(tol) / (scale * T)
    fr, fi = sphere_to_complex(theta, phi)
    return LINE_INTEGRATE[algo](fr, fi, t.unsqueeze(-1), T.unsqueeze(-1), gamma.unsqueeze(-1))


def rep_func_inputs(p, t, S, algo="fourier", options=None):
    """Rows ``[theta_s(0..S-1) | phi_s(0..S-1) | p]`` fed to the representation MLP.

    p: (B, P); t: (B, Tt) or (Tt,) -> (B, Tt, 2S+P), t2 (B, Tt)
    """
    alpha, tol, scale = ilt_options(algo, options)
    B = p.shape[0]
    t = t.to(torch.float64)
    if t.dim() == 0:
        t = t.view(1)
    t2 = t.view(1, -1).expand(B, -1) if t.dim() == 1 else t
    if algo in LINEAR_ALGOS:
        nr, ni, _, _ = linear_tables(algo, S)
        sr, si = nr / t2.unsqueeze(-1), ni / t2.unsqueeze(-1)
    else:
        sr, si, _, _ = query_points(t2, S, alpha, tol, scale)
    th_s, ph_s = complex_to_sphere(sr, si)
    Tt = t2.shape[1]
    inp = torch.cat((th_s, ph_s, p.view(B, 1, -1).expand(B, Tt, p.shape[1])), dim=-1)
    return inp, t2


def laplace_reconstruct(
    laplace_rep_func,
    p,
    t,
    recon_dim=None,
    ilt_algorithm="fourier",
    ilt_reconstruction_terms=33,
    options=None,
):
    """Oracle for ``torchlaplace.laplace_reconstruct`` (call sites w_nl.py:137-144).

    laplace_rep_func: callable((B, Tt, 2S+P)) -> (theta, phi) each (B*Tt, d, S)
    p: (B, P) ; t: (B, Tt) or (Tt,)  ->  (B, Tt, d)
    """
    S = ilt_reconstruction_terms
    if recon_dim is None:
        recon_dim = p.shape[1]
    inp, t2 = rep_func_inputs(p, t, S, ilt_algorithm, options)
    B, Tt = t2.shape
    theta, phi = laplace_rep_func(inp)
    theta = theta.reshape(B * Tt, recon_dim, S)
    phi = phi.reshape(B * Tt, recon_dim, S)
    x = ilt_from_sphere(theta, phi, t2.reshape(-1), ilt_algorithm, options)
    return x.view(B, Tt, recon_dim)
