"""Oracle: MPPI with an action-delay window (stages a1-a4, a11, a12 of SURVEY.md §8).

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.

Restates ``planners/mppi_delay.py``: ``MPPIDelay.__init__`` :64-184, ``command`` :193-224,
``_compute_total_cost_batch`` :315-345, ``_compute_rollout_costs`` :232-313,
``_bound_action`` :347-353, ``reset`` :226-230; and the harness delay buffer
``get_action`` ``mppi_with_model.py:25-28``.
"""

import torch
from torch.distributions.multivariate_normal import MultivariateNormal


def perturb(U, noise, u_scale, u_min, u_max, sample_null_action=False):
    """:321-328 -- V = U + eps; clamp(V*u_scale)/u_scale; eps <- V - U."""
    V = U + noise
    if sample_null_action:
        V[-1] = 0
    if u_max is not None:
        V = torch.max(torch.min(V * u_scale, u_max), u_min) / u_scale
    else:
        V = V * u_scale / u_scale
    return V, V - U


def rollout(state, action_buffer, V, u_scale, dynamics, running_cost, nx, rollout_samples=1, rollout_var_cost=0.0,
            rollout_var_discount=0.95):
    """:232-313, encode_obs_time=False (the terminal cost is added by the caller).
    rollout_samples M > 1 (:291-292, 310): the reference does NOT replicate the state M times -- its cost_samples rows are
    M copies of the same running cost, and ``c.var(dim=0)`` of the (K,) cost vector is the (unbiased) variance OVER THE K
    SAMPLES, one number per horizon step: every sample's cost gets the same rollout_var_cost * sum_t var_t discount^t."""
    K, T, nu = V.shape
    B = action_buffer.shape[0]
    x = state if state.shape == (K, nx) else state.view(1, -1).repeat(K, 1)
    A = u_scale * V
    hist = torch.cat((action_buffer[1:].view(1, -1, nu).repeat(K, 1, 1), A), dim=1)
    cost = torch.zeros(K, dtype=V.dtype)
    cost_var = torch.zeros(K, dtype=V.dtype)
    states, actions = [], []
    for t in range(T):
        x = dynamics(x, hist[:, t : t + B, :])
        u = hist[:, t + B - 1, :]
        c = running_cost(x, u)
        cost = cost + c
        if rollout_samples > 1:
            cost_var = cost_var + c.var(dim=0) * (rollout_var_discount**t)
        states.append(x)
        actions.append(u)
    cost = cost + cost_var * rollout_var_cost
    return cost, torch.stack(states, dim=-2), torch.stack(actions, dim=-2)


def mppi_command(
    U,
    state,
    action_buffer,
    noise,
    dynamics,
    running_cost,
    nx,
    noise_sigma_inv,
    lambda_=1.0,
    u_scale=1.0,
    u_min=None,
    u_max=None,
    u_init=None,
    sample_null_action=False,
    noise_abs_cost=False,
    u_per_command=1,
    terminal_state_cost=None,
    rollout_samples=1,
    rollout_var_cost=0.0,
    rollout_var_discount=0.95,
):
    """One ``command()`` given the noise draw; returns a dict of every public output."""
    U = torch.roll(U, -1, dims=0)
    U[-1] = 0.0 if u_init is None else u_init
    V, eps = perturb(U, noise, u_scale, u_min, u_max, sample_null_action)
    if noise_abs_cost:
        action_cost = lambda_ * torch.abs(eps) @ noise_sigma_inv
    else:
        action_cost = lambda_ * eps @ noise_sigma_inv
    cost, states, actions = rollout(state, action_buffer, V, u_scale, dynamics, running_cost, nx, rollout_samples,
                                    rollout_var_cost, rollout_var_discount)
    if terminal_state_cost is not None:  # :306-308, called with the (K,T,nx) states and the scaled (K,T,nu) actions
        cost = cost + terminal_state_cost(states, actions)
    actions = actions / u_scale
    cost = cost + torch.sum(U * action_cost, dim=(1, 2))
    beta = torch.min(cost)
    w = torch.exp(-(1.0 / lambda_) * (cost - beta))
    eta = torch.sum(w)
    omega = (1.0 / eta) * w
    U = U.clone()
    for t in range(U.shape[0]):
        U[t] += torch.sum(omega.view(-1, 1) * eps[:, t], dim=0)
    action = U[:u_per_command]
    if u_per_command == 1:
        action = action[0]
    return dict(
        U=U,
        action=action * u_scale,
        cost_total=cost,
        cost_total_non_zero=w,
        omega=omega,
        noise=eps,
        perturbed_action=V,
        states=states,
        actions=actions,
        beta=beta,
        eta=eta,
    )


def shard_partials(cost, eps, lambda_=1.0):
    """Per-rank partials (beta_r, eta_r, S_r[t,j]) of SURVEY §8e for one K-shard."""
    beta = torch.min(cost)
    w = torch.exp(-(1.0 / lambda_) * (cost - beta))
    return torch.cat((beta.view(1), w.sum().view(1), torch.einsum("k,ktj->tj", w, eps).reshape(-1)))


def merge_partials(parts, lambda_=1.0):
    """Merge gathered (G, 2+T*nu) partials: beta=min, rescale by exp(-(beta_r-beta)/lambda)."""
    beta = parts[:, 0].min()
    scale = torch.exp(-(parts[:, 0] - beta) / lambda_)
    eta = (scale * parts[:, 1]).sum()
    dU = (scale.view(-1, 1) * parts[:, 2:]).sum(0) / eta
    return beta, eta, dU


def get_action(action_buffer, action, action_delay):
    """Harness side of the delay contract (mppi_with_model.py:25-28)."""
    action_buffer = torch.roll(action_buffer, -1, dims=0)
    action_buffer[-1] = action
    return action_buffer, action_buffer[-(action_delay + 1)]


class MPPIOracle:
    """Stateful restatement with the reference's RNG consumption (ctor :164, command :319, reset :230)."""

    def __init__(
        self,
        dynamics,
        running_cost,
        nx,
        noise_sigma,
        num_samples=100,
        horizon=15,
        lambda_=1.0,
        u_min=None,
        u_max=None,
        u_scale=1,
        U_init=None,
        sample_null_action=False,
        noise_abs_cost=False,
    ):
        self.K, self.T, self.nx = num_samples, horizon, nx
        self.nu = 1 if noise_sigma.dim() == 0 else noise_sigma.shape[0]
        self.dtype = noise_sigma.dtype
        self.lambda_ = lambda_
        if self.nu == 1:
            noise_sigma = noise_sigma.view(-1, 1)
        if u_max is not None and u_min is None:
            u_min = -torch.as_tensor(u_max)
        if u_min is not None and u_max is None:
            u_max = -torch.as_tensor(u_min)
        self.u_min, self.u_max, self.u_scale = u_min, u_max, u_scale
        self.noise_sigma = noise_sigma
        self.noise_sigma_inv = torch.inverse(noise_sigma)
        self.noise_dist = MultivariateNormal(torch.zeros(self.nu, dtype=self.dtype), covariance_matrix=noise_sigma)
        self.U = U_init if U_init is not None else self.noise_dist.sample((self.T,))
        self.F, self.running_cost = dynamics, running_cost
        self.sample_null_action, self.noise_abs_cost = sample_null_action, noise_abs_cost
        self.last = None

    def reset(self):
        self.U = self.noise_dist.sample((self.T,))

    def command(self, state, action_buffer, noise=None):
        state = torch.as_tensor(state, dtype=self.dtype)
        if noise is None:
            noise = self.noise_dist.sample((self.K, self.T))
        out = mppi_command(
            self.U,
            state,
            action_buffer,
            noise,
            self.F,
            self.running_cost,
            self.nx,
            self.noise_sigma_inv,
            self.lambda_,
            self.u_scale,
            self.u_min,
            self.u_max,
            sample_null_action=self.sample_null_action,
            noise_abs_cost=self.noise_abs_cost,
        )
        self.U = out["U"]
        self.last = out
        return out["action"]
