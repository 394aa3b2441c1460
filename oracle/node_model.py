"""Oracle: the NODE baseline dynamics model.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.

Restates ``NODE`` / ``xOdeFuncInXAndU`` (``train_utils.py:637-738``; factory ``:101-125``; ``node_hidden_units=270``,
``node_augment_dim=1``, ``node_method="euler"`` ``config.py:40-42``) behind the harness closure
``state + model(state, window, ts_pred)`` (``mppi_with_model.py:103-122``):

    x   = cat((obs - mean) / std, zeros(augment_dim))                    # normalize=True
    u   = window[:, -1, :]                                               # RAW newest action, never normalised
    f   = Linear(d+aug+nu, H) -> tanh -> Linear(H, H) -> tanh -> Linear(H, d+aug)   on cat(x, u)
    out = odeint(f, x, [0, ts_pred[0] / (dt * 8)], method="euler", options={"step_size": 0.05})[-1][:, :d]

i.e. the model returns the INTEGRATED normalised state (not a difference); the harness adds it to the raw state.

``torchdiffeq`` is absent from the build container and unpinned in the reference's requirements (``torchdiffeq`` in
``requirements.txt``): **parity unpinned for ``odeint``**.  Its fixed-grid Euler solver is restated here from the
published algorithm (``torchdiffeq/_impl/fixed_grid.py``, ``solvers.py`` ``FixedGridODESolver``): grid
``t_k = t_0 + k * step_size`` with ``ceil((t_1 - t_0) / step_size + 1)`` points, the last one replaced by ``t_1``;
``y_{k+1} = y_k + (t_{k+1} - t_k) * f(t_k, y_k)``; the value at ``t_1`` is the last grid value.  Everything else
(the MLP, the normalisation, the planner) is pinned against the real classes by G11.
"""

import math

import numpy as np
import torch
import torch.nn as nn


def euler_substeps(t_end, step_size=0.05):
    """Step sizes of torchdiffeq's fixed-grid solver over [0, t_end] with ``options={"step_size": step_size}``."""
    niters = int(math.ceil(t_end / step_size + 1))
    grid = np.arange(0, niters, dtype=np.float64) * step_size
    grid[-1] = t_end
    return [float(grid[i + 1] - grid[i]) for i in range(niters - 1)]


def odeint_euler(func, y0, t, method="euler", options=None, **_):
    """Stand-in for ``torchdiffeq.odeint(func, y0, t, method="euler", options={"step_size": h})`` with two time
    points (the only call shape the reference uses for NODE, train_utils.py:717-723): returns (2, ...)."""
    assert method == "euler" and t.numel() == 2
    step = float((options or {}).get("step_size", float(t[1] - t[0])))
    t0 = float(t[0])
    y = y0
    tk = t0
    for h in euler_substeps(float(t[1]) - t0, step):
        y = y + h * func(torch.as_tensor(tk, dtype=y.dtype), y)
        tk += h
    return torch.stack((y0, y))


def make_synthetic_state_dict(seed=0, d=5, nu=1, hidden=270, augment_dim=1, state_std=None, action_std=None, dt=0.05,
                              out_scale=0.3):
    """Seeded synthetic weights in the reference constructor's order (three Linears, then xavier_uniform_ on each
    weight in module order: train_utils.py:641-651).  ``out_scale`` shrinks the last layer ("trained-like")."""
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    l0, l2, l4 = nn.Linear(d + nu + augment_dim, hidden), nn.Linear(hidden, hidden), nn.Linear(hidden, d + augment_dim)
    for m in (l0, l2, l4):
        nn.init.xavier_uniform_(m.weight)
    torch.random.set_rng_state(gen_state)
    sd = {}
    for idx, m in ((0, l0), (2, l2), (4, l4)):
        sd[f"x_ode_func_in_x_and_u.linear_tanh_stack.{idx}.weight"] = m.weight.detach().to(torch.float64).clone()
        sd[f"x_ode_func_in_x_and_u.linear_tanh_stack.{idx}.bias"] = m.bias.detach().to(torch.float64).clone()
    sd["x_ode_func_in_x_and_u.linear_tanh_stack.4.weight"] *= out_scale
    sd["x_ode_func_in_x_and_u.linear_tanh_stack.4.bias"] *= out_scale
    sd["state_mean"] = torch.zeros(d, dtype=torch.float64)
    sd["state_std"] = torch.as_tensor(state_std if state_std is not None else np.ones(d), dtype=torch.float64)
    sd["action_mean"] = torch.zeros(1, dtype=torch.float64)
    sd["action_std"] = torch.as_tensor(action_std if action_std is not None else [1.0], dtype=torch.float64)
    sd["dt"] = torch.tensor(dt, dtype=torch.float32).to(torch.float64)
    return sd


def ode_func(sd, x, u):
    """``xOdeFuncInXAndU.forward`` (train_utils.py:659-661)."""
    p = "x_ode_func_in_x_and_u.linear_tanh_stack."
    h = torch.tanh(torch.cat((x, u), 1) @ sd[p + "0.weight"].T + sd[p + "0.bias"])
    h = torch.tanh(h @ sd[p + "2.weight"].T + sd[p + "2.bias"])
    return h @ sd[p + "4.weight"].T + sd[p + "4.bias"]


def forward(sd, obs, window, ts_pred, normalize=True, normalize_time=True, step_size=0.05):
    """``NODE.forward`` (train_utils.py:696-724): obs (N, d), window (N, B, nu) or (N, nu), ts_pred (N, 1)."""
    obs = obs.to(torch.float64)
    d = obs.shape[1]
    aug = sd["x_ode_func_in_x_and_u.linear_tanh_stack.4.weight"].shape[0] - d
    x = (obs - sd["state_mean"]) / sd["state_std"] if normalize else obs
    ts = ts_pred.to(torch.float64)
    if normalize_time:
        ts = ts / (sd["dt"] * 8.0)
    if aug > 0:
        x = torch.cat([x, torch.zeros(obs.shape[0], aug, dtype=torch.float64)], 1)
    window = window.to(torch.float64)
    if window.dim() == 2:
        window = window.unsqueeze(1)
    u = window[:, -1, :]
    t = torch.cat((torch.zeros(1, dtype=torch.float64), ts[0]))
    out = odeint_euler(lambda _t, y: ode_func(sd, y, u), x, t, method="euler", options={"step_size": step_size})
    return out[-1, :, : out.shape[-1] - aug]


def make_dynamics(sd, dt=0.05, normalize=True, normalize_time=True):
    """Harness closure (mppi_with_model.py:103-122)."""

    def dynamics(state, window):
        ts = torch.full((state.shape[0], 1), dt, dtype=torch.float64)
        return state + forward(sd, state, window, ts, normalize, normalize_time)

    return dynamics
