"""Dynamics / running-cost objects the fused planner recognises.

The reference hands ``MPPIDelay`` two Python closures built in ``mppi_with_model.py:103-171``.  A closure
is opaque, so a drop-in harness passes these objects instead; each one is ALSO a callable with the
closure's semantics, which is what the generic (external-callable) planner path and the reference's own
``MPPIDelay`` use.

  NLDynamics(model, ts_pred)          state + model(state, window, ts_pred)          mppi_with_model.py:103-122
  OracleDynamics(env, ts, delay)      oracle.*_dynamics_dt_delay                      mppi_with_model.py:129-143
  EnvCost(env)                        -(diff_obs_reward_ + diff_ac_reward_)           mppi_with_model.py:145-171
"""

import math

import torch

from . import _lib

ENV_DIMS = {"oderl-cartpole": (5, 1, 3.0), "oderl-pendulum": (3, 1, 2.0), "oderl-acrobot": (6, 2, 5.0),
            "oderl-cartpole-notrig": (4, 1, 3.0)}


def _check_env(env_name):
    if env_name not in _lib.ENV_IDS:
        raise ValueError(f"unknown env {env_name!r}; expected one of {sorted(_lib.ENV_IDS)}")
    return env_name


class NLDynamics:
    """Learned-model dynamics closure of the harness: ``state + model(state, window, ts_pred)``.

    ``model`` is a :class:`NeuralLaplaceModel` or the Delta-t RNN baseline :class:`DeltaTRNN` (the closure is the
    same for every learned model, ``mppi_with_model.py:103-122``; only ``model_name == "nl"`` gets the time channel)."""

    def __init__(self, model, ts_pred):
        self.model = model
        t = torch.as_tensor(ts_pred, dtype=torch.float64).reshape(-1)
        if t.numel() > 1 and not bool((t == t[0]).all()):
            raise NotImplementedError("the fused rollout needs one constant ts_pred (the harness passes dt)")
        self.ts_pred = float(t[0])

    def __call__(self, state, perturbed_action):
        # one ts_pred tensor per (batch, device), as the harness builds it once outside the loop (mppi_with_model.py:74)
        key = (state.shape[0], str(state.device))
        if getattr(self, "_ts_key", None) != key:
            self._ts = torch.full((state.shape[0], 1), self.ts_pred, dtype=torch.float64, device=state.device)
            self._ts_key = key
        ts = self._ts
        is_nl = getattr(self.model, "_dyn_id", _lib.DYN_NL) == _lib.DYN_NL
        if is_nl and self.model.encode_obs_time and perturbed_action.shape[2] == self.model.action_dim:
            # the harness closure appends a constant time channel B-1 .. 0 (mppi_with_model.py:110-119)
            B = perturbed_action.shape[1]
            tch = torch.flip(torch.arange(B, device=perturbed_action.device), (0,)).view(1, B, 1)
            perturbed_action = torch.cat((perturbed_action, tch.repeat(perturbed_action.shape[0], 1, 1)), dim=2)
        return state + self.model(state, perturbed_action, ts).view(state.shape)


class OracleDynamics:
    """Closed-form delayed Euler step of the env (``model_name == 'oracle'``)."""

    def __init__(self, env_name, ts=0.05, delay=0, friction=False):
        self.env_name = _check_env(env_name)
        self.ts = float(torch.as_tensor(ts, dtype=torch.float64).reshape(-1)[0])
        self.delay = int(delay)
        self.friction = bool(friction)

    def __call__(self, state, perturbed_action):
        raise NotImplementedError(
            "OracleDynamics is evaluated inside the fused HIP rollout; pass it to neurallaplacecontrol_amd.MPPIDelay"
        )


def _trig2angle(c, s):
    C = c * c + s * s
    c, s = c / C, s / C
    return torch.atan2(s / C, c / C)


class EnvCost:
    """Running cost of the three reference envs on the trig observation, and of cartpole on its raw 4-dim state
    (``obs_trans=False``); same formulas as the HIP kernels."""

    def __init__(self, env_name):
        self.env_name = _check_env(env_name)

    def __call__(self, state, action):
        uu = (action * action).sum(-1)
        if self.env_name == "oderl-cartpole":
            e0, e1 = state[..., 0] + state[..., 3] - 0.0, state[..., 2] - 1.0
            sr = -(e0 * e0 + e1 * e1)
            vr = -(state[..., 1] ** 2) - state[..., 4] ** 2
            return -((sr + 0.01 * vr) + (-0.01 * uu))
        if self.env_name == "oderl-cartpole-notrig":  # ctcartpole.py:297-300: explicit angle
            cl, sl = 1.0 * torch.cos(state[..., 2]), 1.0 * torch.sin(state[..., 2])
            e0, e1 = state[..., 0] + sl - 0.0, cl - 1.0
            sr = -(e0 * e0 + e1 * e1)
            vr = -(state[..., 1] ** 2) - state[..., 3] ** 2
            return -((sr + 0.01 * vr) + (-0.01 * uu))
        if self.env_name == "oderl-pendulum":
            sr = -((1.0 - state[..., 0]) ** 2 + state[..., 1] ** 2)
            vr = -(state[..., 2] ** 2)
            return -((sr + 0.01 * vr) + (-0.01 * uu))
        th1 = _trig2angle(state[..., 0], state[..., 1])
        th2 = _trig2angle(state[..., 2], state[..., 3])
        vr = -(state[..., 4] ** 2) - state[..., 5] ** 2
        p1x, p1y = -torch.cos(th1), torch.sin(th1)
        p2x, p2y = p1x - torch.cos(th1 + th2), p1y + torch.sin(th1 + th2)
        sr = -((p2x - 1.0 - 1.0) ** 2) - p2y * p2y
        return -((sr + 1e-1 * vr) + (-1e-4 * uu))


def noise_sigma(nu, sigma=1.0, dtype=torch.double, device="cpu"):
    """Sigma = sigma^2 (1/2 11^T + 1/2 I), the harness's MPPI noise covariance (mppi_with_model.py:66-70)."""
    g = sigma**2
    return torch.ones((nu, nu), device=device, dtype=dtype) * 0.5 * g + torch.eye(nu, device=device, dtype=dtype) * (
        g - 0.5 * g
    )


def initial_state(env_name, generator=None):
    """Start observation used by the bench / smoke."""
    if env_name == "oderl-cartpole":
        st = (torch.rand(4, generator=generator, dtype=torch.float64) - 0.5) * 0.1
        th = st[2] + math.pi
        return torch.stack((st[0], st[1], torch.cos(th), torch.sin(th), st[3]))
    if env_name == "oderl-cartpole-notrig":
        st = (torch.rand(4, generator=generator, dtype=torch.float64) - 0.5) * 0.1
        return torch.stack((st[0], st[1], st[2] + math.pi, st[3]))
    if env_name == "oderl-pendulum":
        return torch.tensor([math.cos(math.pi), math.sin(math.pi), 1.0], dtype=torch.float64)
    st = (torch.rand(4, generator=generator, dtype=torch.float64) - 0.5) * 0.2
    return torch.stack((torch.cos(st[0]), torch.sin(st[0]), torch.cos(st[1]), torch.sin(st[1]), st[2], st[3]))
