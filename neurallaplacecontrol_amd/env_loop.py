"""The env side of the reference's evaluation loop, for E independent envs on the device.

The reference steps one env per process on the host: ``step_env`` (``mppi_with_model.py:193-216``) = ``get_action``
(delay buffer, ``:25-28``) + ``env.integrate_system(2, g, s0)`` (``base_env.py:136-173``; ``solver="euler"``,
``overlay.py:39``: one Euler step of ``torch_rhs`` on the reduced state) + ``get_obs`` -- numpy <-> torch hops and an
``odeint`` call per control step.  ``BatchedEnv`` keeps E reduced states and E action buffers in HBM and advances them
with one HIP launch (``nlc_env_step``), so a control step next to ``BatchedMPPIDelay`` needs no host round trip::

    env = BatchedEnv("oderl-cartpole", num_envs=E, dt=0.05, action_delay=delay, seed=0)
    obs = env.reset()
    for it in range(200):
        actions = mppi.command(obs, env.action_buffer)      # (E, nu), device
        obs, reward = env.step(actions)

Same per-env semantics as the reference classes (``envs/oderl/envs/ct{cartpole,pendulum,acrobot}.py``): ``reset``
draws from each env's own ``RandomState`` stream (``seed + e``), ``state`` is the reduced state, ``get_obs`` the trig
observation, the reward is ``diff_reward(new_state, applied_action)``.
"""

import math

import numpy as np
import torch

from . import _lib
from .envs import ENV_DIMS, _check_env

STATE_DIM = {"oderl-cartpole": 4, "oderl-pendulum": 2, "oderl-acrobot": 4}


class BatchedEnv:
    def __init__(self, env_name, num_envs, dt=0.05, action_delay=0, action_buffer_size=4, friction=False,
                 device=None, seed=0):
        self.env_name = _check_env(env_name)
        self.E = int(num_envs)
        self.dt, self.delay, self.B = float(dt), int(action_delay), int(action_buffer_size)
        if not 0 <= self.delay <= self.B - 1:
            raise ValueError("action_delay must be in [0, action_buffer_size - 1]")
        self.friction = bool(friction)
        self.nx, self.nu, self.action_high = ENV_DIMS[env_name]
        self.n = STATE_DIM[env_name]
        if not torch.cuda.is_available():
            raise RuntimeError("neurallaplacecontrol_amd.BatchedEnv needs an AMD MI355X; there is no CPU path")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.ctx = _lib.Ctx(self.device.index)
        self._rngs = None
        self.seed(seed)
        mk = lambda *s: torch.zeros(s, dtype=torch.float64, device=self.device)  # noqa: E731
        self.state = mk(self.E, self.n)
        self.action_buffer = mk(self.E, self.B, self.nu)
        self._obs, self._reward = mk(self.E, self.nx), mk(self.E)
        self.time_step = 0
        self.reset()

    def seed(self, seed=0):
        """Env e draws from its own ``np.random.RandomState(seed + e)`` (the reference seeds one env per process)."""
        self._rngs = [np.random.RandomState(int(seed) + e) for e in range(self.E)]

    def _draw(self, rng):
        if self.env_name == "oderl-cartpole":  # ctcartpole.py:160-170 (swing_up)
            st = rng.uniform(low=-0.05, high=0.05, size=(4,))
            st[2] += np.pi
        elif self.env_name == "oderl-pendulum":  # ctpendulum.py:92-98
            st = rng.uniform(low=-0.1, high=0.1, size=(2,))
            st[0] += np.pi
        else:  # ctacrobot.py:148-151
            st = rng.uniform(low=-0.1, high=0.1, size=(4,))
        return st

    def reset(self, env_ids=None, harness_start=False):
        """``env.reset()`` for the listed envs (default all) and a zeroed action buffer (``mppi_with_model.py:245``).
        ``harness_start``: the evaluation harness then forces the pendulum to ``[pi, 1]`` (``:188-189``)."""
        ids = range(self.E) if env_ids is None else [int(i) for i in env_ids]
        st = self.state.cpu()
        for e in ids:
            st[e] = torch.as_tensor(self._draw(self._rngs[e]))
            if harness_start and self.env_name == "oderl-pendulum":
                st[e] = torch.tensor([math.pi, 1.0], dtype=torch.float64)
        self.state.copy_(st)
        if env_ids is None:
            self.action_buffer.zero_()
            self.time_step = 0
        else:
            self.action_buffer[list(ids)] = 0.0
        return self.get_obs()

    def set_state_(self, state):
        """``env.set_state_`` for all envs: (E, n) reduced states."""
        self.state.copy_(torch.as_tensor(state, dtype=torch.float64).reshape(self.E, self.n))
        return self.get_obs()

    def get_obs(self):
        with torch.cuda.device(self.device):
            self.ctx.use_torch_stream()
            self.ctx.check(self.ctx.lib.nlc_env_obs(self.ctx.h, _lib.ENV_IDS[self.env_name], self.E,
                                                    _lib.ptr(self.state), _lib.ptr(self._obs)))
        return self._obs

    def step(self, actions):
        """One control step of every env: returns (obs (E, nx), reward (E)) device tensors (re-used buffers)."""
        act = torch.as_tensor(actions).detach().to(self.device, torch.float64).reshape(self.E, self.nu).contiguous()
        with torch.cuda.device(self.device):
            self.ctx.use_torch_stream()
            self.ctx.check(
                self.ctx.lib.nlc_env_step(
                    self.ctx.h, _lib.ENV_IDS[self.env_name], int(self.friction), self.dt, self.delay, self.E, self.B,
                    self.nu, _lib.ptr(self.state), _lib.ptr(self.action_buffer), _lib.ptr(act), _lib.ptr(self._obs),
                    _lib.ptr(self._reward),
                )
            )
        self.time_step += 1
        return self._obs, self._reward
