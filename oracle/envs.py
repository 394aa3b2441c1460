"""Oracle: env running costs (a10) and closed-form delayed Euler dynamics (§8f-1).

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.

Running cost = -(diff_obs_reward_(state, exp_reward=False) + diff_ac_reward_(u))
(harness ``mppi_with_model.py:145-171``, default branch ``:163``):
* cartpole  ``envs/oderl/envs/ctcartpole.py:289-346`` (swing_up, length 1, consts base_env.py:28-29)
* pendulum  ``envs/oderl/envs/ctpendulum.py:139-155``
* acrobot   ``envs/oderl/envs/ctacrobot.py:153-166,233-255`` (consts ``:110-111``), angle
  recovery ``base_env.py:297-301``.
Oracle dynamics: ``oracle.py:11-86`` (cartpole), ``:89-174`` (acrobot), ``:177-224`` (pendulum).
"""

import math

import torch

ENV_IDS = {"oderl-cartpole": 0, "oderl-pendulum": 1, "oderl-acrobot": 2}
ACTION_HIGH = {"oderl-cartpole": 3.0, "oderl-pendulum": 2.0, "oderl-acrobot": 5.0}
OBS_DIM = {"oderl-cartpole": 5, "oderl-pendulum": 3, "oderl-acrobot": 6}
ACT_DIM = {"oderl-cartpole": 1, "oderl-pendulum": 1, "oderl-acrobot": 2}


def trig2angle(c, s):
    """base_env.py:297-301: normalise by C twice, then atan2."""
    C = c * c + s * s
    c, s = c / C, s / C
    return torch.atan2(s / C, c / C)


def cartpole_cost(state, u):
    x, xd, cl, sl, thd = (state[..., i] for i in range(5))
    e0 = x + sl - 0.0
    e1 = cl - 1.0
    state_reward = -(e0 * e0 + e1 * e1)
    vel_reward = -(xd * xd) - thd * thd
    ac_reward = -0.01 * (u * u).sum(-1)
    return -((state_reward + 0.01 * vel_reward) + ac_reward)


def cartpole_cost_variant(state_constraint=False, change_goal=False, change_goal_flipped=False):
    """The non-default branches of the harness running_cost (mppi_with_model.py:146-162) for cartpole
    (ctcartpole.py:311-329): moved goal x = -2 / +2, or the soft wall exp(10 e0 + 7) on the cart position."""
    goal_x = (2.0 if change_goal_flipped else -2.0) if change_goal else 0.0

    def cost(state, u):
        x, xd, cl, sl, thd = (state[..., i] for i in range(5))
        e0 = x + sl - goal_x
        e1 = cl - 1.0
        if state_constraint:
            state_reward = -((e0 * e0 + torch.exp(e0 * 10.0 + 7.0)) + e1 * e1)
        else:
            state_reward = -(e0 * e0 + e1 * e1)
        vel_reward = -(xd * xd) - thd * thd
        ac_reward = -0.01 * (u * u).sum(-1)
        return -((state_reward + 0.01 * vel_reward) + ac_reward)

    return cost


def pendulum_cost(state, u):
    c, s, thd = state[..., 0], state[..., 1], state[..., 2]
    state_reward = -((1.0 - c) ** 2 + s * s)
    vel_reward = -(thd * thd)
    ac_reward = -0.01 * (u * u).sum(-1)
    return -((state_reward + 0.01 * vel_reward) + ac_reward)


def acrobot_cost(state, u):
    th1 = trig2angle(state[..., 0], state[..., 1])
    th2 = trig2angle(state[..., 2], state[..., 3])
    v1, v2 = state[..., 4], state[..., 5]
    vel_reward = -(v1 * v1) - v2 * v2
    p1x, p1y = -torch.cos(th1), torch.sin(th1)
    p2x, p2y = p1x - torch.cos(th1 + th2), p1y + torch.sin(th1 + th2)
    state_reward = -((p2x - 1.0 - 1.0) ** 2) - p2y * p2y
    ac_reward = -1e-4 * (u * u).sum(-1)
    return -((state_reward + 1e-1 * vel_reward) + ac_reward)


RUNNING_COST = {"oderl-cartpole": cartpole_cost, "oderl-pendulum": pendulum_cost, "oderl-acrobot": acrobot_cost}


def cartpole_dynamics(state, window, ts, delay, friction=False):
    u = window[:, -(delay + 1), :1].clamp(-3.0, 3.0)
    ts = ts.view(-1, 1)
    x, xd, c, s, thd = (state[:, i : i + 1] for i in range(5))
    C = c * c + s * s
    c, s = c / C, s / C
    th = torch.atan2(s / C, c / C)
    g, fmag, mc, mp, length = 9.8, 3.0, 1.0, 0.1, 1.0
    mt, pml = mp + mc, mp * length
    force = u * fmag
    if friction:
        temp = (force + pml * thd * thd * s - 5e-4 * torch.sign(xd)) / mt
        thacc = (g * s - c * temp - 2e-6 * thd / pml) / (length * (4.0 / 3.0 - mp * c * c / mt))
    else:
        temp = (force + pml * thd * thd * s) / mt
        thacc = (g * s - c * temp) / (length * (4.0 / 3.0 - mp * c * c / mt))
    xacc = temp - pml * thacc * c / mt
    nthd = thd + thacc * ts
    nth = th + thd * ts
    nxd = xd + xacc * ts
    nx = x + xd * ts
    return torch.cat((nx, nxd, torch.cos(nth), torch.sin(nth), nthd), dim=1)


def pendulum_dynamics(state, window, ts, delay, friction=False):
    u = window[:, -(delay + 1), :1].clamp(-2.0, 2.0)
    ts = ts.view(-1, 1)
    c, s, thd = (state[:, i : i + 1] for i in range(3))
    C = c * c + s * s
    th = torch.atan2((s / C) / C, (c / C) / C)
    g, m, l = 10, 1, 1  # noqa: E741
    nth = th + thd * ts
    nthd = thd + (-3 * g / (2 * l) * torch.sin(th + math.pi) + 3.0 / (m * l**2) * u) * ts
    return torch.cat((torch.cos(nth), torch.sin(nth), nthd), dim=1)


def acrobot_dynamics(state, window, ts, delay, friction=False):
    u = window[:, -(delay + 1), :2].clamp(-5.0, 5.0)
    ts = ts.view(-1, 1)
    c1, s1, c2, s2, d1, d2 = (state[:, i : i + 1] for i in range(6))
    th1 = trig2angle(c1, s1)
    th2 = trig2angle(c2, s2)
    m1 = m2 = l1 = I1 = I2 = 1.0
    lc1 = lc2 = 0.5
    g = 9.8
    D1 = m1 * lc1**2 + m2 * (l1**2 + lc2**2 + 2 * l1 * lc2 * torch.cos(th2)) + I1 + I2
    D2 = m2 * (lc2**2 + l1 * lc2 * torch.cos(th2)) + I2
    phi2 = m2 * lc2 * g * torch.cos(th1 + th2 - math.pi / 2.0)
    phi1 = (
        -m2 * l1 * lc2 * d2**2 * torch.sin(th2)
        - 2 * m2 * l1 * lc2 * d2 * d1 * torch.sin(th2)
        + (m1 * lc1 + m2 * l1) * g * torch.cos(th1 - math.pi / 2)
        + phi2
    )
    dd2 = (u[:, 0:1] + D2 / D1 * phi1 - m2 * l1 * lc2 * d1**2 * torch.sin(th2) - phi2) / (
        m2 * lc2**2 + I2 - D2**2 / D1
    )
    dd1 = -(u[:, 1:2] + D2 * dd2 + phi1) / D1
    nd1, nd2 = d1 + dd1 * ts, d2 + dd2 * ts
    nth1, nth2 = th1 + d1 * ts, th2 + d2 * ts
    return torch.cat((torch.cos(nth1), torch.sin(nth1), torch.cos(nth2), torch.sin(nth2), nd1, nd2), dim=1)


ORACLE_DYNAMICS = {
    "oderl-cartpole": cartpole_dynamics,
    "oderl-pendulum": pendulum_dynamics,
    "oderl-acrobot": acrobot_dynamics,
}


def initial_state(env_name, seed=0):
    """Bench/parity start states (SURVEY §8d 'Synthetic inputs')."""
    gen = torch.Generator().manual_seed(seed)
    if env_name == "oderl-cartpole":  # hanging-down start + U(-0.05, 0.05)  (ctcartpole.py:165-167)
        st = (torch.rand(4, generator=gen, dtype=torch.float64) - 0.5) * 0.1
        th = st[2] + math.pi
        return torch.stack((st[0], st[1], torch.cos(th), torch.sin(th), st[3]))
    if env_name == "oderl-pendulum":  # mppi_with_model.py:188-189  state = [pi, 1]
        return torch.tensor([math.cos(math.pi), math.sin(math.pi), 1.0], dtype=torch.float64)
    if env_name == "oderl-acrobot":  # ctacrobot.py:149  U(-0.1, 0.1)
        st = (torch.rand(4, generator=gen, dtype=torch.float64) - 0.5) * 0.2
        return torch.stack((torch.cos(st[0]), torch.sin(st[0]), torch.cos(st[1]), torch.sin(st[1]), st[2], st[3]))
    raise ValueError(env_name)
