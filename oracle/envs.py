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

import numpy as np
import torch

# "oderl-cartpole-notrig": the same env built with obs_trans=False (ctcartpole.py:60): 4-dim state [x, xdot, theta, thetadot]
ENV_IDS = {"oderl-cartpole": 0, "oderl-pendulum": 1, "oderl-acrobot": 2, "oderl-cartpole-notrig": 3}
ACTION_HIGH = {"oderl-cartpole": 3.0, "oderl-pendulum": 2.0, "oderl-acrobot": 5.0, "oderl-cartpole-notrig": 3.0}
OBS_DIM = {"oderl-cartpole": 5, "oderl-pendulum": 3, "oderl-acrobot": 6, "oderl-cartpole-notrig": 4}
ACT_DIM = {"oderl-cartpole": 1, "oderl-pendulum": 1, "oderl-acrobot": 2, "oderl-cartpole-notrig": 1}


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


def cartpole_notrig_cost(state, u):
    """ctcartpole.py:297-300 (s.shape[-1] == 4: cos / sin of the explicit angle), then the swing-up reward :303-339."""
    x, xd, th, thd = (state[..., i] for i in range(4))
    cl, sl = 1.0 * torch.cos(th), 1.0 * torch.sin(th)
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


RUNNING_COST = {"oderl-cartpole": cartpole_cost, "oderl-pendulum": pendulum_cost, "oderl-acrobot": acrobot_cost,
                "oderl-cartpole-notrig": cartpole_notrig_cost}


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


def cartpole_notrig_dynamics(state, window, ts, delay, friction=False):
    """oracle.py:11-86 on the 4-dim state (the `else` branches :38-44 and :80-86: explicit angle in, explicit angle out)."""
    u = window[:, -(delay + 1), :1].clamp(-3.0, 3.0)
    ts = ts.view(-1, 1)
    x, xd, th, thd = (state[:, i : i + 1] for i in range(4))
    c, s = torch.cos(th), torch.sin(th)
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
    return torch.cat((x + xd * ts, xd + xacc * ts, th + thd * ts, thd + thacc * ts), dim=1)


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
    "oderl-cartpole-notrig": cartpole_notrig_dynamics,
}


def initial_state(env_name, seed=0):
    """Bench/parity start states (SURVEY §8d 'Synthetic inputs')."""
    gen = torch.Generator().manual_seed(seed)
    if env_name == "oderl-cartpole":  # hanging-down start + U(-0.05, 0.05)  (ctcartpole.py:165-167)
        st = (torch.rand(4, generator=gen, dtype=torch.float64) - 0.5) * 0.1
        th = st[2] + math.pi
        return torch.stack((st[0], st[1], torch.cos(th), torch.sin(th), st[3]))
    if env_name == "oderl-cartpole-notrig":  # the same start as a raw state
        st = (torch.rand(4, generator=gen, dtype=torch.float64) - 0.5) * 0.1
        return torch.stack((st[0], st[1], st[2] + math.pi, st[3]))
    if env_name == "oderl-pendulum":  # mppi_with_model.py:188-189  state = [pi, 1]
        return torch.tensor([math.cos(math.pi), math.sin(math.pi), 1.0], dtype=torch.float64)
    if env_name == "oderl-acrobot":  # ctacrobot.py:149  U(-0.1, 0.1)
        st = (torch.rand(4, generator=gen, dtype=torch.float64) - 0.5) * 0.2
        return torch.stack((torch.cos(st[0]), torch.sin(st[0]), torch.cos(st[1]), torch.sin(st[1]), st[2], st[3]))
    raise ValueError(env_name)


# ------------------------------------------------------------------------------------------------------------------
# The env side of the evaluation loop: one control step of ``step_env`` (mppi_with_model.py:193-216)
# = get_action (delay buffer, :25-28) + env.integrate_system(2, g, s0) (base_env.py:136-173) + get_obs (:83-89).  With
# the harness's solver="euler" (overlay.py:39) and ts = [0, dt] the odeint call is one explicit Euler step of
# torch_rhs on the REDUCED state (angles, not their cos/sin).  Pinned against the real env classes by G10
# (tests/golden/make_golden_env.py); only "odeint(method='euler') = s + dt * rhs" is restated from torchdiffeq.
STATE_DIM = {"oderl-cartpole": 4, "oderl-pendulum": 2, "oderl-acrobot": 4}


def env_rhs(env_name, s, a, friction=False):
    """``torch_rhs`` on reduced states s (..., n), actions a (..., nu)."""
    if env_name == "oderl-cartpole":  # ctcartpole.py:185-237, 4-D branch
        xd, th, thd = s[..., 1], s[..., 2], s[..., 3]
        c, sn = torch.cos(th), torch.sin(th)
        g, fmag, mc, mp, length = 9.8, 3.0, 1.0, 0.1, 1.0
        mt, pml = mp + mc, mp * length
        force = torch.clamp(a, min=-fmag, max=fmag)[..., 0] * fmag
        if friction:
            temp = (force + pml * thd * thd * sn - 5e-4 * torch.sign(xd)) / mt
            thacc = (g * sn - c * temp - 2e-6 * thd / pml) / (length * (4.0 / 3.0 - mp * c * c / mt))
        else:
            temp = (force + pml * thd * thd * sn) / mt
            thacc = (g * sn - c * temp) / (length * (4.0 / 3.0 - mp * c * c / mt))
        xacc = temp - pml * thacc * c / mt
        return torch.stack([xd, xacc, thd, thacc], -1)
    if env_name == "oderl-pendulum":  # ctpendulum.py:111-125 (no clamp in the env's rhs)
        th, thd = s[..., 0], s[..., 1]
        g, m, l = 10.0, 1.0, 1.0  # noqa: E741
        return torch.stack([thd, (-3 * g / (2 * l) * torch.sin(th + np.pi) + 3.0 / (m * l**2) * a[..., 0])], -1)
    if env_name == "oderl-acrobot":  # ctacrobot.py:168-228, fully actuated, 4-D branch (no clamp)
        th1, th2, d1, d2 = s[..., 0], s[..., 1], s[..., 2], s[..., 3]
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
        dd2 = (a[..., 0] + D2 / D1 * phi1 - m2 * l1 * lc2 * d1**2 * torch.sin(th2) - phi2) / (m2 * lc2**2 + I2 - D2**2 / D1)
        dd1 = -(a[..., 1] + D2 * dd2 + phi1) / D1
        return torch.stack([d1, d2, dd1, dd2], -1)
    raise ValueError(env_name)


def env_obs(env_name, s):
    """``torch_transform_states``: reduced state -> trig observation."""
    if env_name == "oderl-cartpole":  # ctcartpole.py:107-129 (length = 1)
        return torch.stack([s[..., 0], s[..., 1], 1.0 * torch.cos(s[..., 2]), 1.0 * torch.sin(s[..., 2]), s[..., 3]], -1)
    if env_name == "oderl-pendulum":  # ctpendulum.py:72-78
        return torch.stack([torch.cos(s[..., 0]), torch.sin(s[..., 0]), s[..., 1]], -1)
    return torch.stack(  # ctacrobot.py:125-137
        [torch.cos(s[..., 0]), torch.sin(s[..., 0]), torch.cos(s[..., 1]), torch.sin(s[..., 1]), s[..., 2], s[..., 3]], -1
    )


def env_obs2state(env_name, obs):
    """``obs2state`` (ctcartpole.py:172-183, ctpendulum.py:104-109, ctacrobot.py:153-166)."""
    if env_name == "oderl-cartpole":
        return torch.stack([obs[..., 0], obs[..., 1], trig2angle(obs[..., 2], obs[..., 3]), obs[..., 4]], -1)
    if env_name == "oderl-pendulum":
        return torch.stack([trig2angle(obs[..., 0], obs[..., 1]), obs[..., 2]], -1)
    return torch.stack([trig2angle(obs[..., 0], obs[..., 1]), trig2angle(obs[..., 2], obs[..., 3]), obs[..., 4], obs[..., 5]], -1)


def env_reward(env_name, s, a):
    """``diff_reward(s, a)`` on REDUCED states, as ``integrate_system`` evaluates it (base_env.py:164)."""
    if env_name == "oderl-cartpole":  # ctcartpole.py:285-346, 4-D branch, swing_up
        x, xd, th, thd = s[..., 0], s[..., 1], s[..., 2], s[..., 3]
        e0 = x + 1.0 * torch.sin(th) - 0.0
        e1 = 1.0 * torch.cos(th) - 1.0
        return (-(e0 * e0 + e1 * e1) + 0.01 * (-(xd * xd) - thd * thd)) + (-0.01 * (a * a).sum(-1))
    if env_name == "oderl-pendulum":  # ctpendulum.py:139-155
        th, thd = s[..., 0], s[..., 1]
        c, sn = torch.cos(th), torch.sin(th)
        return (-(1.0**2) * ((1 - c) ** 2 + sn**2) + 0.01 * (-(thd**2))) + (-0.01 * (a * a).sum(-1))
    th1, th2, v1, v2 = s[..., 0], s[..., 1], s[..., 2], s[..., 3]  # ctacrobot.py:230-255
    p1x, p1y = -1.0 * torch.cos(th1), 1.0 * torch.sin(th1)
    p2x, p2y = p1x - 1.0 * torch.cos(th1 + th2), p1y + 1.0 * torch.sin(th1 + th2)
    state_reward = -((p2x - 1.0 - 1.0) ** 2) - p2y**2
    return (state_reward + 1e-1 * (-(v1**2) - v2**2)) + (-1e-4 * (a * a).sum(-1))


def env_step(env_name, s, a, dt=0.05, friction=False):
    """One control step on reduced states: (s1, obs1, reward) = Euler step, get_obs, diff_reward(s1, a)."""
    s1 = s + dt * env_rhs(env_name, s, a, friction)
    return s1, env_obs(env_name, s1), env_reward(env_name, s1, a)


def env_reset(env_name, np_random):
    """``reset()`` of the three envs (ctcartpole.py:160-170, ctpendulum.py:92-98, ctacrobot.py:148-151): reduced state."""
    if env_name == "oderl-cartpole":
        st = np_random.uniform(low=-0.05, high=0.05, size=(4,))
        st[2] += np.pi
    elif env_name == "oderl-pendulum":
        st = np_random.uniform(low=-0.1, high=0.1, size=(2,))
        st[0] += np.pi
    else:
        st = np_random.uniform(low=-0.1, high=0.1, size=(4,))
    return torch.as_tensor(st, dtype=torch.float64)
