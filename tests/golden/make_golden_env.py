#!/usr/bin/env python3
"""Generate g10_env_<env>.npz: the env side of the evaluation loop from the REAL env classes.

Runs only in the build container (needs /root/reference).  One control step of the harness
(``mppi_with_model.py:193-216`` ``step_env``) is ``get_action`` (delay buffer, ``:25-28``) + ``env.integrate_system(2, g,
s0)`` (``base_env.py:136-173``) + ``env.get_obs()``.  With the harness's ``solver="euler"`` (``overlay.py:39``) and the
fixed grid ``ts = [0, dt]`` the ``odeint`` call is ONE explicit Euler step of ``env.torch_rhs`` on the reduced state
(``torchdiffeq`` itself is absent here; that single formula ``s + dt * rhs(s, a)`` is the only restated arithmetic, every
rhs / observation / reward value below comes from the reference's own methods).

    python tests/golden/make_golden_env.py
"""

import ast
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

REF = "/root/reference"
DT = 0.05
RANGES = {  # reduced-state sampling ranges (beyond +-pi on purpose)
    "oderl-cartpole": [2.0, 3.0, 3.8, 6.0],
    "oderl-pendulum": [4.0, 8.0],
    "oderl-acrobot": [3.5, 3.5, 6.0, 6.0],
}


def load_reference_function(path, name):
    tree = ast.parse(open(path).read())
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    ns = {"torch": torch, "np": np}
    exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    return ns[name]


def main():
    _MPPIDelay, _w_nl, envs, _dyn = mg.load_reference_modules()
    get_action = load_reference_function(f"{REF}/mppi_with_model.py", "get_action")
    for env_name, mk in envs.items():
        env = mk()
        A = float(env.action_space.high[0])
        nu = env.action_space.shape[0]
        g = torch.Generator().manual_seed(77)
        rng = torch.tensor(RANGES[env_name], dtype=torch.double)
        E = 48
        s0 = (torch.rand(E, rng.numel(), dtype=torch.double, generator=g) * 2 - 1) * rng
        a = (torch.rand(E, nu, dtype=torch.double, generator=g) * 2 - 1) * A * 1.3
        out = dict(E=E, n=rng.numel(), nu=nu, A=A, dt=DT)

        def one(tag, env):
            with torch.no_grad():
                rhs = env.torch_rhs(s0, a)
                s1 = s0 + DT * rhs  # odeint(method="euler") over ts = [0, dt]
                out.update({
                    f"{tag}s0": mg.np_(s0), f"{tag}a": mg.np_(a), f"{tag}rhs": mg.np_(rhs), f"{tag}s1": mg.np_(s1),
                    f"{tag}obs0": mg.np_(env.torch_transform_states(s0)),
                    f"{tag}obs1": mg.np_(env.torch_transform_states(s1)),
                    f"{tag}reward": mg.np_(env.diff_reward(s1, a)),
                    f"{tag}back": mg.np_(env.obs2state(env.torch_transform_states(s1))),
                })

        one("", env)
        if env_name == "oderl-cartpole":
            envf = mk()
            envf.friction = True
            one("fr_", envf)
        # closed-loop trace of ONE env: the harness's get_action + Euler step + reward, delay 2, 4-row action buffer
        delay, B, steps = 2, 4, 9
        acts = (torch.rand(steps, nu, dtype=torch.double, generator=g) * 2 - 1) * A
        ab = torch.zeros((B, nu), dtype=torch.double)
        s = s0[0].clone()
        tr = dict(s=[], obs=[], rew=[], ab=[], at=[])
        with torch.no_grad():
            for i in range(steps):
                ab, at = get_action(ab, acts[i], action_delay=delay)
                at = at.clone()
                s = s + DT * env.torch_rhs(s, at)
                tr["s"].append(mg.np_(s)); tr["obs"].append(mg.np_(env.torch_transform_states(s.unsqueeze(0))[0]))
                tr["rew"].append(mg.np_(env.diff_reward(s, at))); tr["ab"].append(mg.np_(ab)); tr["at"].append(mg.np_(at))
        out.update(loop_delay=delay, loop_B=B, loop_actions=mg.np_(acts),
                   **{f"loop_{k}": np.stack(v) for k, v in tr.items()})
        # the reset distribution (numpy RandomState stream of the env, seeded)
        env.seed(5)
        env.reset()
        out["reset_seed5_state"] = np.asarray(env.state, dtype=np.float64)
        out["reset_seed5_obs"] = np.asarray(env.get_obs(), dtype=np.float64)
        np.savez_compressed(f"{HERE}/g10_env_{env_name.split('-')[1]}.npz", **out)
        print("g10", env_name, "reward[0:3]", out["reward"][:3], "loop rew", out["loop_rew"][:3])


if __name__ == "__main__":
    main()
