#!/usr/bin/env python3
"""Generate g11_node_<env>.npz: the NODE baseline from the REFERENCE classes.

Runs only in the build container (needs /root/reference).  As for G9, ``train_utils.py`` does not import here, so the
two class definitions ``xOdeFuncInXAndU`` and ``NODE`` alone are taken from the reference file's syntax tree at run
time and executed against torch.  ``NODE.forward`` calls ``torchdiffeq.odeint``, which is absent: the name ``odeint``
is bound to the build's restatement of torchdiffeq's fixed-grid Euler solver (``oracle/node_model.py::odeint_euler`` --
**parity unpinned for odeint**); every other operation in the fixture (normalisation, augmentation, the MLP, the
action pick, the output slice, the planner) is the reference's own code.

    python tests/golden/make_golden_node.py
"""

import ast
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

from oracle import envs as oenvs  # noqa: E402
from oracle import nl_model as onl  # noqa: E402
from oracle import node_model as onode  # noqa: E402

REF = "/root/reference"


def load_reference_classes(names):
    tree = ast.parse(open(f"{REF}/train_utils.py").read())
    ns = {"torch": torch, "nn": torch.nn, "np": np, "odeint": onode.odeint_euler, "device": torch.device("cpu")}
    for name in names:
        node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == name)
        exec(compile(ast.Module(body=[node], type_ignores=[]), f"{REF}/train_utils.py", "exec"), ns)
    return [ns[n] for n in names]


def main():
    MPPIDelay, _w_nl, envs, _dyn = mg.load_reference_modules()
    _func_cls, NODE = load_reference_classes(["xOdeFuncInXAndU", "NODE"])
    K, T, B, H, AUG = 64, 8, 4, 270, 1
    for env_name, mk in envs.items():
        env = mk()
        st = onl.ENV_STATS[env_name]
        d, nu, A = st["d"], st["nu"], st["act_high"]
        torch.manual_seed(50)
        model = NODE(
            d, nu, d, hidden_units=H, state_mean=np.zeros(d), state_std=np.array(st["state_std"]),
            action_mean=np.array([0] * 1), action_std=np.array([A / 2.0]), normalize=True, normalize_time=True,
            encode_obs_time=False, method="euler", augment_dim=AUG,
        ).double()
        with torch.no_grad():  # "trained-like": moderate state changes per step (parity does not depend on it)
            model.x_ode_func_in_x_and_u.linear_tanh_stack[4].weight.mul_(0.3)
            model.x_ode_func_in_x_and_u.linear_tanh_stack[4].bias.mul_(0.3)
        out = {f"sd_{k}": mg.np_(v) for k, v in model.state_dict().items()}
        g = torch.Generator().manual_seed(51)
        N = 97
        obs = torch.randn(N, d, dtype=torch.double, generator=g) * torch.tensor(st["state_std"])
        window = (torch.rand(N, B, nu, dtype=torch.double, generator=g) * 2 - 1) * A
        with torch.no_grad():
            for tag, tval in (("", 0.05), ("t2_", 0.11)):  # 0.11 / 0.4 = 0.275: six Euler sub-steps
                ts = torch.full((N, 1), tval, dtype=torch.double)
                out[f"fwd_{tag}ts"] = mg.np_(ts)
                out[f"fwd_{tag}out"] = mg.np_(model(obs, window, ts))
            out.update(fwd_obs=mg.np_(obs), fwd_window=mg.np_(window))
            x = torch.randn(N, d + AUG, dtype=torch.double, generator=g)
            model.x_ode_func_in_x_and_u.update_u(window[:, -1, :])
            out.update(func_x=mg.np_(x), func_out=mg.np_(model.x_ode_func_in_x_and_u(None, x)))
        ts_pred = torch.tensor(0.05, dtype=torch.double).view(1, 1).repeat(K, 1)

        def dynamics(state, perturbed_action):  # mppi_with_model.py:103-122
            with torch.no_grad():
                return state + model(state, perturbed_action, ts_pred)

        def running_cost(state, action, env=env):
            return -(env.diff_obs_reward_(state, exp_reward=False) + env.diff_ac_reward_(action))

        torch.manual_seed(52)
        mppi = MPPIDelay(
            dynamics, running_cost, d, mg.noise_sigma(nu), num_samples=K, horizon=T, device="cpu", lambda_=1.0,
            u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A,
        )
        state = oenvs.initial_state(env_name, seed=4)
        action_buffer = (torch.rand(B, nu, dtype=torch.double) - 0.5) * A
        for step in range(2):
            c = mg.capture_command(mppi, state.numpy(), action_buffer)
            for k, v in c.items():
                out[f"s{step}_{k}"] = v
            out[f"s{step}_state"] = mg.np_(state)
            out[f"s{step}_action_buffer"] = mg.np_(action_buffer)
            state = mppi.states[0, 0].clone()
            action_buffer = torch.roll(action_buffer, -1, dims=0)
            action_buffer[-1] = torch.as_tensor(c["action"])
        np.savez_compressed(f"{HERE}/g11_node_{env_name.split('-')[1]}.npz", K=K, T=T, B=B, H=H, AUG=AUG, nx=d, nu=nu,
                            A=A, **out)
        print("g11", env_name, "action", out["s1_action"], "max |out|", np.abs(out["fwd_out"]).max(),
              "state range", np.abs(out["s1_states"]).max())


if __name__ == "__main__":
    main()
