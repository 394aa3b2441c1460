#!/usr/bin/env python3
"""Generate g9_dtrnn_<env>.npz: the Delta-t RNN baseline pinned against the REFERENCE class.

Runs only in the build container (needs /root/reference).  ``train_utils.py`` does not import here (wandb,
torchdiffeq, gym, pyvirtualdisplay ... are absent), so the ``DeltaTRNN`` class definition alone is taken from the
reference file's syntax tree at run time and executed against torch (nothing of the reference is written into this
repo: the fixture holds inputs and outputs only).  The planner side is the reference ``MPPIDelay`` with the harness
dynamics closure of ``mppi_with_model.py:103-122`` and the real env reward methods, as in g1/g3.

    python tests/golden/make_golden_rnn.py

Fixture contents per env (cartpole, pendulum, acrobot):
  sd_*                  the model's state_dict (reference constructor under torch.manual_seed, linear_out scaled)
  fwd_obs/window/ts/out DeltaTRNN.forward on random inputs, normalize=normalize_time=True
  raw_out               the same inputs through a normalize_time=False model (raw obs, action / 3, raw ts)
  rnnsd_*, rnn_out, rnn_raw_out   the plain RNN baseline (train_utils.py:550-586, hidden 64) on the same inputs,
                        normalize=True / False
  s{0,1}_*              two consecutive reference MPPIDelay.command() calls with the model as dynamics
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

REF = "/root/reference"


def load_reference_class(name):
    """Execute ONE class definition of train_utils.py (the module itself needs packages this container lacks)."""
    src = open(f"{REF}/train_utils.py").read()
    tree = ast.parse(src)
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == name)
    ns = {"torch": torch, "nn": torch.nn, "np": np}
    exec(compile(ast.Module(body=[node], type_ignores=[]), f"{REF}/train_utils.py", "exec"), ns)
    return ns[name]


def main():
    MPPIDelay, _w_nl, envs, _dyn = mg.load_reference_modules()
    DeltaTRNN = load_reference_class("DeltaTRNN")
    RNN = load_reference_class("RNN")
    K, T, B, H = 64, 8, 4, 160
    for env_name, mk in envs.items():
        env = mk()
        st = onl.ENV_STATS[env_name]
        d, nu, A = st["d"], st["nu"], st["act_high"]

        def build(normalize_time, seed):
            torch.manual_seed(seed)
            m = DeltaTRNN(
                d, nu, hidden_units=H, encode_obs_time=False,
                state_mean=np.zeros(d), state_std=np.array(st["state_std"]),
                action_mean=np.array([0] * 1), action_std=np.array([A / 2.0]),
                normalize=True, normalize_time=normalize_time,
            ).double()
            with torch.no_grad():  # "trained-like": small predicted state differences (parity does not depend on it)
                m.linear_out.weight.mul_(0.2)
                m.linear_out.bias.mul_(0.2)
            return m

        model = build(True, 40)
        out = {f"sd_{k}": mg.np_(v) for k, v in model.state_dict().items()}
        g = torch.Generator().manual_seed(41)
        N = 97
        obs = torch.randn(N, d, dtype=torch.double, generator=g) * torch.tensor(st["state_std"])
        window = (torch.rand(N, B, nu, dtype=torch.double, generator=g) * 2 - 1) * A
        ts = torch.rand(N, 1, dtype=torch.double, generator=g) * 0.1 + 0.01
        with torch.no_grad():
            out.update(fwd_obs=mg.np_(obs), fwd_window=mg.np_(window), fwd_ts=mg.np_(ts),
                       fwd_out=mg.np_(model(obs, window, ts)))
            raw_model = build(False, 40)
            out["raw_out"] = mg.np_(raw_model(obs, window, ts))

        ts_pred = torch.tensor(0.05, dtype=torch.double).view(1, 1).repeat(K, 1)  # mppi_with_model.py:74

        def dynamics(state, perturbed_action):  # mppi_with_model.py:103-122 (model_name != "nl": no time channel)
            with torch.no_grad():
                return state + model(state, perturbed_action, ts_pred)

        def running_cost(state, action, env=env):  # mppi_with_model.py:163-164
            return -(env.diff_obs_reward_(state, exp_reward=False) + env.diff_ac_reward_(action))

        torch.manual_seed(42)
        mppi = MPPIDelay(
            dynamics, running_cost, d, mg.noise_sigma(nu), num_samples=K, horizon=T, device="cpu", lambda_=1.0,
            u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A,
        )
        state = oenvs.initial_state(env_name, seed=3)
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
        # the plain RNN baseline (train_utils.py:550-586): forward on both branches
        def build_rnn(normalize, seed):
            torch.manual_seed(seed)
            m = RNN(d, nu, hidden_units=64, encode_obs_time=False, state_mean=np.zeros(d),
                    state_std=np.array(st["state_std"]), action_mean=np.array([0] * 1), action_std=np.array([A / 2.0]),
                    normalize=normalize).double()
            with torch.no_grad():
                m.linear_out.weight.mul_(0.2)
                m.linear_out.bias.mul_(0.2)
            return m

        rnn = build_rnn(True, 43)
        out.update({f"rnnsd_{k}": mg.np_(v) for k, v in rnn.state_dict().items()})
        with torch.no_grad():
            out["rnn_out"] = mg.np_(rnn(obs, window, ts))
            out["rnn_raw_out"] = mg.np_(build_rnn(False, 43)(obs, window, ts))

        np.savez_compressed(f"{HERE}/g9_dtrnn_{env_name.split('-')[1]}.npz", K=K, T=T, B=B, H=H, nx=d, nu=nu, A=A, **out)
        print("g9", env_name, "action", out["s1_action"], "max |dx|", np.abs(out["fwd_out"]).max())


if __name__ == "__main__":
    main()
