#!/usr/bin/env python3
"""G15: BASELINE's literal "state_dim=4" -- cartpole WITHOUT the trig observation (``CTCartpole(obs_trans=False)``,
ctcartpole.py:60: state [x, xdot, theta, thetadot]) -- pinned against the imported reference.  Runs only in the build
container (needs /root/reference); writes data (inputs + expected outputs), never reference source.

    python tests/golden/make_golden_notrig.py

  g15_notrig_cartpole.npz
    o<delay>_*   reference MPPIDelay + reference oracle.cartpole_dynamics_dt_delay on the 4-dim state (its `else` branches,
                 oracle.py:38-44, 80-86) + the REAL env's reward methods with obs_trans=False (the s.shape[-1] == 4 branch,
                 ctcartpole.py:297-300), delays 0 and 2, two consecutive commands
    n_*          reference NeuralLaplaceModel(state_dim=4) behind the harness closure + reference MPPIDelay, two commands
                 (ONLY laplace_reconstruct is the build's restatement, as in G3), plus a model.forward batch
"""

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (stubs + reference loaders; imports the reference at run time)
from oracle import envs as oenvs  # noqa: E402
from oracle import nl_model as onl  # noqa: E402

ENV = "oderl-cartpole-notrig"


def main():
    MPPIDelay, w_nl, envs, dyn = mg.load_reference_modules()
    from envs.oderl.envs import CTCartpole

    env = CTCartpole(dt=0.05, obs_trans=False, device="cpu", solver="euler", friction=False)
    assert env.n == 4, env.n
    nx, nu, A = 4, 1, 3.0
    K, T, B = 64, 8, 4
    out = {}

    def running_cost(state, action, env=env):
        return -(env.diff_obs_reward_(state, exp_reward=False) + env.diff_ac_reward_(action))

    from functools import partial

    ts_pred = torch.full((K, 1), 0.05, dtype=torch.double)
    for delay in (0, 2):
        torch.manual_seed(400 + delay)
        mppi = MPPIDelay(partial(dyn["oderl-cartpole"], ts=ts_pred, delay=delay, friction=False), running_cost, nx, mg.noise_sigma(nu),
                         num_samples=K, horizon=T, device="cpu", lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A)
        state = oenvs.initial_state(ENV, seed=delay)
        action_buffer = (torch.rand(B, nu, dtype=torch.double) - 0.5) * A
        for step in range(2):
            c = mg.capture_command(mppi, state.numpy(), action_buffer)
            for k, v in c.items():
                out[f"o{delay}_s{step}_{k}"] = v
            out[f"o{delay}_s{step}_state"] = mg.np_(state)
            out[f"o{delay}_s{step}_action_buffer"] = mg.np_(action_buffer)
            state = mppi.states[0, 0].clone()
            action_buffer = torch.roll(action_buffer, -1, dims=0)
            action_buffer[-1] = torch.as_tensor(c["action"])
        print("g15 oracle delay", delay, "action", out[f"o{delay}_s1_action"])

    # ---- NL dynamics at d = 4 with the reference's own model class
    st = onl.ENV_STATS[ENV]
    S = 17
    torch.manual_seed(0)
    model = w_nl.NeuralLaplaceModel(4, 1, 4, hidden_units=128, s_recon_terms=S, ilt_algorithm="fourier", encode_obs_time=False,
                                    state_mean=np.zeros(4), state_std=np.array(st["state_std"]), action_mean=np.array([0]),
                                    action_std=np.array([A / 2.0]), normalize=True, normalize_time=True).double()
    mine = onl.make_synthetic_state_dict(0, 4, 1, 128, S, state_std=st["state_std"], action_std=[A / 2.0])
    for k, v in model.state_dict().items():
        assert np.array_equal(mg.np_(v), mg.np_(mine[k])), f"synthetic weights differ from the reference ctor: {k}"
    with torch.no_grad():
        model.laplace_rep_func.linear_tanh_stack[4].bias[4 * S:] += onl.PHI_BIAS_SHIFT
        sd = {k: mg.np_(v) for k, v in model.state_dict().items()}
        torch.manual_seed(7)
        N = 48
        obs = torch.randn(N, 4, dtype=torch.double) * torch.tensor(st["state_std"])
        window = (torch.rand(N, B, nu, dtype=torch.double) * 2 - 1) * A
        ts = torch.full((N, 1), 0.05, dtype=torch.double)
        out.update(fwd_obs=mg.np_(obs), fwd_window=mg.np_(window), fwd_ts=mg.np_(ts), fwd_out=mg.np_(model(obs, window, ts)))

        def dynamics(state, perturbed_action):
            return state + model(state, perturbed_action, ts_pred)

        torch.manual_seed(11)
        mppi = MPPIDelay(dynamics, running_cost, 4, mg.noise_sigma(nu), num_samples=K, horizon=T, device="cpu", lambda_=1.0,
                         u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A)
        state = oenvs.initial_state(ENV, seed=3)
        action_buffer = torch.zeros(B, nu, dtype=torch.double)
        for step in range(2):
            c = mg.capture_command(mppi, state.numpy(), action_buffer)
            for k, v in c.items():
                out[f"n_s{step}_{k}"] = v
            out[f"n_s{step}_state"] = mg.np_(state)
            out[f"n_s{step}_action_buffer"] = mg.np_(action_buffer)
            state = mppi.states[0, 0].clone()
            action_buffer = torch.roll(action_buffer, -1, dims=0)
            action_buffer[-1] = torch.as_tensor(c["action"])
        print("g15 nl action", out["n_s1_action"])
    np.savez_compressed(f"{HERE}/g15_notrig_cartpole.npz", K=K, T=T, B=B, nx=nx, nu=nu, A=A, S=S,
                        **out, **{f"w::{k}": v for k, v in sd.items()})


if __name__ == "__main__":
    main()
