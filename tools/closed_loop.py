"""Closed-loop sanity check: 200 control steps of the reference's evaluation loop (mppi_with_model.py:244-317) with
oracle dynamics, the planner on the GPU and the env's Euler step on the host (oracle/envs.py restates it: for the
trig observation the env's one-step Euler integration equals oracle.*_dynamics_dt_delay with delay 0).
Prints episode returns next to the reference's published oracle+MPC returns (process_results/plot_util.py:7-11,21-25)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurallaplacecontrol_amd as nlc
from oracle import envs as oenvs, mppi as omppi

PUBLISHED = {("oderl-cartpole", 0): -139.69, ("oderl-pendulum", 0): -121.05, ("oderl-acrobot", 0): -571.11,
             ("oderl-cartpole", 1): -146.26, ("oderl-pendulum", 1): -123.44, ("oderl-acrobot", 1): -558.76}
K, T, STEPS = 1000, 40, 200
out = []
for (env, delay), pub in PUBLISHED.items():
    nx, nu, A = oenvs.OBS_DIM[env], oenvs.ACT_DIM[env], oenvs.ACTION_HIGH[env]
    rets = []
    t0 = time.perf_counter()
    for seed in range(3):
        mppi = nlc.MPPIDelay(nlc.OracleDynamics(env, 0.05, delay), nlc.EnvCost(env), nx, nlc.noise_sigma(nu), K, T, "cpu",
                             lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox",
                             seed=seed, U_init=torch.zeros(T, nu, dtype=torch.float64))
        obs = oenvs.initial_state(env, seed)
        if env == "oderl-pendulum":  # harness starts the pendulum at [pi, 1]
            obs = torch.tensor([-1.0, 0.0, 1.0], dtype=torch.float64)
        ab = torch.zeros(4, nu, dtype=torch.float64)
        ts = torch.full((1, 1), 0.05, dtype=torch.float64)
        total = 0.0
        for _ in range(STEPS):
            a = mppi.command(obs, ab)
            ab, applied = omppi.get_action(ab, a, delay)
            obs = oenvs.ORACLE_DYNAMICS[env](obs.view(1, -1), applied.view(1, 1, nu), ts, 0).view(-1)
            total += -float(oenvs.RUNNING_COST[env](obs.view(1, -1), applied.view(1, nu)))
        rets.append(total)
    dt = time.perf_counter() - t0
    out.append(dict(env=env, delay=delay, returns=rets, mean=sum(rets) / len(rets), published_reference=pub,
                    ms_per_control_step=dt / (3 * STEPS) * 1e3))
    print(out[-1], flush=True)
print(json.dumps(out))
