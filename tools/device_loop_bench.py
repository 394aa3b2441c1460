"""Closed-loop evaluation entirely on the device: E episodes x 200 control steps of the reference's
evaluation loop (mppi_with_model.py:244-317) with oracle dynamics, K = 1000, T = 40 -- BatchedMPPIDelay + BatchedEnv,
no host round trip per control step -- next to the same loop with ONE env stepped on the host per command (the
reference's structure).  Prints one JSON line (profiles/r1j_device_loop.json)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurallaplacecontrol_amd as nlc
from oracle import envs as oenvs, mppi as omppi

PUBLISHED = {("oderl-cartpole", 0): -139.69, ("oderl-pendulum", 0): -121.05, ("oderl-acrobot", 0): -571.11,
             ("oderl-cartpole", 1): -146.26, ("oderl-pendulum", 1): -123.44, ("oderl-acrobot", 1): -558.76}
K, T, STEPS, B = 1000, 40, 200, 4
E = int(os.environ.get("E", 256))
out = []
for (env, delay), pub in PUBLISHED.items():
    nx, nu, A = nlc.envs.ENV_DIMS[env]
    mppi = nlc.BatchedMPPIDelay(nlc.OracleDynamics(env, 0.05, delay), nlc.EnvCost(env), nx, nlc.noise_sigma(nu), E, K, T,
                                "cuda", lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A,
                                noise_rng="philox", seed=0, U_init=torch.zeros(E, T, nu, dtype=torch.float64),
                                store_rollouts=False)
    envs = nlc.BatchedEnv(env, E, dt=0.05, action_delay=delay, action_buffer_size=B, seed=0)
    obs = envs.reset(harness_start=True)
    total = torch.zeros(E, dtype=torch.float64, device="cuda")
    for _ in range(3):  # warm-up commands on a throw-away copy of the state
        mppi.command(obs, envs.action_buffer)
    mppi.U = torch.zeros(E, T, nu, dtype=torch.float64)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(STEPS):
        act = mppi.command(obs, envs.action_buffer)
        obs, rew = envs.step(act)
        total += rew
    torch.cuda.synchronize(); dt_dev = time.perf_counter() - t0
    rets = total.cpu().numpy()
    # the reference's structure: one env, one planner, env stepped on the host every control step (2 episodes)
    t0 = time.perf_counter()
    host_rets = []
    for seed in range(2):
        m1 = nlc.MPPIDelay(nlc.OracleDynamics(env, 0.05, delay), nlc.EnvCost(env), nx, nlc.noise_sigma(nu), K, T, "cpu",
                           lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox",
                           seed=seed, U_init=torch.zeros(T, nu, dtype=torch.float64))
        s = oenvs.env_reset(env, np.random.RandomState(seed))
        if env == "oderl-pendulum":
            s = torch.tensor([np.pi, 1.0], dtype=torch.float64)
        ab = torch.zeros(B, nu, dtype=torch.float64)
        tot = 0.0
        for _ in range(STEPS):
            a = m1.command(oenvs.env_obs(env, s), ab)
            ab, at = omppi.get_action(ab, a, delay)
            s, _, r = oenvs.env_step(env, s, at.clone(), 0.05)
            tot += float(r)
        host_rets.append(tot)
    dt_host = (time.perf_counter() - t0) / 2
    out.append(dict(env=env, delay=delay, episodes=E, mean_return=float(rets.mean()), std_return=float(rets.std()),
                    published_reference_oracle_mpc=pub, device_loop_s=dt_dev,
                    device_control_steps_per_s=E * STEPS / dt_dev, device_episodes_per_s=E / dt_dev,
                    host_stepped_single_env_control_steps_per_s=STEPS / dt_host, host_stepped_returns=host_rets))
    print(out[-1], file=sys.stderr, flush=True)
print(json.dumps(dict(metric="closed-loop control steps/s (planner + env), oracle dynamics, K=1000, T=40, 200 steps/episode",
                      results=out)))
