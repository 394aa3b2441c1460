"""Aggregate planning throughput of the expert-data collector's workload:
oracle dynamics, K = 1000 samples, T = 40 (config.py:21-23; mppi_dataset_collector.py:224-321), E episodes planned
side by side by BatchedMPPIDelay vs one MPPIDelay command at a time.

The env step between commands is the oracle Euler step on the device (a stand-in: the reference env integrates
with torchdiffeq, which is not part of the planner path); states / action buffers never leave the GPU.

    python tools/collector_bench.py [--env oderl-cartpole] [--episodes 1,16,64,256,1024] [--steps 20]
"""

import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))

import neurallaplacecontrol_amd as nlc  # noqa: E402

ENV = {"oderl-cartpole": (5, 1, 3.0), "oderl-pendulum": (3, 1, 2.0), "oderl-acrobot": (6, 2, 5.0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="oderl-cartpole")
    ap.add_argument("--episodes", default="1,16,64,256,1024")
    ap.add_argument("--samples", type=int, default=1000)
    ap.add_argument("--horizon", type=int, default=40)
    ap.add_argument("--delay", type=int, default=2)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--dynamics", default="oracle", choices=["oracle", "nl"])
    a = ap.parse_args()
    nx, nu, A = ENV[a.env]
    sig = nlc.noise_sigma(nu)
    kw = dict(lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox", seed=0,
              store_rollouts=False)
    if a.dynamics == "nl":
        if a.env != "oderl-cartpole":
            raise SystemExit("--dynamics nl uses bench.py's synthetic cartpole model")
        import bench  # the synthetic 'trained-like' cartpole model of the headline bench

        model = bench.synthetic_state_dict(nx, nu, 17).cuda()
        make_dyn = lambda: nlc.NLDynamics(model, 0.05)  # noqa: E731
    else:
        make_dyn = lambda: nlc.OracleDynamics(a.env, 0.05, a.delay)  # noqa: E731
    out = []
    for E in [int(x) for x in a.episodes.split(",")]:
        g = torch.Generator().manual_seed(E)
        states = torch.stack([nlc.initial_state(a.env, g) for _ in range(E)]).cuda()
        abuf = torch.zeros(E, 4, nu, dtype=torch.float64, device="cuda")
        if E == 1:
            pl = nlc.MPPIDelay(make_dyn(), nlc.EnvCost(a.env), nx, sig, a.samples, a.horizon, "cuda", **kw)
            cmd = lambda: pl.command(states[0].cpu(), abuf[0].cpu())  # noqa: E731  (the reference hands host state over)
        else:
            pl = nlc.BatchedMPPIDelay(make_dyn(), nlc.EnvCost(a.env), nx, sig, E, a.samples, a.horizon, "cuda", **kw)
            cmd = lambda: pl.command(states, abuf)  # noqa: E731
        with torch.no_grad():
            for _ in range(3):
                act = cmd()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                act = cmd()
                # harness get_action (mppi_with_model.py:25-28) on the device
                abuf = torch.roll(abuf, -1, dims=-2)
                abuf[..., -1, :] = act.reshape(abuf[..., -1, :].shape)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.steps
        out.append({"episodes": E, "ms_per_command": dt * 1e3, "planning_steps_per_s": E / dt})
        print(json.dumps(out[-1]), flush=True)
    print(json.dumps({"workload": f"{a.env} {a.dynamics} K={a.samples} T={a.horizon} delay={a.delay}", "rows": out}))


if __name__ == "__main__":
    main()
