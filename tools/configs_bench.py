"""Planning steps/s of the five BASELINE.json configs on ONE MI355X (NL dynamics, synthetic tamed weights, device Philox
noise): at each config's full population and at its per-GPU share (configs[2]: 2 GPUs, configs[3]: 8 GPUs).  The
headline metric is configs[1] (bench.py); this table is context for the others.  One JSON document on stdout
(profiles/r1k_configs.json)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurallaplacecontrol_amd as nlc

STATS = {  # train_utils.py:187-200
    "oderl-cartpole": (5, 1, 3.0, [2.88646771, 11.54556671, 0.70729307, 0.70692035, 17.3199048]),
    "oderl-pendulum": (3, 1, 2.0, [0.70634571, 0.70784512, 2.89072771]),
    "oderl-acrobot": (6, 2, 5.0, [0.70711024, 0.70710328, 0.7072186, 0.7069949, 2.88642115, 2.88627309]),
    "oderl-cartpole-notrig": (4, 1, 3.0, [2.88646771, 11.54556671, 1.81379936, 17.3199048]),  # obs_trans=False (ctcartpole.py:60)
}
CONFIGS = [  # (label, env, delay -> action_buffer rows, K, T, gpus, ilt, S)
    ("configs[0] cartpole d=0 K=1024 H=20", "oderl-cartpole", 4, 1024, 20, 1, "fourier", 17),
    ("configs[1] cartpole d=2 K=16384 H=40", "oderl-cartpole", 4, 16384, 40, 1, "fourier", 17),
    ("configs[2] pendulum d=4 K=65536 H=40 (2 GPUs)", "oderl-pendulum", 5, 65536, 40, 2, "fourier", 17),
    ("configs[3] acrobot d=2 K=262144 H=60 (8 GPUs)", "oderl-acrobot", 4, 262144, 60, 8, "fourier", 17),
    ("configs[4] cartpole K=16384 H=40 de Hoog S=33", "oderl-cartpole", 4, 16384, 40, 1, "dehoog", 33),
    # the other leg of configs[4]'s "de Hoog (33 terms) vs FKT ablation": the Fourier series at the same 33 terms, same shape
    # north_star's literal synthetic shape: state_dim = 4 -- the reference's cartpole without the trig observation
    ("north_star literal: cartpole obs_trans=False (state_dim=4) K=16384 H=40", "oderl-cartpole-notrig", 4, 16384, 40, 1, "fourier", 17),
    ("configs[4] ablation leg: cartpole K=16384 H=40 fourier S=33", "oderl-cartpole", 4, 16384, 40, 1, "fourier", 33),
]


# the other closed-form values of the reference's nl_ilt_algorithm knob (config.py:36) at configs[1]'s shape: the LIN instances of
# the rollout kernels (round 3, default at hidden width 128), the staged all-HIP path (other widths; option linear_fused = 0)
# and the generic path (model.forward as the dynamics callable)
EXTRA = [
    ("cartpole K=16384 H=40 fixed_tablot S=17", "oderl-cartpole", 4, 16384, 40, "fixed_tablot", 17),
    ("cartpole K=16384 H=40 stehfest S=16", "oderl-cartpole", 4, 16384, 40, "stehfest", 16),
]


def rate(env, B, K, T, algo, S, steps, generic=False, opts=None):
    d, nu, A, std = STATS[env]
    torch.manual_seed(0)
    model = nlc.NeuralLaplaceModel(d, nu, d, hidden_units=128, s_recon_terms=S, ilt_algorithm=algo, state_mean=np.zeros(d),
                                   state_std=np.array(std), action_mean=np.array([0]), action_std=np.array([A / 2.0]),
                                   normalize=True, normalize_time=True).double()
    with torch.no_grad():
        model.laplace_rep_func.linear_tanh_stack[4].bias[d * S:] += -3.0  # "trained-like" taming
    model = model.to("cuda")
    dyn = nlc.NLDynamics(model, 0.05)
    mppi = nlc.MPPIDelay((lambda s_, w_: dyn(s_, w_)) if generic else dyn, nlc.EnvCost(env), d, nlc.noise_sigma(nu), K, T, "cuda", lambda_=1.0,
                         u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox",
                         U_init=torch.zeros(T, nu, dtype=torch.float64), store_rollouts=False,
                         planner_options=dict({"recognise_closures": 0}, **(opts or {})))
    assert mppi.fused != generic
    st, ab = nlc.initial_state(env, torch.Generator().manual_seed(0)), torch.zeros(B, nu, dtype=torch.float64)
    with torch.no_grad():
        for _ in range(2):
            mppi.command(st, ab)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            mppi.command(st, ab).cpu()
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    del mppi, model
    torch.cuda.empty_cache()
    return dict(K=K, ms_per_step=dt * 1e3, steps_per_s=1.0 / dt, sample_steps_per_s=K * T / dt)


out = []
only_extra = "--extra-only" in sys.argv
for label, env, B, K, T, G, algo, S in ([] if only_extra else CONFIGS):
    row = dict(config=label, full_population_one_gpu=rate(env, B, K, T, algo, S, 10 if K > 100000 else 20))
    if G > 1:
        row[f"per_gpu_share_K_over_{G}"] = rate(env, B, K // G, T, algo, S, 20)
    out.append(row)
    print(row, file=sys.stderr, flush=True)
for label, env, B, K, T, algo, S in EXTRA:
    row = dict(config=label, rollout_kernels_lin_instances=rate(env, B, K, T, algo, S, 20),
               staged_hip_path=rate(env, B, K, T, algo, S, 20, opts={"linear_fused": 0}),
               generic_path=rate(env, B, K, T, algo, S, 5, generic=True))
    out.append(row)
    print(row, file=sys.stderr, flush=True)
print(json.dumps(dict(metric="MPPI planning steps/s per BASELINE config, one MI355X, f64", results=out)))
