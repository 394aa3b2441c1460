"""configs[4] (cartpole, de Hoog S = 33, K = 16384, H = 40): the persistent step-chain kernel (`dehoog_chain` 1) against the
staged 2 T + 1 launches on one / two streams (`dehoog_chain` 0, `dehoog_streams` 1 / 2), interleaved on one box: wall time
per command and the per-kernel event times of one profiled pass.  python tools/cfg5_chain_ab.py [K]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurallaplacecontrol_amd as nlc

K = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
env, d, nu, A, T, S = "oderl-cartpole", 5, 1, 3.0, 40, 33
std = [2.88646771, 11.54556671, 0.70729307, 0.70692035, 17.3199048]
torch.manual_seed(0)
model = nlc.NeuralLaplaceModel(d, nu, d, hidden_units=128, s_recon_terms=S, ilt_algorithm="dehoog", state_mean=np.zeros(d),
                               state_std=np.array(std), action_mean=np.array([0]), action_std=np.array([A / 2.0]),
                               normalize=True, normalize_time=True).double()
with torch.no_grad():
    model.laplace_rep_func.linear_tanh_stack[4].bias[d * S:] += -3.0
model = model.to("cuda")
variants = {"auto_measured_choice": {}, "chain": {"dehoog_chain": 1}, "chain_32_samples_two_per_cu": {"dehoog_chain": 2}, "staged_2_streams": {"dehoog_chain": 0, "dehoog_streams": 2},
            "staged_1_stream": {"dehoog_chain": 0, "dehoog_streams": 1},
            # timing breakdown of the chain kernel (results meaningless): only its representation phase / only its QD phase
            "chain_phase_A_only": {"dehoog_chain": 1, "dehoog_chain_phases": 1},
            "chain_phase_B_only": {"dehoog_chain": 1, "dehoog_chain_phases": 2}}
planners = {name: nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, nlc.noise_sigma(nu), K, T, "cpu", lambda_=1.0,
                                u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox", seed=0,
                                U_init=torch.zeros(T, nu, dtype=torch.float64), store_rollouts=False, planner_options=o)
            for name, o in variants.items()}
st, ab = nlc.initial_state(env, torch.Generator().manual_seed(0)), torch.zeros(4, nu, dtype=torch.float64)
out = {n: [] for n in planners}
with torch.no_grad():
    for p in planners.values():
        for _ in range(130):  # (auto takes its candidate forms in turns for at least half a second before it settles)
            p.command(st, ab)
    for rep in range(3):
        for name, p in planners.items():
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20):
                p.command(st, ab)
            torch.cuda.synchronize()
            out[name].append((time.perf_counter() - t0) / 20 * 1e3)
    prof = {}
    for name, p in planners.items():
        p.ctx.profile_reset(); p.ctx.profile(True)
        for _ in range(5):
            p.command(st, ab)
        torch.cuda.synchronize(); p.ctx.profile(False)
        prof[name] = {k: round(v["total_ms"] / 5, 4) for k, v in p.ctx.profile_read().items()}
res = {n: dict(ms_per_step=[round(x, 4) for x in v], best_ms=round(min(v), 4), steps_per_s=round(1e3 / min(v), 1), kernels_ms_per_command=prof[n])
       for n, v in out.items()}
print(json.dumps(dict(K=K, T=T, S=S, results=res)))
