"""Planning steps/s with the Delta-t RNN baseline (train_utils.py:589-631, rnn_hidden_units=160) as the dynamics:
BASELINE configs[1] sizes (cartpole, K=16384, T=40, B=4), device Philox noise, plus the CPU oracle on a bounded
sample.  Prints one JSON line (profiles/r1j_dtrnn_planner.json)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurallaplacecontrol_amd as nlc
from oracle import envs as oenvs, mppi as omppi, nl_model as onl, rnn_model as ornn

env, K, T, B, H = "oderl-cartpole", int(os.environ.get("K", 16384)), 40, 4, int(os.environ.get("H", 160))
st = onl.ENV_STATS[env]
d, nu, A = st["d"], st["nu"], st["act_high"]
sd = ornn.make_synthetic_state_dict(0, d, nu, H, st["state_std"], [A / 2.0])
model = nlc.DeltaTRNN(d, nu, hidden_units=H, state_mean=np.zeros(d), state_std=np.ones(d), action_mean=np.array([0]),
                      action_std=np.array([1.0]), normalize=True, normalize_time=True).double()
model.load_state_dict(sd)
model = model.cuda()
mppi = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, nlc.noise_sigma(nu), num_samples=K, horizon=T,
                     device="cuda", lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A,
                     noise_rng="philox", seed=0, store_rollouts=False)
state = oenvs.initial_state(env, seed=0).numpy()
ab = torch.zeros(B, nu, dtype=torch.float64)
for _ in range(5):
    mppi.command(state, ab)
mppi.ctx.profile_reset(); mppi.ctx.profile(True)
torch.cuda.synchronize(); t0 = time.perf_counter()
steps = 50
for _ in range(steps):
    mppi.command(state, ab)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
mppi.ctx.profile(False)
prof = {k: v["total_ms"] / v["launches"] for k, v in mppi.ctx.profile_read().items() if v["launches"]}
GT, KS = H // 16, H // 4
mfma = B * 3 * GT + (B - 1) * 3 * GT * KS + KS
flops = mfma * 2048 * (K * T / 16)
enc_ms = prof["rnn_encode_kernel"]
# CPU oracle (reference op sequence) on a bounded sample: K/16 samples of the same problem
Kc = max(K // 16, 64)
torch.set_num_threads(min(32, os.cpu_count() or 1))
dyn, cost = ornn.make_dynamics(sd), oenvs.RUNNING_COST[env]
U = torch.zeros(T, nu, dtype=torch.float64)
noise = torch.randn(Kc, T, nu, dtype=torch.float64)
t0 = time.perf_counter()
omppi.mppi_command(U, torch.as_tensor(state), ab, noise, dyn, cost, d, torch.inverse(nlc.noise_sigma(nu)), 1.0, A,
                   torch.tensor(-A), torch.tensor(A))
cpu_s = time.perf_counter() - t0
print(json.dumps({
    "metric": "MPPI planning steps/sec, Delta-t RNN dynamics (rnn_hidden_units=%d)" % H, "value": 1.0 / dt,
    "unit": "planning steps/s", "ms_per_step": dt * 1e3, "dtype": "f64",
    "config": {"workload": f"oderl-cartpole, K={K}, H=40, action_buffer_size=4, DeltaTRNN hidden {H}", "noise": "device Philox"},
    "kernels_avg_ms": prof,
    "roofline": {"bound": "mfma", "kernel": "rnn_encode_kernel", "avg_launch_ms": enc_ms, "flops_per_launch": flops,
                 "achieved": flops / enc_ms / 1e9, "peak": 78.6, "unit": "TFLOP/s", "frac": flops / enc_ms / 1e9 / 78.6},
    "cpu_baseline": {"value": 1.0 / (cpu_s * K / Kc), "unit": "planning steps/s", "kind": "port", "cores": torch.get_num_threads(),
                     "sample": f"one command() of the oracle at K={Kc} ({cpu_s:.2f} s), scaled linearly to K={K}"},
}))
