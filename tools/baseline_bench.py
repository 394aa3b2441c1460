"""Planning steps/s with a baseline model as the dynamics -- MODEL=dtrnn: the Delta-t RNN (train_utils.py:589-631,
rnn_hidden_units=160); MODEL=node: the NODE (train_utils.py:637-738, node_hidden_units=270, augment 1, Euler) -- at
BASELINE configs[1] sizes (cartpole, K=16384, T=40, B=4), device Philox noise, plus the CPU oracle on a bounded
sample.  Prints one JSON line (profiles/r1j_dtrnn_planner.json, profiles/r1j_node_planner.json)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurallaplacecontrol_amd as nlc
from oracle import envs as oenvs, mppi as omppi, nl_model as onl, node_model as onode, rnn_model as ornn

MODEL = os.environ.get("MODEL", "dtrnn")

env, K, T, B = "oderl-cartpole", int(os.environ.get("K", 16384)), 40, 4
H = int(os.environ.get("H", 160 if MODEL == "dtrnn" else 270))
st = onl.ENV_STATS[env]
d, nu, A = st["d"], st["nu"], st["act_high"]
if MODEL == "dtrnn":
    sd = ornn.make_synthetic_state_dict(0, d, nu, H, st["state_std"], [A / 2.0])
    model = nlc.DeltaTRNN(d, nu, hidden_units=H, state_mean=np.zeros(d), state_std=np.ones(d), action_mean=np.array([0]),
                          action_std=np.array([1.0]), normalize=True, normalize_time=True).double()
    make_dyn, kern = ornn.make_dynamics, "rnn_encode_kernel"
    GT, KS = H // 16, H // 4
    mfma_per_tile_task = B * 3 * GT + (B - 1) * 3 * GT * KS + KS  # per 16 windows
    tile_tasks = K * T / 16
else:
    sd = onode.make_synthetic_state_dict(0, d, nu, H, 1, st["state_std"], [A / 2.0])
    model = nlc.NODE(d, nu, d, hidden_units=H, state_mean=np.zeros(d), state_std=np.ones(d), action_mean=np.array([0]),
                     action_std=np.array([1.0]), normalize=True, normalize_time=True, method="euler",
                     augment_dim=1).double()
    make_dyn, kern = onode.make_dynamics, "node_rollout_kernel"
    HT = 4 if H <= 64 else (8 if H <= 128 else 17)
    nsub = len(onode.euler_substeps(0.05 / (float(sd["dt"]) * 8.0)))
    mfma_per_tile_task = nsub * (3 * HT + 4 * HT * HT + 4 * HT)  # per 16 samples and horizon step
    tile_tasks = K * T / 16
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
flops = mfma_per_tile_task * 2048 * tile_tasks
enc_ms = prof[kern]
# CPU oracle (reference op sequence) on a bounded sample: K/16 samples of the same problem
Kc = max(K // 16, 64)
torch.set_num_threads(min(32, os.cpu_count() or 1))
dyn, cost = make_dyn(sd), oenvs.RUNNING_COST[env]
U = torch.zeros(T, nu, dtype=torch.float64)
noise = torch.randn(Kc, T, nu, dtype=torch.float64)
t0 = time.perf_counter()
omppi.mppi_command(U, torch.as_tensor(state), ab, noise, dyn, cost, d, torch.inverse(nlc.noise_sigma(nu)), 1.0, A,
                   torch.tensor(-A), torch.tensor(A))
cpu_s = time.perf_counter() - t0
print(json.dumps({
    "metric": "MPPI planning steps/sec, %s dynamics (hidden_units=%d)" % ({"dtrnn": "Delta-t RNN", "node": "NODE"}[MODEL], H), "value": 1.0 / dt,
    "unit": "planning steps/s", "ms_per_step": dt * 1e3, "dtype": "f64",
    "config": {"workload": f"oderl-cartpole, K={K}, H=40, action_buffer_size=4, {MODEL} hidden {H}", "noise": "device Philox"},
    "kernels_avg_ms": prof,
    "roofline": {"bound": "mfma", "kernel": kern, "avg_launch_ms": enc_ms, "flops_per_launch": flops,
                 "achieved": flops / enc_ms / 1e9, "peak": 78.6, "unit": "TFLOP/s", "frac": flops / enc_ms / 1e9 / 78.6},
    "cpu_baseline": {"value": 1.0 / (cpu_s * K / Kc), "unit": "planning steps/s", "kind": "port", "cores": torch.get_num_threads(),
                     "sample": f"one command() of the oracle at K={Kc} ({cpu_s:.2f} s), scaled linearly to K={K}"},
}))
