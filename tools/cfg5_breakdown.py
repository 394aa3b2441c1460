"""Per-kernel time of one command() of BASELINE configs[4] (cartpole, K=16384, T=40, de Hoog S=33: the staged all-HIP path)."""
import sys, time, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurallaplacecontrol_amd as nlc
S=int(os.environ.get("CFG5_S", "33"))
ALGO=os.environ.get("CFG5_ALGO", "dehoog")  # tools: the same breakdown for a fixed_tablot / stehfest model (staged path)
torch.manual_seed(0)
model = nlc.NeuralLaplaceModel(5, 1, 5, hidden_units=128, s_recon_terms=S, ilt_algorithm=ALGO, state_mean=np.zeros(5),
    state_std=np.array([2.88646771, 11.54556671, 0.70729307, 0.70692035, 17.3199048]), action_mean=np.array([0]), action_std=np.array([1.5]), normalize=True, normalize_time=True).double()
with torch.no_grad(): model.laplace_rep_func.linear_tanh_stack[4].bias[5*S:] += -3.0
model = model.to("cuda")
mppi = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost("oderl-cartpole"), 5, nlc.noise_sigma(1), 16384, 40, "cuda", lambda_=1.0,
    u_min=torch.tensor(-3.0), u_max=torch.tensor(3.0), u_scale=3.0, noise_rng="philox", U_init=torch.zeros(40,1,dtype=torch.float64), store_rollouts=False,
    planner_options=dict(kv.split("=") for kv in os.environ.get("CFG5_OPTS", "").split(",") if kv))
st, ab = nlc.initial_state("oderl-cartpole"), torch.zeros(4,1,dtype=torch.float64)
with torch.no_grad():
    for _ in range(2): mppi.command(st, ab)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): mppi.command(st, ab).cpu()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
    mppi.ctx.profile_reset(); mppi.ctx.profile(True)
    for _ in range(5): mppi.command(st, ab).cpu()
    torch.cuda.synchronize(); mppi.ctx.profile(False)
p = mppi.ctx.profile_read()
tot=0
for k,v in p.items():
    per_cmd = v["total_ms"]/5
    tot+=per_cmd
    print(f"{k:26s} launches/cmd {v['launches']/5:5.0f}  ms/cmd {per_cmd:.3f}  avg us {v['total_ms']/v['launches']*1e3:.1f}")
print("sum kernels ms/cmd", round(tot,3), " wall ms/cmd (no profiling)", round(dt*1e3,3))
