"""A few commands of the configs[4] planner (cartpole, de Hoog S = 33, K = 16384, H = 40) on the persistent step-chain kernel
(`dehoog_chain` 1), for `rocprofv3 --pmc` passes (tools/collect_profiles.sh -> profiles/r4_pmc_cfg5_chain.json)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurallaplacecontrol_amd as nlc

env, d, nu, A, K, T, S = "oderl-cartpole", 5, 1, 3.0, 16384, 40, 33
torch.manual_seed(0)
model = nlc.NeuralLaplaceModel(d, nu, d, hidden_units=128, s_recon_terms=S, ilt_algorithm="dehoog", state_mean=np.zeros(d),
                               state_std=np.array([2.88646771, 11.54556671, 0.70729307, 0.70692035, 17.3199048]),
                               action_mean=np.array([0]), action_std=np.array([A / 2.0]), normalize=True, normalize_time=True).double()
with torch.no_grad():
    model.laplace_rep_func.linear_tanh_stack[4].bias[d * S:] += -3.0
model = model.to("cuda")
p = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, nlc.noise_sigma(nu), K, T, "cpu", lambda_=1.0,
                  u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox", seed=0,
                  U_init=torch.zeros(T, nu, dtype=torch.float64), store_rollouts=False, planner_options={"dehoog_chain": 1})
st, ab = nlc.initial_state(env, torch.Generator().manual_seed(0)), torch.zeros(4, nu, dtype=torch.float64)
with torch.no_grad():
    for _ in range(4):
        p.command(st, ab)
torch.cuda.synchronize()
