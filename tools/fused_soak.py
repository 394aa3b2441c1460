"""Soak test of the fused planner body (tools only): many consecutive commands with changing states and action buffers;
every CHECK-th command is compared bit for bit with a two-launch planner fed the same U, state and counters.
    python tools/fused_soak.py [K] [commands]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import neurallaplacecontrol_amd as nlc

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
CHECK = 97
d, nu, T = 5, 1, bench.HORIZON
model = bench.synthetic_state_dict(d, nu, bench.S_TERMS).to("cuda:0")


def planner(variant):
    return nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(bench.ENV), d, nlc.noise_sigma(nu), num_samples=K, horizon=T,
                         device="cuda:0", lambda_=1.0, u_min=torch.tensor(-3.0), u_max=torch.tensor(3.0), u_scale=3.0,
                         noise_rng="philox", seed=11, U_init=torch.zeros(T, nu, dtype=torch.float64),
                         planner_options={"rollout_variant": variant})


fused, ref = planner(3), planner(2)
g = torch.Generator().manual_seed(0)
ab = torch.zeros(4, nu, dtype=torch.float64)
t0 = time.time()
checked = 0
for i in range(N):
    state = nlc.initial_state(bench.ENV, g)
    if i % CHECK == 0:
        ref.U = fused.U
        ref._commands = fused._commands
        a_ref = ref.command(state, ab)
        a = fused.command(state, ab)
        assert torch.equal(a, a_ref) and torch.equal(fused.cost_total, ref.cost_total) and torch.equal(fused.states, ref.states), i
        checked += 1
    else:
        a = fused.command(state, ab)
    ab = torch.roll(ab, -1, 0)
    ab[-1] = a.cpu()
    if i % 5000 == 4999:
        print(f"K={K}: {i + 1} commands, {checked} checked, {time.time() - t0:.1f} s", flush=True)
print(f"K={K}: {N} commands ok ({checked} compared bit for bit with the two-launch path), {time.time() - t0:.1f} s")
