"""Soak of the int8-sliced encoder (`gru_gemm = 1`): N commands of the headline planner (K = 16384, T = 40), a new state every
command; at every 97th state two fresh sliced planners must return the same bits (the kernel has no atomics and a fixed summation order:
run-to-run identical) and a fresh FP64-encoder planner the same action to 1e-9.

    python tools/i8_soak.py [commands]"""
import os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import neurallaplacecontrol_amd as nlc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
d, nu, A, T, K = 5, 1, 3.0, 40, 16384
model = bench.synthetic_state_dict(d, nu, 17).to("cuda")


def planner(opts):
    return nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(bench.ENV), d, nlc.noise_sigma(nu), K, T, "cpu", lambda_=1.0,
                         u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox", seed=0,
                         U_init=torch.zeros(T, nu, dtype=torch.float64), planner_options=opts)


g = torch.Generator().manual_seed(1)
p = planner({"gru_gemm": 1})
t0 = time.time()
checked = worst = 0
ab = torch.zeros(4, nu, dtype=torch.float64)
with torch.no_grad():
    for i in range(N):
        st = nlc.initial_state(bench.ENV, g)
        a_p = p.command(st, ab)
        assert torch.isfinite(a_p).all(), i
        if i % 97 == 0:
            # two fresh sliced planners and a fresh FP64 one on this state: the same first command
            q1, q2, r = planner({"gru_gemm": 1}), planner({"gru_gemm": 1}), planner({})
            a1, a2, a3 = q1.command(st, ab), q2.command(st, ab), r.command(st, ab)
            assert torch.equal(a1, a2), (i, a1, a2)
            worst = max(worst, float((a1 - a3).abs().max()))
            assert worst < 1e-9, (i, worst)
            checked += 1
            del q1, q2, r
        ab = torch.roll(ab, -1, 0)
        ab[-1] = a_p
print(f"{N} commands at K = {K} in {time.time() - t0:.0f} s, {checked} compared: run-to-run identical, sliced vs FP64 encoder action difference at most {worst:.2e}")
