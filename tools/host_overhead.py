"""Host-side overhead of one command(): tiny K so kernels are negligible; wall per call + cProfile top list."""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurallaplacecontrol_amd as nlc
import bench

model = bench.synthetic_state_dict(5, 1, 17).to("cuda")
for dev in ("cuda", "cpu"):
    for rng in ("philox", "torch"):
        m = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost("oderl-cartpole"), 5, nlc.noise_sigma(1), 256, 40, dev,
                          lambda_=1.0, u_min=torch.tensor(-3.0), u_max=torch.tensor(3.0), u_scale=3.0, noise_rng=rng,
                          U_init=torch.zeros(40, 1, dtype=torch.float64))
        st, ab = nlc.initial_state("oderl-cartpole"), torch.zeros(4, 1, dtype=torch.float64)
        for _ in range(20):
            a = m.command(st, ab)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 300
        for _ in range(n):
            a = m.command(st, ab)
            a_host = a.cpu()
        torch.cuda.synchronize()
        print(f"device={dev} rng={rng}: {(time.perf_counter() - t0) / n * 1e6:.1f} us per command (K=256, T=40)")
m = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost("oderl-cartpole"), 5, nlc.noise_sigma(1), 256, 40, "cuda",
                  lambda_=1.0, u_min=torch.tensor(-3.0), u_max=torch.tensor(3.0), u_scale=3.0, noise_rng="philox",
                  U_init=torch.zeros(40, 1, dtype=torch.float64))
for _ in range(20):
    m.command(st, ab)
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    m.command(st, ab).cpu()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
