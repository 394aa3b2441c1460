"""Cooperative GRU encode (one 16-window tile per workgroup) vs the wave-per-tile kernel: bit-exactness and time
(tools only): python tools/gru_coop_probe.py"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import neurallaplacecontrol_amd as nlc

d, nu = 5, 1
H = int(os.environ.get("PROBE_H", "128"))  # hidden_units: GRU width H / 2
if H == 128:
    model = bench.synthetic_state_dict(d, nu, bench.S_TERMS).to("cuda:0")
else:
    import numpy as np
    model = nlc.NeuralLaplaceModel(d, nu, d, hidden_units=H, s_recon_terms=17, ilt_algorithm="fourier", state_mean=np.zeros(d),
                                   state_std=np.ones(d), action_mean=np.array([0]), action_std=np.array([1.5]), normalize=True,
                                   normalize_time=True).double().to("cuda:0")
torch.manual_seed(0)
ctx = model.hip_ctx(torch.device("cuda:0"))
for N in [int(x) for x in os.environ.get("PROBE_N", "16,256,4096,40960,81920,163840,327680,655360").split(",")]:
    win = (torch.rand(N, 4, nu, dtype=torch.float64, device="cuda") * 2 - 1) * 3.0
    outs, times = [], []
    for coop in (0, 1):
        ctx.set_option("gru_coop", coop)
        with torch.no_grad():
            o = model.encode_actions(win)
            torch.cuda.synchronize()
            ctx.profile_reset(); ctx.profile(True)
            for _ in range(10):
                model.encode_actions(win)
            torch.cuda.synchronize(); ctx.profile(False)
        prof = ctx.profile_read()
        k = [v for n, v in prof.items() if "gru" in n][0]
        outs.append(o.clone()); times.append(k["total_ms"] / k["launches"])
    print(json.dumps(dict(H=H, N=N, equal=bool(torch.equal(outs[0], outs[1])), max_abs=float((outs[0] - outs[1]).abs().max()),
                          ms_wave_per_tile=round(times[0], 4), ms_coop=round(times[1], 4))), flush=True)
ctx.set_option("gru_coop", -1)
