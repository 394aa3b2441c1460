"""Where one tile-step of nl_rollout_kernel<8, 11> spends its shader clocks: the headline planner (cartpole,
Fourier S = 17, K = 16384, H = 40) on a build of the library with s_memtime stamps at the phase boundaries of the model evaluation
(`make -C neurallaplacecontrol_amd/csrc variant` -> tools/_libnlc_phase.so).  Prints, per phase and tile-step, the measured clocks
beside the clocks its MFMAs alone hold the SIMD (64 each) -- the difference is VALU issue + stalls of that phase."""
import ctypes, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurallaplacecontrol_amd import _lib
LIB = os.environ.get("NLC_LIB_PATH", os.path.join(os.path.dirname(os.path.abspath(__file__)), "_libnlc_phase.so"))
if not os.path.exists(LIB):
    sys.exit(f"{LIB} is missing: build it with `make -C neurallaplacecontrol_amd/csrc variant` (a tools-only build of the library)")
_lib.use_library(LIB)
import neurallaplacecontrol_amd as nlc

env, d, nu, A, K, T, S = "oderl-cartpole", 5, 1, 3.0, int(os.environ.get("K", 16384)), 40, 17
torch.manual_seed(0)
model = nlc.NeuralLaplaceModel(d, nu, d, hidden_units=128, s_recon_terms=S, ilt_algorithm="fourier", state_mean=np.zeros(d),
                               state_std=np.array([2.88646771, 11.54556671, 0.70729307, 0.70692035, 17.3199048]),
                               action_mean=np.array([0]), action_std=np.array([A / 2.0]), normalize=True, normalize_time=True).double()
model = model.to("cuda")
p = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, nlc.noise_sigma(nu), K, T, "cpu", lambda_=1.0,
                  u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox", seed=0,
                  U_init=torch.zeros(T, nu, dtype=torch.float64), store_rollouts=False,
                  planner_options={"rollout_variant": 1})
st, ab = nlc.initial_state(env, torch.Generator().manual_seed(0)), torch.zeros(4, nu, dtype=torch.float64)
h = ctypes.CDLL(LIB)
h.nlc_debug_phase_clocks.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
out = (ctypes.c_ulonglong * 16)()
with torch.no_grad():
    for _ in range(5):
        p.command(st, ab)
    torch.cuda.synchronize()
    assert h.nlc_debug_phase_clocks(out) == 0  # clears the warm-up's sums
    n_cmd = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n_cmd):
        p.command(st, ab)
    e1.record()
    torch.cuda.synchronize()
    assert h.nlc_debug_phase_clocks(out) == 0
names = ["layer 1 (+ latent loads, normalisation)", "tanh, layer 1", "layer 2", "tanh, layer 2", "layer 3 first half",
         "epilogue first half", "layer 3 second half", "epilogue second half", "state update + costs (tail)", "other"]
# MFMAs of each phase per tile-step: HT = 8, NT3 = 11 (NA = 6, NB = 5), KS = 32; the epilogues' ILT MFMAs are 2 per tile
mfma = [16, 0, 256, 0, 192, 12, 160, 10, 0, 0]
waves = out[10]
steps = waves * T
rows, tot = [], 0.0
for i, nm in enumerate(names):
    c = out[i] / steps if steps else 0.0
    tot += c
    rows.append({"phase": nm, "clk": round(c, 1), "mfma_clk": mfma[i] * 64, "non_mfma_clk": round(c - mfma[i] * 64, 1)})
res = {"workload": f"cartpole fourier S={S} K={K} H={T}", "library": os.path.basename(LIB), "waves_sampled": int(waves),
       "commands": n_cmd, "ms_per_command_with_stamps": round(e0.elapsed_time(e1) / n_cmd, 4),
       "clk_per_tile_step": round(tot, 1), "mfma_clk_per_tile_step": sum(mfma) * 64, "phases": rows}
print(json.dumps(res, indent=1))
