"""The encoder's hidden-state GEMMs on the INT8 matrix pipe (`gru_gemm = 1`, csrc/kernels_gru_i8.hip) against the FP64-MFMA
encoder: latents on random action windows, then the headline planner
(BASELINE configs[1]: K = 16384, T = 40) with either encoder -- ms per command, per-kernel hipEvent averages, and how far the
actions of the first commands drift apart.

    python tools/i8_gemm_probe.py [--steps 40]"""
import argparse, json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import neurallaplacecontrol_amd as nlc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--windows", type=int, default=200000)
    ap.add_argument("--lib", default="", help="another build of the library (tools/_libnlc_<variant>.so)")
    a = ap.parse_args()
    if a.lib:
        from neurallaplacecontrol_amd import _lib
        _lib.use_library(os.path.abspath(a.lib))
    out = {}
    d, nu, A, T, K = 5, 1, 3.0, 40, 16384
    model = bench.synthetic_state_dict(d, nu, 17).to("cuda")
    g = torch.Generator().manual_seed(3)
    win = (torch.rand(a.windows, 4, nu, dtype=torch.float64, generator=g) * 2 - 1) * A
    ctx = model.hip_ctx(torch.device("cuda:0"))
    ctx.set_option("gru_coop", 0)
    with torch.no_grad():
        ctx.set_option("gru_gemm", 0)
        lat_f64 = model.encode_actions(win.cuda()).cpu()
        ctx.set_option("gru_gemm", 1)
        lat_i8 = model.encode_actions(win.cuda()).cpu()
        ctx.set_option("gru_gemm", 0)
    out["latents"] = dict(windows=a.windows, max_abs_i8_vs_f64=float((lat_i8 - lat_f64).abs().max()),
                          max_abs_latent=float(lat_f64.abs().max()))
    print(json.dumps(out), flush=True)

    def planner(opts):
        return nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(bench.ENV), d, nlc.noise_sigma(nu), K, T, "cpu", lambda_=1.0,
                             u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox", seed=0,
                             U_init=torch.zeros(T, nu, dtype=torch.float64), planner_options=opts)

    runs = {}
    for name, opts in (("f64", {}), ("i8", {"gru_gemm": 1})):
        p = planner(opts)
        st, ab = nlc.initial_state(bench.ENV, torch.Generator().manual_seed(0)), torch.zeros(4, nu, dtype=torch.float64)
        acts = []
        with torch.no_grad():
            for i in range(8):
                act = p.command(st, ab)
                acts.append(act.clone())
                ab = torch.roll(ab, -1, 0); ab[-1] = act
            t_end = time.perf_counter() + 0.5
            while time.perf_counter() < t_end:
                p.command(st, ab)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                act = p.command(st, ab)
                ab = torch.roll(ab, -1, 0); ab[-1] = act
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.steps * 1e3
            p.ctx.profile_reset(); p.ctx.profile(True)
            for _ in range(a.steps // 2):
                p.command(st, ab)
            torch.cuda.synchronize()
            p.ctx.profile(False)
            prof = {k: round(v["total_ms"] / max(v["launches"], 1), 4) for k, v in p.ctx.profile_read().items()}
        runs[name] = dict(ms_per_command=round(ms, 4), steps_per_s=round(1e3 / ms, 1), kernels_avg_ms=prof,
                          first_actions=[float(x[0]) for x in acts], cost_min=float(p.cost_total.min()))
        print(name, json.dumps(runs[name]), flush=True)
        del p
    out["planner"] = runs
    out["planner_action_drift_first_8_commands"] = max(abs(x - y) for x, y in zip(runs["f64"]["first_actions"], runs["i8"]["first_actions"]))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
