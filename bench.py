#!/usr/bin/env python3
"""Headline benchmark: MPPI planning steps/s on the Neural-Laplace-Control hot path (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]          # N = 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W    # N > 1, one rank per GPU (RCCL)

A "step" is one ``MPPIDelay.command()`` -- the region the reference times at mppi_with_model.py:257-259 --
on BASELINE configs[1]: oderl-cartpole (nx=5, nu=1), K=16384 samples, horizon T=40, action_buffer_size 4,
Neural-Laplace dynamics h=128 / S=17 / Fourier ILT, float64, seeded synthetic weights (no checkpoints ship).
Noise is drawn on the device (Philox) so every input of the timed region is HBM-resident.  With N > 1 the
SAME K=16384 population is sharded over the ranks (strong scaling, as the metric is worded) and each
command() does one RCCL all-gather of 2+T*nu doubles.

Prints ONE JSON line on rank 0 with the contract fields plus ``roofline`` (dominant kernel: the FP64-MFMA
GRU encoder), ``roofline_ilt`` (stand-alone Fourier ILT kernel, HBM-bound; ``dehoog33`` inside it is the de Hoog
kernel of the ILT ablation) and ``cpu_baseline`` (the CPU
oracle = reference op sequence, timed on this host's cores; rank 0, N=1 only).
"""

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ENV, K_SAMPLES, HORIZON, ABUF, S_TERMS, HIDDEN, A_HIGH = "oderl-cartpole", 16384, 40, 4, 17, 128, 3.0
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
# newest committed PMC summary (separate rocprofv3 --pmc passes, tools/pmc_summarize.py)
PMC_JSON = next((p for p in (os.path.join(REPO, "profiles", n) for n in ("r1k_pmc_kernels.json", "r1j_pmc_kernels.json", "r1h_pmc_kernels.json"))
                 if os.path.exists(p)), os.path.join(REPO, "profiles", "r1j_pmc_kernels.json"))
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix (= FP64 vector) dense peak, AMD datasheet; the guide lists no f64 row


def synthetic_state_dict(d, nu, S, seed=0):
    """Reference-constructor init (seed 0) + the 'trained-like' phi-bias shift (see DESIGN.md §synthetic weights).

    Product-side twin of oracle.nl_model.make_synthetic_state_dict(tame=True); tests check they agree.
    """
    import neurallaplacecontrol_amd as nlc

    rng = torch.random.get_rng_state()
    torch.manual_seed(seed)
    model = nlc.NeuralLaplaceModel(
        d, nu, d, hidden_units=HIDDEN, s_recon_terms=S, ilt_algorithm="fourier",
        state_mean=np.zeros(d), state_std=np.array([2.88646771, 11.54556671, 0.70729307, 0.70692035, 17.3199048][:d]),
        action_mean=np.array([0] * nu), action_std=np.array([A_HIGH / 2.0]), normalize=True, normalize_time=True,
    ).double()
    torch.random.set_rng_state(rng)
    with torch.no_grad():
        model.laplace_rep_func.linear_tanh_stack[4].bias[d * S :] += -3.0
    return model


def flops_gru_per_window(g, nin_pad, B):
    # MFMAs issued per 16 windows (kernels_gru.hip header) * 2048 flop / 16
    MT, KS = 3 * g // 16, g // 4
    mfma = B * MT + (B - 1) * KS * MT + B * KS * MT + (B - 1) * KS * MT + KS
    return mfma * 2048 / 16


def flops_rollout_per_sample_step(h, nt3):
    HT, KS = h // 16, h // 4
    mfma = (2 + KS) * HT + KS * nt3 + 2 * nt3
    return mfma * 2048 / 16


def cpu_baseline(sd, d, nu, budget_s=12.0):
    """Oracle (torch-CPU float64, aten::gru like the reference) timed on this host; bounded sample."""
    from oracle import envs as oenvs
    from oracle import mppi as omppi
    from oracle import nl_model as onl

    K, T = K_SAMPLES, HORIZON
    tg = onl.TorchGRUModel(sd, nu)
    ts = torch.full((K, 1), 0.05, dtype=torch.float64)

    def dynamics(state, window):
        return state + tg.forward(state, window, ts, S=S_TERMS).view(state.shape)

    sig = torch.ones((nu, nu), dtype=torch.double) * 0.5 + torch.eye(nu, dtype=torch.double) * 0.5
    torch.manual_seed(0)
    mppi = omppi.MPPIOracle(dynamics, oenvs.RUNNING_COST[ENV], d, sig, K, T, 1.0, torch.tensor(-A_HIGH),
                            torch.tensor(A_HIGH), A_HIGH)
    state, ab = oenvs.initial_state(ENV), torch.zeros(ABUF, nu, dtype=torch.float64)
    # torch's default (one thread per core) is far from the best setting for these small FP64 ops on a
    # many-core host (128 threads: ~42 s per command on the MI355X box); sweep a few thread counts within
    # the time budget and report the best, i.e. the strongest CPU baseline
    ncores = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    best, tried = None, []
    t_start = time.perf_counter()
    with torch.no_grad():
        for nt in [c for c in (16, 8, 32) if c <= ncores] or [ncores]:
            if tried and time.perf_counter() - t_start > budget_s:
                break
            torch.set_num_threads(nt)
            t0 = time.perf_counter()
            mppi.command(state, ab)
            el = time.perf_counter() - t0
            tried.append((nt, el))
            if best is None or el < best[1]:
                best = (nt, el)
    torch.set_num_threads(default_threads)
    return dict(value=1.0 / best[1], unit="planning steps/s", cores=best[0], kind="port",
                sample=f"one full command() of the same workload (K={K}, T={T}) per thread count "
                       f"{[(n, round(e, 2)) for n, e in tried]} (threads, seconds); best reported; host has {ncores} "
                       f"logical cores; torch {torch.__version__} CPU float64, aten::gru encoder as in the reference")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--samples", type=int, default=K_SAMPLES,
                    help="override K (experiments only; the headline metric is quoted at the default 16384)")
    args = ap.parse_args()
    # stdout carries exactly ONE JSON line (driver contract).  Libraries write there too -- RCCL prints its
    # NCCL_DEBUG=VERSION banner (set by the box image) to stdout at communicator creation -- so everything else this
    # process and its libraries print goes to stderr, and the result line is written to the saved stdout at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    import torch.distributed as dist

    import neurallaplacecontrol_amd as nlc

    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local)
    pg = None
    if world > 1 or "RANK" in os.environ:  # under torch.distributed.run even a 1-rank job goes through RCCL
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        pg = dist.group.WORLD
    K_total = args.samples

    d, nu = 5, 1
    model = synthetic_state_dict(d, nu, S_TERMS).to(f"cuda:{local}")
    sd_cpu = {k: v.detach().cpu().to(torch.float64) for k, v in model.state_dict().items()}
    planner = nlc.MPPIDelay(
        nlc.NLDynamics(model, 0.05), nlc.EnvCost(ENV), d, nlc.noise_sigma(nu), num_samples=K_total, horizon=HORIZON,
        device=f"cuda:{local}", lambda_=1.0, u_min=torch.tensor(-A_HIGH), u_max=torch.tensor(A_HIGH), u_scale=A_HIGH,
        noise_rng="philox", seed=0, process_group=pg, U_init=torch.zeros(HORIZON, nu, dtype=torch.float64),
    )
    state = nlc.initial_state(ENV, torch.Generator().manual_seed(0))
    abuf = torch.zeros(ABUF, nu, dtype=torch.float64)

    def step(ab):
        a = planner.command(state, ab)
        ab = torch.roll(ab, -1, dims=0)  # harness get_action (mppi_with_model.py:25-28)
        ab[-1] = a.cpu()
        return ab

    def fence():
        torch.cuda.synchronize()
        if pg is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        abuf = step(abuf)
    planner.ctx.profile_reset()
    planner.ctx.profile(True)  # hipEvent pairs around every launch, on the launch stream
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        abuf = step(abuf)
    fence()
    elapsed = time.perf_counter() - t0
    planner.ctx.profile(False)
    prof = planner.ctx.profile_read()
    if pg is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- stand-alone ILT kernel at N = K*T points (the BASELINE 'ILT GB/s vs HBM peak' figure)
    ilt = None
    if rank == 0:
        N = K_SAMPLES * HORIZON
        g = torch.Generator(device="cuda").manual_seed(1)
        theta = (torch.rand(N, d, S_TERMS, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * np.pi
        phi = (torch.rand(N, d, S_TERMS, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * (np.pi / 2)
        tt = torch.full((N,), 0.125, dtype=torch.float64, device="cuda")
        from neurallaplacecontrol_amd.laplace import default_ctx

        ictx = default_ctx(local)
        for _ in range(3):
            nlc.ilt_reconstruct(theta, phi, tt)
        ictx.profile_reset()
        ictx.profile(True)
        for _ in range(20):
            nlc.ilt_reconstruct(theta, phi, tt)
        torch.cuda.synchronize()
        ictx.profile(False)
        p = ictx.profile_read()["ilt_fourier_kernel"]
        ms = p["total_ms"] / p["launches"]
        nbytes = N * (2 * d * S_TERMS + d) * 8
        traffic, traffic_src = None, None
        pmc = PMC_JSON
        if os.path.exists(pmc):  # PMC passes cannot run inside this process: separate rocprofv3 --pmc runs
            pj = json.load(open(pmc))
            traffic = pj["ilt_fourier"]["hbm_bytes_per_launch"]
            traffic_src = f"profiles/{os.path.basename(pmc)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH x2 gfx950 correction, same N)"
        ilt = dict(bound="hbm", achieved=nbytes / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                   frac=nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, traffic=traffic, traffic_source=traffic_src,
                   algorithmic_bytes=nbytes, kernel="ilt_fourier_kernel",
                   avg_launch_ms=ms, points=N, bytes_per_point=(2 * d * S_TERMS + d) * 8)
        del theta, phi
        # the ablation's second kernel (BASELINE configs[4]): de Hoog with 33 terms at the same N.  FP64-VALU bound
        # (about 150 VALU instructions per 8-byte term), so its HBM fraction is a utilisation figure, not a target.
        S2 = 33
        theta = (torch.rand(N, d, S2, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * np.pi
        phi = (torch.rand(N, d, S2, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * (np.pi / 2) * 0.9
        for _ in range(2):
            nlc.ilt_reconstruct(theta, phi, tt, "dehoog")
        ictx.profile_reset()
        ictx.profile(True)
        for _ in range(10):
            nlc.ilt_reconstruct(theta, phi, tt, "dehoog")
        torch.cuda.synchronize()
        ictx.profile(False)
        p = ictx.profile_read()["ilt_dehoog_kernel"]
        ms2 = p["total_ms"] / p["launches"]
        nb2 = N * (2 * d * S2 + d) * 8
        ilt["dehoog33"] = dict(bound="fp64-valu", kernel="ilt_dehoog_kernel", avg_launch_ms=ms2, points=N,
                               algorithmic_bytes=nb2, achieved=nb2 / (ms2 * 1e-3) / 1e9, unit="GB/s",
                               frac_hbm=nb2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               traffic=pj.get("ilt_dehoog_final", pj.get("ilt_dehoog", {})).get("hbm_bytes_per_launch")
                               if os.path.exists(pmc) else None,
                               traffic_note=f"profiles/{os.path.basename(pmc)}")
        del theta, phi
        # backward of the Fourier ILT (training through laplace_reconstruct): reads theta, phi, writes both gradients
        theta = ((torch.rand(N, d, S_TERMS, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * np.pi).requires_grad_()
        phi = ((torch.rand(N, d, S_TERMS, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * (np.pi / 2) * 0.99).requires_grad_()
        gx = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
        for it in range(13):
            if it == 3:
                ictx.profile_reset()
                ictx.profile(True)
            torch.autograd.grad(nlc.ilt_reconstruct(theta, phi, tt), (theta, phi), gx)
        torch.cuda.synchronize()
        ictx.profile(False)
        p = ictx.profile_read()["ilt_fourier_bwd_kernel"]
        ms3 = p["total_ms"] / p["launches"]
        nb3 = N * 4 * d * S_TERMS * 8
        ilt["backward"] = dict(bound="hbm", kernel="ilt_fourier_bwd_kernel", avg_launch_ms=ms3, points=N,
                               algorithmic_bytes=nb3, bytes_per_point=4 * d * S_TERMS * 8,
                               achieved=nb3 / (ms3 * 1e-3) / 1e9, unit="GB/s", frac=nb3 / (ms3 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               traffic=pj.get("ilt_fourier_bwd", {}).get("hbm_bytes_per_launch") if os.path.exists(pmc) else None)
        del theta, phi, gx

    if rank != 0:
        if pg is not None:
            dist.destroy_process_group()
        return

    kernels = {k: dict(avg_ms=v["total_ms"] / max(v["launches"], 1), launches=v["launches"]) for k, v in prof.items()}
    k_local = K_total // world
    gru_flops = flops_gru_per_window(HIDDEN // 2, 4, ABUF) * k_local * HORIZON
    roll_flops = flops_rollout_per_sample_step(HIDDEN, 11) * k_local * HORIZON  # nt3 = 11 tiles for d=5, S=17
    gk = kernels.get("gru_encode_kernel", dict(avg_ms=float("nan")))
    rk = kernels.get("nl_rollout_kernel", dict(avg_ms=float("nan")))
    gru_tf = gru_flops / (gk["avg_ms"] * 1e-3) / 1e12
    g_traffic, r_traffic, t_src = None, None, None
    pmc = PMC_JSON
    if os.path.exists(pmc) and k_local == K_SAMPLES:  # measured at the headline size only
        pj = json.load(open(pmc))
        g_traffic, r_traffic = pj["gru_encode"]["hbm_bytes_per_launch"], pj["nl_rollout"]["hbm_bytes_per_launch"]
        t_src = f"profiles/{os.path.basename(pmc)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH x2 gfx950 correction)"
    roofline = dict(bound="mfma", achieved=gru_tf, peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=gru_tf / FP64_MFMA_PEAK_TFLOPS, traffic=g_traffic, traffic_source=t_src,
                    algorithmic_hbm_bytes=24 * k_local * HORIZON, kernel="gru_encode_kernel",
                    avg_launch_ms=gk["avg_ms"], flops_per_launch=gru_flops,
                    also=dict(kernel="nl_rollout_kernel", avg_launch_ms=rk["avg_ms"], flops_per_launch=roll_flops,
                              traffic=r_traffic,
                              achieved=roll_flops / (rk["avg_ms"] * 1e-3) / 1e12,
                              frac=roll_flops / (rk["avg_ms"] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS))
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(sd_cpu, d, nu, args.cpu_budget)
    info = planner.ctx.device_info()
    out = dict(
        metric="MPPI planning steps/sec (16384 samples, H=40)",
        value=args.steps / elapsed,
        unit="planning steps/s",
        n_gpus=world,
        steps=args.steps,
        warmup=args.warmup,
        ms_per_step=elapsed / args.steps * 1e3,
        higher_is_better=True,
        scaling="strong",
        vs_baseline=None,
        dtype="f64",
        data="synthetic",
        config=dict(workload="oderl-cartpole (nx=5, nu=1), K=16384 MPPI samples sharded over the ranks, H=40, "
                             "action_buffer_size=4, NL dynamics h=128 S=17 fourier ILT (BASELINE configs[1])",
                    samples_per_gpu=k_local, noise="device Philox4x32-10", device=info["name"]),
        roofline=roofline,
        roofline_ilt=ilt,
        cpu_baseline=cpu,
        kernels_avg_ms=kernels,
    )
    if cpu:
        out["speedup_vs_cpu_baseline"] = out["value"] / cpu["value"]
    if K_total != K_SAMPLES:
        out["config"]["workload"] += f" -- EXPERIMENT with K={K_total}, not the headline configuration"
    sys.stdout.flush()
    os.write(result_fd, (json.dumps(out) + "\n").encode())
    if pg is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
