#!/usr/bin/env python3
"""Headline benchmark: MPPI planning steps/s on the Neural-Laplace-Control hot path (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]          # N > 1: starts its own ranks as a child process
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W    # N > 1, one rank per GPU (RCCL): the driver's form

A "step" is one ``MPPIDelay.command()`` -- the region the reference times at mppi_with_model.py:257-259 --
on BASELINE configs[1]: oderl-cartpole (nx=5, nu=1), K=16384 samples, horizon T=40, action_buffer_size 4,
Neural-Laplace dynamics h=128 / S=17 / Fourier ILT, float64, seeded synthetic weights (no checkpoints ship).
Noise is drawn on the device (Philox) so every input of the timed region is HBM-resident.  With N > 1 the
SAME K=16384 population is sharded over the ranks (strong scaling, as the metric is worded) and each
command() does one RCCL all-gather of 2+T*nu doubles.

The K timed steps run WITHOUT the library's per-launch event profiling; a second, untimed pass of the same steps
with profiling on gives the per-kernel averages the roofline uses (hipEvent pairs on the launch stream).

Prints ONE JSON line on rank 0 with the contract fields plus ``roofline`` (dominant kernel: the FP64-MFMA GRU
encoder at the headline size, the fused one-launch planner body on a small shard), ``roofline_ilt`` (stand-alone
Fourier ILT kernel, HBM-bound; ``dehoog33`` / ``backward`` inside it) and ``cpu_baseline`` (the CPU oracle =
reference op sequence, timed on this host's cores; rank 0, N=1 only).
"""

import argparse
import json
import os
import statistics
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ENV, K_SAMPLES, HORIZON, ABUF, S_TERMS, HIDDEN, A_HIGH = "oderl-cartpole", 16384, 40, 4, 17, 128, 3.0
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix (= FP64 vector) dense peak, AMD datasheet; the guide lists no f64 row
# newest committed PMC summary (separate rocprofv3 --pmc passes: tools/collect_profiles.sh + tools/pmc_summarize.py)
PMC_CANDIDATES = ("r4_pmc_kernels.json", "r3_pmc_kernels.json", "r2_pmc_kernels.json", "r1k_pmc_kernels.json")
# library kernel (nlc_profile_read name) -> key of the PMC summary; the summary's "_meta.kernel_names" must list a
# rocprof kernel name containing the library name, or the traffic figure belongs to some other build
PMC_KEYS = {"gru_encode_kernel": "gru_encode", "nl_rollout_kernel": "nl_rollout", "ilt_fourier_kernel": "ilt_fourier",
            "ilt_dehoog_kernel": "ilt_dehoog", "ilt_fourier_bwd_kernel": "ilt_fourier_bwd",
            "nl_plan_fused_kernel": "nl_plan_fused"}


def load_pmc():
    for name in PMC_CANDIDATES:
        p = os.path.join(REPO, "profiles", name)
        if os.path.exists(p):
            return name, json.load(open(p))
    return None, None


def pmc_traffic(pmc_name, pj, lib_kernel):
    """HBM bytes per launch of `lib_kernel` from the committed PMC summary, with its provenance; (None, None) if the
    summary has no such kernel.  A summary that names kernels this library does not have is a stale file: fail loudly."""
    if pj is None:
        return None, None
    key = PMC_KEYS[lib_kernel]
    if key not in pj:
        return None, None
    meta = pj.get("_meta")
    if meta is not None:
        names = meta.get("kernel_names", {}).get(key, [])
        if not any(lib_kernel in n for n in names):
            raise RuntimeError(f"profiles/{pmc_name}: entry {key!r} was collected from kernels {names}, none of which is "
                               f"the library's {lib_kernel!r} -- stale PMC summary, re-run tools/collect_profiles.sh")
        src = (f"profiles/{pmc_name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH x2 gfx950 "
               f"correction; collected at commit {meta.get('commit', '?')} on {meta.get('device', '?')}, {meta.get('date', '?')})")
    else:
        src = f"profiles/{pmc_name} (round-1 summary without provenance record; rocprofv3 --pmc, FETCH x2 gfx950 correction)"
    return pj[key].get("hbm_bytes_per_launch"), src


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


# ---- flop accounting.  "issued" = MFMAs the kernel executes x 2048 flop (padded tiles included);
# "needed" = the algorithm's multiply-adds (SURVEY §8d formulas, with the W_hh h0 = 0 products the kernel skips removed)
def flops_gru_issued_per_window(g, B):
    MT, KS = 3 * g // 16, g // 4
    mfma = B * MT + (B - 1) * KS * MT + B * KS * MT + (B - 1) * KS * MT + KS
    return mfma * 2048 / 16


def flops_gru_needed_per_window(g, nin, B):
    return B * 2 * 3 * g * nin + (B - 1) * 2 * 3 * g * g + B * 2 * 3 * g * g + (B - 1) * 2 * 3 * g * g + 2 * g * 2


def flops_rollout_issued_per_sample_step(h, nt3):
    HT, KS = h // 16, h // 4
    mfma = (2 + KS) * HT + KS * nt3 + 2 * nt3
    return mfma * 2048 / 16


def flops_rollout_needed_per_sample_step(h, d, S):
    P = d + 2  # the 2S constant sphere inputs are folded into the layer-1 bias
    return 2 * (P * h + h * h + 2 * d * S * h) + 2 * d * S


def host_cpu_info():
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = logical
    return dict(model=model, physical_cores=len(cores) or logical, logical_cpus=logical, usable_cpus=usable)


def cpu_baseline(sd, d, nu, budget_s=30.0):
    """Oracle (torch-CPU float64, aten::gru like the reference) timed on this host over the same region as the GPU
    step (mppi_with_model.py:257-259).  SURVEY §8d: warm-up, median of >= 3, 1 thread and all physical cores stated."""
    from oracle import envs as oenvs
    from oracle import mppi as omppi
    from oracle import nl_model as onl

    T = HORIZON
    tg = onl.TorchGRUModel(sd, nu)
    sig = torch.ones((nu, nu), dtype=torch.double) * 0.5 + torch.eye(nu, dtype=torch.double) * 0.5
    state, ab = oenvs.initial_state(ENV), torch.zeros(ABUF, nu, dtype=torch.float64)
    info = host_cpu_info()
    default_threads = torch.get_num_threads()

    def make(K):
        ts = torch.full((K, 1), 0.05, dtype=torch.float64)

        def dynamics(st, window):
            return st + tg.forward(st, window, ts, S=S_TERMS).view(st.shape)

        torch.manual_seed(0)
        return omppi.MPPIOracle(dynamics, oenvs.RUNNING_COST[ENV], d, sig, K, T, 1.0, torch.tensor(-A_HIGH),
                                torch.tensor(A_HIGH), A_HIGH)

    def timed(mppi, n):
        out = []
        for _ in range(n):
            t0 = time.perf_counter()
            mppi.command(state, ab)
            out.append(time.perf_counter() - t0)
        return out

    t_start = time.perf_counter()
    with torch.no_grad():
        # 1. thread-count sweep on a 1/8 population (warm-up + one command each): torch's default of one thread per
        #    logical CPU is far from the best setting for these small FP64 ops on a many-core host
        K8 = K_SAMPLES // 8
        small = make(K8)
        limit = info["usable_cpus"]
        cand = sorted({c for c in (8, 16, 32, info["physical_cores"]) if c <= limit} or {limit})
        sweep = []
        for nt in cand:
            torch.set_num_threads(nt)
            timed(small, 1)
            sweep.append((nt, timed(small, 1)[0]))
        best_nt = min(sweep, key=lambda x: x[1])[0]
        # 2. the reported figure: full population, best thread count, one warm-up, median of >= 3
        torch.set_num_threads(best_nt)
        full = make(K_SAMPLES)
        warm = timed(full, 1)[0]
        n_rep = 3
        if warm * 6 < budget_s - (time.perf_counter() - t_start):
            n_rep = 5
        reps = timed(full, n_rep)
        med = statistics.median(reps)
        # 3. single figures for 1 thread and for all physical cores, time-boxed on the 1/8 population and scaled by 8
        #    (the work is linear in K; at 1/8 the per-op overheads weigh more, so the extrapolation favours neither)
        torch.set_num_threads(1)
        timed(small, 1)
        one = timed(small, 1)[0] * 8.0
        allc = dict(sweep).get(info["physical_cores"])
        allc = allc * 8.0 if allc is not None else None
    torch.set_num_threads(default_threads)
    return dict(
        value=1.0 / med, unit="planning steps/s", cores=best_nt, kind="port",
        sample=(f"full workload (K={K_SAMPLES}, T={T}): 1 warm-up + median of {n_rep} command() calls at {best_nt} threads "
                f"(runs {[round(r, 2) for r in reps]} s, warm-up {warm:.2f} s); thread count picked by a sweep on K={K8}: "
                f"{[(n, round(e, 3)) for n, e in sweep]} (threads, s)"),
        cpu_model=info["model"], physical_cores=info["physical_cores"], logical_cpus=info["logical_cpus"],
        usable_cpus=info["usable_cpus"],
        # NOT measurements at the quoted size: one command at K / 8, scaled by 8 (context for the measured figure above only)
        one_thread_extrapolated_x8=dict(value_extrapolated=1.0 / one, seconds_per_command_extrapolated=one, measured_at_K=K8,
                                        note=f"EXTRAPOLATED: one command at K={K8}, time x 8"),
        all_physical_cores_extrapolated_x8=(dict(value_extrapolated=1.0 / allc, seconds_per_command_extrapolated=allc,
                                                 threads=info["physical_cores"], measured_at_K=K8,
                                                 note=f"EXTRAPOLATED: one command at K={K8}, time x 8") if allc else None),
        torch=torch.__version__, oracle="oracle/ (torch-CPU float64, aten::gru encoder as in the reference)",
    )


def free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n, argv, result_fd):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    as a child process (one rank per GPU), pass its stderr through, write its LAST stdout line -- rank 0's JSON line --
    to the saved stdout, and return its exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL between the ranks of one node)
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [ln for ln in proc.stdout.decode(errors="replace").splitlines() if ln.strip().startswith("{")]
    if lines:
        os.write(result_fd, (lines[-1] + "\n").encode())
    elif proc.returncode == 0:
        sys.stderr.write("bench.py: the launched ranks printed no JSON line\n")
        return 1
    return proc.returncode


def preheat(one_step, ms, group=None, device="cpu", chunk=16):
    """Untimed commands for about `ms` milliseconds, the SAME number on every rank: a sharded command() contains a
    collective, so ranks that stopped on their own clocks after different counts would leave unmatched collectives behind
    (and merge partials of different commands meanwhile).  Every `chunk` steps the ranks agree (MAX) on whether anyone
    still wants more.  Returns the number of steps taken."""
    import torch.distributed as dist

    n = 0
    if ms <= 0:
        return n
    t0 = time.perf_counter()
    while True:
        for _ in range(chunk):
            one_step()
        n += chunk
        more = (time.perf_counter() - t0) * 1e3 < ms
        if group is not None:
            flag = torch.tensor([1.0 if more else 0.0], dtype=torch.float64, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
            more = bool(flag.item() > 0.0)
        if not more:
            return n


def git_commit():
    try:
        return subprocess.check_output(["git", "-C", REPO, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        return os.environ.get("NLC_COMMIT")  # the GPU box gets a snapshot without .git


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ilt", action="store_true", help="skip the stand-alone ILT kernel section (experiments)")
    ap.add_argument("--cpu-budget", type=float, default=30.0)
    ap.add_argument("--collective", choices=("auto", "torch", "native"), default="auto",
                    help="N > 1: the per-command all-gather inside nlc_mppi_finish on the library's own RCCL communicator "
                         "(include/nlc.h, nlc_comm_init) or through torch.distributed between the two phases; auto (default) "
                         "= the library's, falling back to torch's if any rank cannot bring the communicator up")
    ap.add_argument("--planner-opt", action="append", default=[], metavar="NAME=VALUE",
                    help="extra planner_options entries (experiments), e.g. --planner-opt host_spin=0")
    ap.add_argument("--preheat-ms", type=float, default=300.0,
                    help="untimed commands for this long during set-up (clock ramp from idle), before the W warm-up steps")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launch rehearsal: start the ranks, report the environment each one sees, touch no GPU")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="launch rehearsal WITH the GPU work (tests): every rank plans its shard on cuda:0 and the ranks talk "
                         "over gloo (RCCL refuses two ranks per device) -- the whole N > 1 flow of this file on a 1-GPU box; "
                         "its numbers mean nothing")
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
    if args.gpus > 1 and "RANK" not in os.environ:
        # invoked bare (`python bench.py --gpus N`): start the ranks ourselves, as a CHILD process (nothing in this
        # process has touched the GPU yet), relay its one JSON line and exit with its code
        sys.exit(self_launch(args.gpus, sys.argv[1:], result_fd))
    if world != args.gpus:
        args.gpus = world
    import torch.distributed as dist

    if args.dry_launch:
        # launch rehearsal (CPU test): every rank reports the environment torch.distributed.run gave it over a gloo
        # group; rank 0 prints them as the one JSON line.  No GPU call is made.
        envs = [None] * world
        if "RANK" in os.environ:
            dist.init_process_group("gloo")
            dist.all_gather_object(envs, {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR")})
            dist.destroy_process_group()
        else:
            envs = [{"RANK": None, "LOCAL_RANK": None, "WORLD_SIZE": None, "MASTER_ADDR": None}]
        if rank == 0:
            os.write(result_fd, (json.dumps(dict(dry_launch=True, n_gpus=world, ranks=envs)) + "\n").encode())
        return

    import neurallaplacecontrol_amd as nlc

    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    if args.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    coll_dev = "cpu" if args.rehearse_on_one_gpu else f"cuda:{local}"  # where the bench's own small collectives live
    pg = None
    if world > 1 or "RANK" in os.environ:  # under torch.distributed.run even a 1-rank job goes through RCCL
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        pg = dist.group.WORLD
    K_total = args.samples

    d, nu = 5, 1
    model = synthetic_state_dict(d, nu, S_TERMS).to(f"cuda:{local}")
    sd_cpu = {k: v.detach().cpu().to(torch.float64) for k, v in model.state_dict().items()}
    planner = nlc.MPPIDelay(
        nlc.NLDynamics(model, 0.05), nlc.EnvCost(ENV), d, nlc.noise_sigma(nu), num_samples=K_total, horizon=HORIZON,
        # device="cpu": U and the returned action live on the host, as the harness's env.step needs them (the merge kernel
        # stores the action straight into pinned host memory); every kernel runs on compute_device
        device="cpu", compute_device=f"cuda:{local}", lambda_=1.0, u_min=torch.tensor(-A_HIGH), u_max=torch.tensor(A_HIGH), u_scale=A_HIGH,
        noise_rng="philox", seed=0, process_group=pg, U_init=torch.zeros(HORIZON, nu, dtype=torch.float64),
        planner_options=dict({} if args.collective == "auto" else {"native_collective": int(args.collective == "native")},
                             **{kv.split("=", 1)[0]: float(kv.split("=", 1)[1]) for kv in args.planner_opt}),
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

    # set-up, before the W warm-up steps and outside every timed region: planner construction above, and `--preheat-ms` of
    # untimed commands so that a short run (the driver's K = 20) does not time the GPU's clock ramp from idle (at one
    # 8-GPU shard, 0.7 ms per step, 50 cold steps measured 0.74 ms per step where a 200-step loop measures 0.71)
    holder = [abuf]

    def one_step():
        holder[0] = step(holder[0])

    preheat(one_step, args.preheat_ms, pg, coll_dev)
    abuf = holder[0]
    for _ in range(args.warmup):
        abuf = step(abuf)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        abuf = step(abuf)
    fence()
    elapsed = time.perf_counter() - t0
    if pg is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # second, untimed pass: the same steps with hipEvent pairs around every launch (on the launch stream)
    planner.ctx.profile_reset()
    planner.ctx.profile(True)
    for _ in range(args.steps):
        abuf = step(abuf)
    fence()
    planner.ctx.profile(False)
    prof = planner.ctx.profile_read()

    pmc_name, pj = load_pmc()

    # ---- stand-alone ILT kernel at N = K*T points (the BASELINE 'ILT GB/s vs HBM peak' figure)
    ilt = None
    if rank == 0 and not args.no_ilt:
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
        traffic, traffic_src = pmc_traffic(pmc_name, pj, "ilt_fourier_kernel")
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
        tr2, tr2_src = pmc_traffic(pmc_name, pj, "ilt_dehoog_kernel")
        ilt["dehoog33"] = dict(bound="fp64-valu", kernel="ilt_dehoog_kernel", avg_launch_ms=ms2, points=N,
                               algorithmic_bytes=nb2, achieved=nb2 / (ms2 * 1e-3) / 1e9, unit="GB/s",
                               frac_hbm=nb2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, traffic=tr2, traffic_source=tr2_src)
        del theta, phi
        # fixed Talbot at the Fourier kernel's shape: the same coalesced stream with the algorithm's per-term phase / weight
        theta = (torch.rand(N, d, S_TERMS, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * np.pi
        phi = (torch.rand(N, d, S_TERMS, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * (np.pi / 2) * 0.9
        for _ in range(2):
            nlc.ilt_reconstruct(theta, phi, tt, "fixed_tablot")
        ictx.profile_reset()
        ictx.profile(True)
        for _ in range(10):
            nlc.ilt_reconstruct(theta, phi, tt, "fixed_tablot")
        torch.cuda.synchronize()
        ictx.profile(False)
        p = ictx.profile_read()["ilt_linear_stream_kernel"]
        ms4 = p["total_ms"] / p["launches"]
        ilt["fixed_tablot17"] = dict(bound="hbm", kernel="ilt_fourier_kernel<.., LIN> (ilt_linear_stream_kernel)", avg_launch_ms=ms4,
                                     points=N, algorithmic_bytes=nbytes, achieved=nbytes / (ms4 * 1e-3) / 1e9, unit="GB/s",
                                     frac=nbytes / (ms4 * 1e-3) / 1e9 / HBM_PEAK_GBS, traffic=None)
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
        tr3, tr3_src = pmc_traffic(pmc_name, pj, "ilt_fourier_bwd_kernel")
        ilt["backward"] = dict(bound="hbm", kernel="ilt_fourier_bwd_kernel", avg_launch_ms=ms3, points=N,
                               algorithmic_bytes=nb3, bytes_per_point=4 * d * S_TERMS * 8,
                               achieved=nb3 / (ms3 * 1e-3) / 1e9, unit="GB/s", frac=nb3 / (ms3 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               traffic=tr3, traffic_source=tr3_src)
        del theta, phi, gx

    def teardown():
        # orderly end: rank 0 arrives late (stand-alone ILT section); nobody tears a communicator down under a peer
        if pg is not None:
            fence()
            if planner.native_collective:
                planner.ctx.comm_destroy()
            dist.destroy_process_group()

    if rank != 0:
        teardown()
        return

    kernels = {k: dict(avg_ms=v["total_ms"] / max(v["launches"], 1), launches=v["launches"]) for k, v in prof.items()}
    k_local = K_total // world
    windows = k_local * HORIZON
    g_hidden = HIDDEN // 2
    gru_need = flops_gru_needed_per_window(g_hidden, nu, ABUF) * windows
    gru_iss = flops_gru_issued_per_window(g_hidden, ABUF) * windows
    roll_need = flops_rollout_needed_per_sample_step(HIDDEN, d, S_TERMS) * windows
    roll_iss = flops_rollout_issued_per_sample_step(HIDDEN, 11) * windows  # nt3 = 11 layer-3 tiles for d=5, S=17

    def mfma_entry(kernel, need, issued, alg_bytes):
        k = kernels[kernel]
        sec = k["avg_ms"] * 1e-3
        traffic, src = (pmc_traffic(pmc_name, pj, kernel) if k_local == K_SAMPLES else (None, None))
        return dict(bound="mfma", achieved=need / sec / 1e12, peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=need / sec / 1e12 / FP64_MFMA_PEAK_TFLOPS, traffic=traffic, traffic_source=src,
                    kernel=kernel, avg_launch_ms=k["avg_ms"], flops_per_launch=need,
                    issued_flops_per_launch=issued, frac_issued=issued / sec / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                    algorithmic_hbm_bytes=alg_bytes,
                    note="achieved/frac count the algorithm's flops (SURVEY 8d, W_hh h0 = 0 products skipped); "
                         "frac_issued counts the MFMAs the kernel executes (16-wide tile padding included)")

    if "nl_plan_fused_kernel" in kernels:
        # small shard: GRU encode and split rollout are roles of ONE launch (kernels_fused.hip)
        roofline = mfma_entry("nl_plan_fused_kernel", gru_need + roll_need, gru_iss + roll_iss,
                              (8 * nu + 16 + 16 + 8 * (2 * nu + d)) * windows)
    else:
        roofline = mfma_entry("gru_encode_kernel", gru_need, gru_iss, (8 * nu + 16) * windows)
        roofline["also"] = mfma_entry("nl_rollout_kernel", roll_need, roll_iss, (16 + 8 * (2 * nu + d)) * windows)
    # whole step against the same roof: every algorithmic flop of one command() (GRU encode + rollout, all ranks) over the
    # measured wall time per step -- the sampling / weighting / merge kernels add time but no matrix flops
    step_flops = (gru_need + roll_need) * world
    roofline["step"] = dict(flops=step_flops, achieved=step_flops / (elapsed / args.steps) / 1e12, peak=FP64_MFMA_PEAK_TFLOPS * world,
                            unit="TFLOP/s", frac=step_flops / (elapsed / args.steps) / 1e12 / (FP64_MFMA_PEAK_TFLOPS * world),
                            note="algorithmic flops of one command() (GRU encode + rollout; SURVEY 8d) / ms_per_step / FP64-MFMA peak")
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(sd_cpu, d, nu, args.cpu_budget)
    info = planner.ctx.device_info()
    workload = (f"oderl-cartpole (nx=5, nu=1), K={K_total} MPPI samples sharded over the ranks, H={HORIZON}, "
                f"action_buffer_size={ABUF}, NL dynamics h={HIDDEN} S={S_TERMS} fourier ILT")
    workload += " (BASELINE configs[1])" if K_total == K_SAMPLES else " -- EXPERIMENT: not the headline population of 16384"
    if args.rehearse_on_one_gpu:
        workload += " -- REHEARSAL: every rank on cuda:0 over gloo, not a measurement"
    out = dict(
        metric=f"MPPI planning steps/sec ({K_total} samples, H={HORIZON})",
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
        config=dict(workload=workload, samples_per_gpu=k_local, noise="device Philox4x32-10", device=info["name"],
                    commit=git_commit(), preheat_ms=args.preheat_ms,
                    collective=None if pg is None else ("rccl all-gather inside nlc_mppi_finish (library communicator)"
                                                        if planner.native_collective else
                                                        f"{dist.get_backend(pg)} all-gather via torch.distributed between the two phases")),
        roofline=roofline,
        roofline_ilt=ilt,
        cpu_baseline=cpu,
        kernels_avg_ms=kernels,
        kernels_note="per-launch hipEvent averages from a second, untimed pass of the same steps",
    )
    if cpu:
        out["speedup_vs_cpu_baseline"] = out["value"] / cpu["value"]
    sys.stdout.flush()
    os.write(result_fd, (json.dumps(out) + "\n").encode())
    teardown()


if __name__ == "__main__":
    main()
