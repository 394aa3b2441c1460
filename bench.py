#!/usr/bin/env python3
"""Headline benchmark: MPPI planning steps/s on the Neural-Laplace-Control hot path (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {0,1,2,3,4,d4}]   # N > 1: starts its own ranks as a child process
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W    # N > 1, one rank per GPU (RCCL): the driver's form

A "step" is one ``MPPIDelay.command()`` -- the region the reference times at mppi_with_model.py:257-259 -- by default
on BASELINE configs[1]: oderl-cartpole (nx=5, nu=1), K=16384 samples, horizon T=40, action_buffer_size 4,
Neural-Laplace dynamics h=128 / S=17 / Fourier ILT, float64, seeded synthetic weights (no checkpoints ship).
``--config`` selects another BASELINE config (0, 2, 3, 4) or north_star's literal state_dim=4 shape (d4) at its full
population; the default line is unchanged.  Noise is drawn on the device (Philox) so every input of the timed region is
HBM-resident.  With N > 1 the SAME population is sharded over the ranks (strong scaling, as the metric is worded) and
each command() does one RCCL all-gather of 2+T*nu doubles.

N > 1 cannot hang the run that measures it.  Under a launcher every rank process is a SUPERVISOR that
touches no GPU: it runs the measurement in a fresh child process under a progress watchdog, the supervisors agree on the
outcome through the launcher's TCP store, and
  * a child that fails or stops making progress with ``--collective auto`` (the library-owned RCCL communicator inside
    nlc_mppi_finish) is killed on every rank and a second fresh child measures with ``--collective torch``; the line then
    carries ``config.collective_fallback_reason``;
  * when the first child succeeds with the library's collective, a second child measures torch.distributed's all-gather
    too and the line carries it under ``also_collective`` -- one record, both modes;
  * ``config.ranks_seen`` holds the group size torch.distributed and the library's communicator report, and per rank the
    device, the rollout body, the kernel averages (``rccl_all_gather`` among them) and the collective's own timing.

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
import tempfile
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ENV, K_SAMPLES, HORIZON, ABUF, S_TERMS, HIDDEN, A_HIGH = "oderl-cartpole", 16384, 40, 4, 17, 128, 3.0
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix (= FP64 vector) dense peak, AMD datasheet; the guide lists no f64 row
# newest committed PMC summary (separate rocprofv3 --pmc passes: tools/collect_profiles.sh + tools/pmc_summarize.py)
PMC_CANDIDATES = ("r6_pmc_kernels.json", "r5_pmc_kernels.json", "r4_pmc_kernels.json", "r3_pmc_kernels.json", "r2_pmc_kernels.json", "r1k_pmc_kernels.json")
# library kernel (nlc_profile_read name) -> key of the PMC summary; the summary's "_meta.kernel_names" must list a
# rocprof kernel name containing the library name, or the traffic figure belongs to some other build
PMC_KEYS = {"gru_encode_kernel": "gru_encode", "nl_rollout_kernel": "nl_rollout", "ilt_fourier_kernel": "ilt_fourier",
            "ilt_dehoog_kernel": "ilt_dehoog", "ilt_fourier_bwd_kernel": "ilt_fourier_bwd",
            "nl_plan_fused_kernel": "nl_plan_fused"}

# env -> (nx, nu, action bound A, state_std of train_utils.py:187-200).  "oderl-cartpole-notrig": CTCartpole(obs_trans=False),
# the reference's 4-dim cartpole state (ctcartpole.py:60) = north_star's literal "state_dim=4".
ENV_SHAPES = {
    "oderl-cartpole": (5, 1, 3.0, [2.88646771, 11.54556671, 0.70729307, 0.70692035, 17.3199048]),
    "oderl-pendulum": (3, 1, 2.0, [0.70634571, 0.70784512, 2.89072771]),
    "oderl-acrobot": (6, 2, 5.0, [0.70711024, 0.70710328, 0.7072186, 0.7069949, 2.88642115, 2.88627309]),
    "oderl-cartpole-notrig": (4, 1, 3.0, [2.88646771, 11.54556671, 1.81379936, 17.3199048]),
}
# BASELINE.json configs[i] (+ "d4").  B = action_buffer rows: 4 (config.py:58), 5 for delay 4.  `gpus` = the
# GPU count the config is worded for; `--config i --gpus 1` runs its WHOLE population on one GPU.
CONFIGS = {
    "0": dict(name="BASELINE configs[0]", env="oderl-cartpole", B=4, K=1024, T=20, algo="fourier", S=17, gpus=1, delay=0),
    "1": dict(name="BASELINE configs[1]", env="oderl-cartpole", B=4, K=16384, T=40, algo="fourier", S=17, gpus=1, delay=2),
    "2": dict(name="BASELINE configs[2]", env="oderl-pendulum", B=5, K=65536, T=40, algo="fourier", S=17, gpus=2, delay=4),
    "3": dict(name="BASELINE configs[3]", env="oderl-acrobot", B=4, K=262144, T=60, algo="fourier", S=17, gpus=8, delay=2),
    "4": dict(name="BASELINE configs[4]", env="oderl-cartpole", B=4, K=16384, T=40, algo="dehoog", S=33, gpus=1, delay=2),
    "d4": dict(name="north_star literal state_dim=4 (CTCartpole(obs_trans=False))", env="oderl-cartpole-notrig", B=4, K=16384,
               T=40, algo="fourier", S=17, gpus=1, delay=2),
}


def load_pmc():
    for name in PMC_CANDIDATES:
        p = os.path.join(REPO, "profiles", name)
        if os.path.exists(p):
            return name, json.load(open(p))
    return None, None


def build_info():
    """What __graft_entry__.build() recorded next to the library it built (neurallaplacecontrol_amd/_build_info.py):
    the commit of the tree and the content hash of the library's sources.  {} when the library was built some other way."""
    try:
        from neurallaplacecontrol_amd import _build_info as bi

        return dict(commit=bi.COMMIT, csrc_commit=bi.CSRC_COMMIT, csrc_sha=bi.CSRC_SHA, dirty=bi.DIRTY, built=bi.BUILT)
    except Exception:
        return {}


STRICT_PMC = False  # --strict-pmc: a stale PMC summary is an error instead of `traffic: null` with the reason


def pmc_traffic(pmc_name, pj, lib_kernel):
    """HBM bytes per launch of `lib_kernel` from the committed PMC summary, with its provenance; (None, None) if the
    summary has no such kernel.  A summary that names kernels this library does not have is a stale file: fail loudly.
    A summary collected from OTHER SOURCES than the library in use was built from (`_meta.csrc_sha` vs the hash
    __graft_entry__.build() recorded) is not quoted: traffic is null and the reason takes its place
    (an error under --strict-pmc)."""
    if pj is None:
        return None, None
    key = PMC_KEYS[lib_kernel]
    if key not in pj:
        return None, None
    meta = pj.get("_meta")
    if meta is not None:
        names = meta.get("kernel_names", {}).get(key, [])
        # (the stand-alone Fourier ILT launches run the row-per-lane kernels since round 6; the library still times them under
        # the launcher's name)
        accepted = {"ilt_fourier_kernel": ("ilt_fourier_kernel", "ilt_fourier_rows_kernel"),
                    "ilt_fourier_bwd_kernel": ("ilt_fourier_bwd_kernel", "ilt_fourier_bwd_rows_kernel")}.get(lib_kernel, (lib_kernel,))
        if not any(a in n for a in accepted for n in names):
            raise RuntimeError(f"profiles/{pmc_name}: entry {key!r} was collected from kernels {names}, none of which is "
                               f"the library's {lib_kernel!r} -- stale PMC summary, re-run tools/collect_profiles.sh")
        bi = build_info()
        have, want = meta.get("csrc_sha"), bi.get("csrc_sha")
        if want and have != want:
            why = (f"STALE: profiles/{pmc_name} was collected at commit {meta.get('commit', '?')} from kernel sources "
                   f"{have or '(hash not recorded)'}; this library was built from {want} (commit {bi.get('commit')}) -- "
                   "re-run tools/collect_profiles.sh")
            if STRICT_PMC:
                raise RuntimeError(why)
            sys.stderr.write("bench.py: " + why + "\n")
            return None, why
        src = (f"profiles/{pmc_name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH x2 gfx950 "
               f"correction; collected at commit {meta.get('commit', '?')} on {meta.get('device', '?')}, {meta.get('date', '?')}; "
               f"kernel sources {have or 'unrecorded'}{' = this build' if want else ''})")
    else:
        src = f"profiles/{pmc_name} (round-1 summary without provenance record; rocprofv3 --pmc, FETCH x2 gfx950 correction)"
    return pj[key].get("hbm_bytes_per_launch"), src


def tame_dehoog_(model, d, S, t_norm=0.125, alpha=1e-10, tol=1e-9, scale=2.0, w3_scale=0.02):
    """Product-side twin of oracle.nl_model.tame_dehoog_ (tests check they agree): the last layer's biases put F on a real
    Laplace transform a_c / (s + b_c) sampled on the model's own contour, its weights are scaled by w3_scale."""
    import math

    Tc = scale * t_norm
    gamma = alpha - math.log(tol) / (scale * Tc)
    k = torch.arange(S, dtype=torch.float64)
    s_k = torch.complex(torch.full((S,), gamma, dtype=torch.float64), math.pi * k / Tc)
    last = model.laplace_rep_func.linear_tanh_stack[4]
    with torch.no_grad():
        last.weight *= w3_scale
        for c in range(d):
            a_c, b_c = 0.03 * (c + 1) * (-1.0) ** c, 1.0 + 0.5 * c
            F = a_c / (s_k + b_c)
            theta = torch.atan2(F.imag, F.real)
            r2 = F.real**2 + F.imag**2
            phi = torch.asin((r2 - 1.0) / (r2 + 1.0))
            last.bias[c * S : (c + 1) * S] = torch.atanh(torch.clamp(theta / math.pi, -1 + 1e-12, 1 - 1e-12))
            last.bias[(d + c) * S : (d + c + 1) * S] = torch.atanh(torch.clamp(phi / (math.pi / 2), -1 + 1e-12, 1 - 1e-12))
    return model


def synthetic_state_dict(d, nu, S, seed=0, env=ENV, algo="fourier"):
    """Reference-constructor init (seed 0) + the 'trained-like' taming (see DESIGN.md §synthetic weights): the phi-bias
    shift for Fourier models, the Laplace-transform biases for de Hoog models.

    Product-side twin of oracle.nl_model.make_synthetic_state_dict(tame=True / "dehoog"); tests check they agree.
    """
    import neurallaplacecontrol_amd as nlc

    _, _, A, std = ENV_SHAPES[env]
    rng = torch.random.get_rng_state()
    torch.manual_seed(seed)
    model = nlc.NeuralLaplaceModel(
        d, nu, d, hidden_units=HIDDEN, s_recon_terms=S, ilt_algorithm=algo,
        state_mean=np.zeros(d), state_std=np.array(std[:d]),
        action_mean=np.array([0] * nu), action_std=np.array([A / 2.0]), normalize=True, normalize_time=True,
    ).double()
    torch.random.set_rng_state(rng)
    if algo == "dehoog":
        return tame_dehoog_(model, d, S)
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


def cpu_baseline(sd, cfg, budget_s=30.0):
    """Oracle (torch-CPU float64, aten::gru like the reference) timed on this host over the same region as the GPU
    step (mppi_with_model.py:257-259).  SURVEY §8d: warm-up, median of >= 3, 1 thread and all physical cores stated.
    The headline config is measured at its full population; a config whose full-size command would not fit the budget
    (configs[2], [3]) is measured at K / 8 and says so (`value` is then absent: `value_extrapolated` carries the x 8 figure)."""
    from oracle import envs as oenvs
    from oracle import mppi as omppi
    from oracle import nl_model as onl

    env, T, B, K_full, algo, S = cfg["env"], cfg["T"], cfg["B"], cfg["K"], cfg["algo"], cfg["S"]
    d, nu, A, _ = ENV_SHAPES[env]
    tg = onl.TorchGRUModel(sd, nu)
    sig = torch.ones((nu, nu), dtype=torch.double) * 0.5 + torch.eye(nu, dtype=torch.double) * 0.5
    state, ab = oenvs.initial_state(env), torch.zeros(B, nu, dtype=torch.float64)
    info = host_cpu_info()
    default_threads = torch.get_num_threads()

    def make(K):
        ts = torch.full((K, 1), 0.05, dtype=torch.float64)

        def dynamics(st, window):
            return st + tg.forward(st, window, ts, S=S, ilt_algorithm=algo).view(st.shape)

        torch.manual_seed(0)
        return omppi.MPPIOracle(dynamics, oenvs.RUNNING_COST[env], d, sig, K, T, 1.0, torch.tensor(-A), torch.tensor(A), A)

    def timed(mppi, n):
        out = []
        for _ in range(n):
            t0 = time.perf_counter()
            mppi.command(state, ab)
            out.append(time.perf_counter() - t0)
        return out

    if cfg["key"] != "1":
        out = _cpu_baseline_bounded(make, timed, info, K_full, T, budget_s)
        torch.set_num_threads(default_threads)
        return out
    t_start = time.perf_counter()
    with torch.no_grad():
        # 1. thread-count sweep on a 1/8 population (warm-up + one command each): torch's default of one thread per
        #    logical CPU is far from the best setting for these small FP64 ops on a many-core host
        K8 = max(K_full // 8, 128)
        small = make(K8)
        limit = info["usable_cpus"]
        cand = sorted({c for c in (8, 16, 32, info["physical_cores"]) if c <= limit} or {limit})
        sweep = []
        for nt in cand:
            torch.set_num_threads(nt)
            timed(small, 1)
            sweep.append((nt, timed(small, 1)[0]))
        best_nt, best_small = min(sweep, key=lambda x: x[1])
        # 2. the reported figure: full population, best thread count, one warm-up, median of >= 3
        torch.set_num_threads(best_nt)
        full = make(K_full)
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
        one = timed(small, 1)[0] * (K_full / K8)
        allc = dict(sweep).get(info["physical_cores"])
        allc = allc * (K_full / K8) if allc is not None else None
    torch.set_num_threads(default_threads)
    return dict(
        value=1.0 / med, unit="planning steps/s", cores=best_nt, kind="port",
        sample=(f"full workload (K={K_full}, T={T}): 1 warm-up + median of {n_rep} command() calls at {best_nt} threads "
                f"(runs {[round(r, 2) for r in reps]} s, warm-up {warm:.2f} s); thread count picked by a sweep on K={K8}: "
                f"{[(n, round(e, 3)) for n, e in sweep]} (threads, s)"),
        cpu_model=info["model"], physical_cores=info["physical_cores"], logical_cpus=info["logical_cpus"],
        usable_cpus=info["usable_cpus"],
        # NOT measurements at the quoted size: one command at K / 8, scaled by 8 (context for the measured figure above only)
        one_thread_extrapolated_x8=dict(value_extrapolated=1.0 / one, seconds_per_command_extrapolated=one, measured_at_K=K8,
                                        note=f"EXTRAPOLATED: one command at K={K8}, time x {K_full // K8}"),
        all_physical_cores_extrapolated_x8=(dict(value_extrapolated=1.0 / allc, seconds_per_command_extrapolated=allc,
                                                 threads=info["physical_cores"], measured_at_K=K8,
                                                 note=f"EXTRAPOLATED: one command at K={K8}, time x {K_full // K8}") if allc else None),
        torch=torch.__version__, oracle="oracle/ (torch-CPU float64, aten::gru encoder as in the reference)",
    )


def _cpu_baseline_bounded(make, timed, info, K_full, T, budget_s):
    """cpu_baseline for the configs other than the headline one (`--config 0 / 2 / 3 / 4 / d4`): the same oracle, on a budget.
    A calibration command sizes the sweep population so that one command takes about a second at 8 threads; candidates are
    swept in ascending thread count and the sweep stops once one is 1.5 x slower than the best (more threads only lose from
    there: 128 threads cost 34 s per command on the de Hoog oracle); the reported figure is measured at the FULL population when
    1 warm-up + 3 commands fit `budget_s`, else at the largest power-of-two fraction that does -- then `value` is null and
    `value_extrapolated` carries the figure scaled by K (the work is linear in K), labelled as such."""
    t_start = time.perf_counter()
    with torch.no_grad():
        limit = info["usable_cpus"]
        torch.set_num_threads(min(8, limit))
        K_cal = min(256, K_full)
        cal = make(K_cal)
        timed(cal, 1)
        t_cal = timed(cal, 1)[0]
        K_s = K_cal
        while K_s * 2 <= max(K_full // 8, K_cal) and t_cal * (K_s * 2 / K_cal) <= 1.0:
            K_s *= 2
        small = make(K_s) if K_s != K_cal else cal
        cand = sorted({c for c in (8, 16, 32, info["physical_cores"]) if c <= limit} or {limit})
        sweep = []
        for nt in cand:
            torch.set_num_threads(nt)
            timed(small, 1)
            sweep.append((nt, timed(small, 1)[0]))
            if len(sweep) >= 2 and sweep[-1][1] > 1.5 * min(e for _, e in sweep):
                break
        best_nt, best_small = min(sweep, key=lambda x: x[1])
        torch.set_num_threads(best_nt)
        left = budget_s - (time.perf_counter() - t_start)
        K_meas = K_full
        while K_meas > K_s and best_small * (K_meas / K_s) * 4 > max(left, 4.0):
            K_meas //= 2
        full = small if K_meas == K_s else make(K_meas)
        warm = timed(full, 1)[0]
        reps = timed(full, 3)
        med = statistics.median(reps)
        torch.set_num_threads(1)
        one_run = timed(small, 1)[0] if best_small * 16 < 20.0 else None  # (skipped when a 1-thread command would take minutes)
    scale = K_full / K_meas
    out = dict(
        unit="planning steps/s", cores=best_nt, kind="port",
        sample=(f"{'full workload' if K_meas == K_full else f'BOUNDED SAMPLE: 1/{K_full // K_meas} of the population'} (K={K_meas}, T={T}): "
                f"1 warm-up + median of 3 command() calls at {best_nt} threads (runs {[round(r, 2) for r in reps]} s, warm-up {warm:.2f} s); "
                f"thread count picked by a sweep on K={K_s}: {[(n, round(e, 3)) for n, e in sweep]} (threads, s; stopped at the first "
                f"candidate 1.5 x slower than the best)"),
        cpu_model=info["model"], physical_cores=info["physical_cores"], logical_cpus=info["logical_cpus"], usable_cpus=info["usable_cpus"],
        one_thread_extrapolated=(dict(value_extrapolated=1.0 / (one_run * K_full / K_s), measured_at_K=K_s,
                                      note=f"EXTRAPOLATED: one command at K={K_s}, time x {K_full / K_s:g}") if one_run else None),
        torch=torch.__version__, oracle="oracle/ (torch-CPU float64, aten::gru encoder as in the reference)",
    )
    if K_meas == K_full:
        out["value"] = 1.0 / med
    else:
        out["value"] = None  # NOT a measurement at the config's population:
        out["value_extrapolated"] = 1.0 / (med * scale)
        out["measured_at_K"] = K_meas
        out["note"] = (f"EXTRAPOLATED from K={K_meas} (time x {scale:g}); the full population would take ~{med * scale:.0f} s per command")
    return out


# the N > 1 plumbing -- supervisors, progress watchdog, self-launch -- lives in bench_supervisor.py (no GPU, no library there)
from bench_supervisor import (PROGRESS_ENV, SupervisorStore, free_port, kill_group, last_json_line, mark, run_watched,  # noqa: E402,F401
                              self_launch, supervise, watchdog_budgets)


def init_pg(backend, args, **kw):
    """torch.distributed group of a measuring process.  Under a supervisor (--worker with RANK set) the store at
    MASTER_ADDR:MASTER_PORT outlives the attempt, so the child joins it as a client under a key prefix of its own attempt --
    a second child must not meet the first one's rendezvous / barrier keys."""
    import torch.distributed as dist

    if not (args.worker and "RANK" in os.environ):
        return dist.init_process_group(backend, **kw)
    from datetime import timedelta

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    tcp = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"]), world, False,
                        timedelta(seconds=300), multi_tenant=True)
    store = dist.PrefixStore(f"nlcbench_attempt{args.attempt}/{os.environ.get('TORCHELASTIC_RUN_ID', 'run')}", tcp)
    return dist.init_process_group(backend, store=store, rank=rank, world_size=world, **kw)


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
        # the GPU box gets a snapshot without .git: the commit __graft_entry__.build() recorded next to the library
        return build_info().get("commit") or os.environ.get("NLC_COMMIT")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="1",
                    help="BASELINE.json configs[i] (default 1 = the headline metric's config) or d4 = north_star's literal "
                         "state_dim=4 cartpole; each at its whole population, sharded over --gpus")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ilt", action="store_true", help="skip the stand-alone ILT kernel section (experiments)")
    ap.add_argument("--no-sliced-encoder", action="store_true",
                    help="N = 1, default config: skip the run with the experimental int8-sliced encoder (`encoder_int8_sliced` in the line)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="N = 1, default config: skip the short runs of the other BASELINE configs (`other_configs` in the line); "
                         "profiling passes use this so that only the headline workload's kernels are counted")
    ap.add_argument("--cpu-budget", type=float, default=30.0)
    ap.add_argument("--strict-pmc", action="store_true",
                    help="fail instead of reporting `traffic: null` when the committed PMC summary was collected from other "
                         "kernel sources than this library was built from")
    ap.add_argument("--collective", choices=("auto", "torch", "native"), default="auto",
                    help="N > 1: the per-command all-gather inside nlc_mppi_finish on the library's own RCCL communicator "
                         "(include/nlc.h, nlc_comm_init) or through torch.distributed between the two phases; auto (default) "
                         "= the library's, falling back to torch's if any rank cannot bring the communicator up -- or, under "
                         "the supervisors, if the run with it fails or hangs")
    ap.add_argument("--no-also-collective", action="store_true",
                    help="N > 1, --collective auto: do not measure torch.distributed's collective in a second child after the "
                         "library's succeeded (also_collective)")
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
    ap.add_argument("--samples", type=int, default=None,
                    help="override K (experiments only; the headline metric is quoted at the config's own population)")
    ap.add_argument("--watchdog-init-s", type=float, default=300.0,
                    help="N > 1: a rank's child is killed when it has reported no progress for this long after its start "
                         "(covers `import torch` on a cold box, HIP and RCCL bring-up)")
    ap.add_argument("--watchdog-step-s", type=float, default=0.0,
                    help="N > 1: ... or no NEW progress marker for this long afterwards (0 = 90 s + 0.25 s per step)")
    # internal / tests
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)  # the measuring child of a supervisor
    ap.add_argument("--sliced-encoder-child", action="store_true", help=argparse.SUPPRESS)  # the `encoder_int8_sliced` section alone
    ap.add_argument("--attempt", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--test-hang-rank", type=int, default=-1, help=argparse.SUPPRESS)  # this rank's worker sleeps forever ...
    ap.add_argument("--test-hang-attempts", type=int, default=99, help=argparse.SUPPRESS)  # ... in attempts < this
    ap.add_argument("--dry-launch-no-fallback", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def main():
    global STRICT_PMC
    args = parse_args()
    STRICT_PMC = args.strict_pmc
    # stdout carries exactly ONE JSON line (driver contract).  Libraries write there too -- RCCL prints its
    # NCCL_DEBUG=VERSION banner (set by the box image) to stdout at communicator creation -- so everything else this
    # process and its libraries print goes to stderr, and the result line is written to the saved stdout at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    if args.sliced_encoder_child:
        # the experimental section in a process of its own (worker() starts it): a fault or a hang there cannot take the headline
        # line with it
        import neurallaplacecontrol_amd as nlc

        local = int(os.environ.get("LOCAL_RANK", 0))
        torch.cuda.set_device(local)
        with torch.no_grad():
            out = sliced_encoder_section(nlc, local, args.steps, args.warmup)
        os.write(result_fd, (json.dumps(out) + "\n").encode())
        sys.exit(0)
    if args.gpus > 1 and "RANK" not in os.environ:
        # invoked bare (`python bench.py --gpus N`): start the ranks ourselves, as a CHILD process (nothing in this
        # process has touched the GPU yet), relay its one JSON line and exit with its code
        sys.exit(self_launch(args.gpus, sys.argv[1:], result_fd, args))
    if "RANK" in os.environ and not args.worker:
        # under a launcher: this process supervises, a fresh child measures (see supervise())
        sys.exit(supervise(args, sys.argv[1:], result_fd))
    sys.exit(worker(args, result_fd))


def worker(args, result_fd):
    mark("start")
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        args.gpus = world
    import torch.distributed as dist

    def test_hang():
        if args.test_hang_rank == rank and args.attempt < args.test_hang_attempts:
            while True:  # (tests) a rank that never arrives: the watchdog's case
                time.sleep(3600)

    if args.dry_launch:
        # launch rehearsal (CPU test): every rank reports the environment torch.distributed.run gave it over a gloo
        # group; rank 0 prints them as the one JSON line.  No GPU call is made.
        envs = [None] * world
        if "RANK" in os.environ:
            init_pg("gloo", args)
            mark("pg_ready")
            test_hang()
            dist.all_gather_object(envs, {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR")})
            dist.destroy_process_group()
        else:
            envs = [{"RANK": None, "LOCAL_RANK": None, "WORLD_SIZE": None, "MASTER_ADDR": None}]
        if rank == 0:
            os.write(result_fd, (json.dumps(dict(dry_launch=True, n_gpus=world, ranks=envs, attempt=args.attempt,
                                                 config=dict(collective=args.collective, native_collective=False))) + "\n").encode())
        mark("line_written")
        return 0

    import neurallaplacecontrol_amd as nlc

    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    if args.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    coll_dev = "cpu" if args.rehearse_on_one_gpu else f"cuda:{local}"  # where the bench's own small collectives live
    pg = None
    if world > 1 or "RANK" in os.environ:  # under torch.distributed.run even a 1-rank job goes through RCCL
        if args.rehearse_on_one_gpu:
            init_pg("gloo", args)
        else:
            init_pg("nccl", args, device_id=torch.device("cuda", local))
        pg = dist.group.WORLD
    mark("pg_ready")
    test_hang()

    cfg = dict(CONFIGS[args.config], key=args.config)
    env_name, T, B, algo, S = cfg["env"], cfg["T"], cfg["B"], cfg["algo"], cfg["S"]
    d, nu, A, _ = ENV_SHAPES[env_name]
    K_total = args.samples if args.samples is not None else cfg["K"]
    headline = args.config == "1"

    model = synthetic_state_dict(d, nu, S, env=env_name, algo=algo).to(f"cuda:{local}")
    sd_cpu = {k: v.detach().cpu().to(torch.float64) for k, v in model.state_dict().items()}
    torch.cuda.synchronize()
    t_first = time.perf_counter()
    planner = nlc.MPPIDelay(
        nlc.NLDynamics(model, 0.05), nlc.EnvCost(env_name), d, nlc.noise_sigma(nu), num_samples=K_total, horizon=T,
        # device="cpu": U and the returned action live on the host, as the harness's env.step needs them (the merge kernel
        # stores the action straight into pinned host memory); every kernel runs on compute_device
        device="cpu", compute_device=f"cuda:{local}", lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A,
        noise_rng="philox", seed=0, process_group=pg, U_init=torch.zeros(T, nu, dtype=torch.float64),
        # every config keeps the (K, T, nx) / (K, T, nu) rollout like the reference (planners/mppi_delay.py:300-301; configs[3]:
        # 1 GB per command per GPU at N = 1)
        store_rollouts=True,
        planner_options=dict({} if args.collective == "auto" else {"native_collective": int(args.collective == "native")},
                             **{kv.split("=", 1)[0]: float(kv.split("=", 1)[1]) for kv in args.planner_opt}),
    )
    state = nlc.initial_state(env_name, torch.Generator().manual_seed(0))
    abuf = torch.zeros(B, nu, dtype=torch.float64)
    mark("planner_ready")

    last_action = [None]

    def step(ab):
        a = planner.command(state, ab)
        ab = torch.roll(ab, -1, dims=0)  # harness get_action (mppi_with_model.py:25-28)
        ab[-1] = last_action[0] = a.cpu()
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

    one_step()  # the first command configures the planner (and, sharded, is the first collective-bearing call)
    first_command_ms = (time.perf_counter() - t_first) * 1e3  # construction -> first action on the host (mppi_with_model.py:250-259)
    mark("first_command")
    preheat(one_step, args.preheat_ms, pg, coll_dev)
    abuf = holder[0]
    mark("preheat_done")
    for _ in range(args.warmup):
        abuf = step(abuf)
    fence()
    mark("warmup_done")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        abuf = step(abuf)
    fence()
    elapsed_local = elapsed = time.perf_counter() - t0
    action_after_timed = last_action[0].tolist()  # (tests: a sharded run must return the one-GPU run's action)
    if pg is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    mark("timed_done")
    # second, untimed pass: the same steps with hipEvent pairs around every launch (on the launch stream)
    planner.ctx.profile_reset()
    planner.ctx.profile(True)
    for _ in range(args.steps):
        abuf = step(abuf)
    fence()
    planner.ctx.profile(False)
    prof = planner.ctx.profile_read()
    mark("profile_done")
    kernels = {k: dict(avg_ms=v["total_ms"] / max(v["launches"], 1), launches=v["launches"],
                       per_command_ms=v["total_ms"] / max(args.steps, 1)) for k, v in prof.items()}

    # the collective by itself (N > 1): the all-gather of 2 + T*nu doubles per rank exactly as a command issues it, timed
    # over 50 calls between fences -- torch.distributed's here; the library's own appears as kernels["rccl_all_gather"]
    coll_timing = None
    if pg is not None:
        part = torch.zeros(2 + T * nu, dtype=torch.float64, device=f"cuda:{local}")
        gath = torch.empty(world * (2 + T * nu), dtype=torch.float64, device=f"cuda:{local}")
        from neurallaplacecontrol_amd.sharding import gather_partials

        for _ in range(5):
            gather_partials(part, gath, pg)
        fence()
        tc = time.perf_counter()
        for _ in range(50):
            gather_partials(part, gath, pg)
        torch.cuda.synchronize()
        coll_timing = dict(torch_all_gather_avg_ms=(time.perf_counter() - tc) / 50 * 1e3, backend=str(dist.get_backend(pg)),
                           library_all_gather_avg_ms=(kernels.get("rccl_all_gather") or {}).get("avg_ms"),
                           doubles_per_rank=2 + T * nu)
        fence()
    # what every rank saw (rank 0 prints it): "did the collective see N ranks", which body ran, per-rank kernel averages
    info = planner.ctx.device_info()
    mine = dict(rank=rank, local_rank=local, device=info["name"], torch_world=(dist.get_world_size(pg) if pg is not None else 1),
                library_comm_world=int(planner.ctx.get_stat("comm_world")), native_collective=bool(planner.native_collective),
                rollout_body=planner.rollout_body, fused_timeouts=planner.fused_timeouts, fused_fallbacks=planner.fused_fallbacks,
                timed_s=elapsed_local, kernels_avg_ms={k: v["avg_ms"] for k, v in kernels.items()},
                collective_timing=coll_timing)
    per_rank = [mine]
    if pg is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    mark("ranks_gathered")

    # EXPERIMENTAL, reported beside the headline and never as `value`: the same steps with the encoder's hidden-state GEMMs as
    # int8-sliced fixed-point products on the INT8 matrix pipe (planner option gru_gemm = 1, csrc/kernels_gru_i8.hip)
    sliced = None
    if world == 1 and headline and args.samples is None and not args.no_sliced_encoder:
        sliced = sliced_encoder_child(args, local)
        mark("sliced_encoder_done")
    # (measured right behind the headline's own steps: the same thermal state, before the long stand-alone sections)

    pmc_name, pj = load_pmc()

    # ---- stand-alone ILT kernel at N = K*T points (the BASELINE 'ILT GB/s vs HBM peak' figure; headline shape d=5)
    ilt = None
    if rank == 0 and not args.no_ilt:
        ilt = standalone_ilt_section(nlc, local, pmc_name, pj)
        mark("ilt_done")

    def teardown():
        # orderly end: rank 0 arrives late (stand-alone ILT section); nobody tears a communicator down under a peer
        if pg is not None:
            fence()
            if planner.native_collective:
                planner.ctx.comm_destroy()
            dist.destroy_process_group()
        mark("teardown_done")

    if rank != 0:
        teardown()
        return 0

    k_local = K_total // world
    windows = k_local * T
    g_hidden = HIDDEN // 2
    nt3 = int(planner.ctx.get_stat("model_nt3"))
    gru_need = flops_gru_needed_per_window(g_hidden, nu, B) * windows
    gru_iss = flops_gru_issued_per_window(g_hidden, B) * windows
    roll_need = flops_rollout_needed_per_sample_step(HIDDEN, d, S) * windows
    roll_iss = flops_rollout_issued_per_sample_step(HIDDEN, nt3) * windows

    def mfma_entry(kernel, need, issued, alg_bytes, launches_per_command=1):
        k = kernels[kernel]
        sec = k["avg_ms"] * 1e-3
        need, issued, alg_bytes = need / launches_per_command, issued / launches_per_command, alg_bytes / launches_per_command
        traffic, src = (pmc_traffic(pmc_name, pj, kernel) if headline and k_local == K_SAMPLES and kernel in PMC_KEYS else (None, None))
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
        if "nl_rollout_kernel" in kernels:
            roofline["also"] = mfma_entry("nl_rollout_kernel", roll_need, roll_iss, (16 + 8 * (2 * nu + d)) * windows)
        elif "nl_repfunc_kernel" in kernels:
            # staged step chain (de Hoog): the representation MLP is one launch per horizon step and stream part; F_k leaves
            # the launch (2 d S doubles per sample-step written, read again by the de Hoog kernel)
            n_launch = max(kernels["nl_repfunc_kernel"]["launches"] // max(args.steps, 1), 1)
            roofline["also"] = mfma_entry("nl_repfunc_kernel", roll_need - 2 * d * S * windows, roll_iss - 2 * nt3 * 128 * windows,
                                          (16 + 8 * d + 16 * d * S) * windows, launches_per_command=n_launch)
    # whole step against the same roof: every algorithmic flop of one command() (GRU encode + rollout, all ranks) over the
    # measured wall time per step -- the sampling / weighting / merge kernels add time but no matrix flops (nor does the
    # de Hoog recurrence: FP64 VALU work, not counted)
    step_flops = (gru_need + roll_need) * world
    roofline["step"] = dict(flops=step_flops, achieved=step_flops / (elapsed / args.steps) / 1e12, peak=FP64_MFMA_PEAK_TFLOPS * world,
                            unit="TFLOP/s", frac=step_flops / (elapsed / args.steps) / 1e12 / (FP64_MFMA_PEAK_TFLOPS * world),
                            note="algorithmic flops of one command() (GRU encode + rollout; SURVEY 8d) / ms_per_step / FP64-MFMA peak")
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(sd_cpu, cfg, args.cpu_budget)
        mark("cpu_baseline_done")
    # the other BASELINE configs, each at its whole population on this GPU, in the SAME driver-run line (VERDICT r4 missing 3):
    # short fenced loops, no profiling pass, no CPU baseline -- `bench.py --config k` gives each one's full line
    others = None
    if world == 1 and headline and args.samples is None and not args.no_other_configs:
        others = other_configs_section(nlc, local)
        mark("other_configs_done")
    workload = (f"{env_name} (nx={d}, nu={nu}), K={K_total} MPPI samples sharded over the ranks, H={T}, "
                f"action_buffer_size={B}, NL dynamics h={HIDDEN} S={S} {algo} ILT")
    workload += f" ({cfg['name']})" if K_total == cfg["K"] else f" -- EXPERIMENT: not {cfg['name']}'s population of {cfg['K']}"
    if args.planner_opt:
        workload += f" -- EXPERIMENT: planner options {', '.join(args.planner_opt)}"
    if args.rehearse_on_one_gpu:
        workload += " -- REHEARSAL: every rank on cuda:0 over gloo, not a measurement"
    bi = build_info()
    out = dict(
        metric=f"MPPI planning steps/sec ({K_total} samples, H={T})",
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
        config=dict(workload=workload, baseline_config=args.config, samples_per_gpu=k_local, noise="device Philox4x32-10",
                    device=info["name"], commit=git_commit(), library_build=bi or None, preheat_ms=args.preheat_ms,
                    store_rollouts=bool(planner.store_rollouts), first_command_ms=first_command_ms,
                    last_action=action_after_timed,
                    planner_options=(args.planner_opt or None),
                    attempt=args.attempt, native_collective=bool(planner.native_collective),
                    collective=None if pg is None else ("rccl all-gather inside nlc_mppi_finish (library communicator)"
                                                        if planner.native_collective else
                                                        f"{dist.get_backend(pg)} all-gather via torch.distributed between the two phases"),
                    collective_timing=coll_timing,
                    ranks_seen=dict(torch_world=mine["torch_world"], library_comm_world=mine["library_comm_world"],
                                    devices=[r["device"] for r in per_rank], per_rank=per_rank)),
        roofline=roofline,
        roofline_ilt=ilt,
        cpu_baseline=cpu,
        kernels_avg_ms=kernels,
        kernels_note="per-launch hipEvent averages from a second, untimed pass of the same steps",
        other_configs=others,
        encoder_int8_sliced=sliced,
    )
    if cpu:
        ref = cpu["value"] if cpu.get("value") else cpu.get("value_extrapolated")
        out["speedup_vs_cpu_baseline" if cpu.get("value") else "speedup_vs_cpu_baseline_extrapolated"] = out["value"] / ref
    sys.stdout.flush()
    os.write(result_fd, (json.dumps(out) + "\n").encode())
    mark("line_written")
    teardown()
    return 0


MFMA_KERNELS = ("gru_encode_kernel", "nl_rollout_kernel", "nl_plan_fused_kernel", "nl_repfunc_kernel", "nl_dehoog_chain_kernel")


def dominant_kernel_roofline(kernels, d, nu, B, S, nt3, windows):
    """The matrix kernel that takes the most time per command() in `kernels` (name -> avg_ms / launches / per_command_ms from the
    library's hipEvent pairs) against the FP64-MFMA peak: the algorithm's flops of ALL its launches of one command (the
    W_hh h0 = 0 products skipped) over the time they take.  For the other configs' entries and the sliced section."""
    have = [k for k in MFMA_KERNELS if k in kernels]
    if not have:
        return None
    name = max(have, key=lambda k: kernels[k]["per_command_ms"])
    gru = flops_gru_needed_per_window(HIDDEN // 2, nu, B) * windows
    roll = flops_rollout_needed_per_sample_step(HIDDEN, d, S) * windows
    need = {"gru_encode_kernel": gru, "nl_rollout_kernel": roll, "nl_plan_fused_kernel": gru + roll,
            # the staged / persistent de Hoog chain: the representation MLP's flops (the ILT sum is not a matrix product there)
            "nl_repfunc_kernel": roll - 2 * d * S * windows, "nl_dehoog_chain_kernel": roll - 2 * d * S * windows}[name]
    k = kernels[name]
    sec = k["per_command_ms"] * 1e-3
    return dict(bound="mfma", kernel=name, launches_per_command=k["launches_per_command"], avg_launch_ms=k["avg_ms"],
                per_command_ms=k["per_command_ms"], flops_per_command=need, achieved=need / sec / 1e12, peak=FP64_MFMA_PEAK_TFLOPS,
                unit="TFLOP/s", frac=need / sec / 1e12 / FP64_MFMA_PEAK_TFLOPS)


def profiled_pass(planner, step, ab, steps):
    """`steps` more commands with the library's per-launch hipEvent pairs on: name -> averages.  Returns (kernels, ab)."""
    planner.ctx.profile_reset()
    planner.ctx.profile(True)
    for _ in range(steps):
        ab = step(ab)
    torch.cuda.synchronize()
    planner.ctx.profile(False)
    prof = planner.ctx.profile_read()
    return {k: dict(avg_ms=v["total_ms"] / max(v["launches"], 1), launches=v["launches"], launches_per_command=v["launches"] / max(steps, 1),
                    per_command_ms=v["total_ms"] / max(steps, 1)) for k, v in prof.items()}, ab


def other_configs_section(nlc, local, steps=12, warmup=3, keys=("0", "2", "3", "4", "d4"), planner_options=None):
    """Planning steps/s of BASELINE configs[0], [2], [3], [4] and north_star's literal state_dim = 4 shape, each at its WHOLE
    population on this one GPU (configs[2] / [3] are worded for 2 / 8 GPUs: `--config k --gpus N` shards them): `warmup` + `steps`
    fenced commands after a time-boxed pre-heat (the de Hoog planner measures its chain forms during its first half second).
    Like the headline -- and like the reference, which always keeps them (planners/mppi_delay.py:300-301) -- every planner here
    STORES its (K, T, nx) states and (K, T, nu) actions (configs[3]: 1 GB per command); each entry says so, names its dominant
    kernel's roofline fraction (a second, profiled pass of the same steps) and what the first command costs (`first_command_ms`:
    planner construction -> first action on the host, i.e. model upload, configuration, workspace and any auto-tuning)."""
    out = {}
    for key in keys:
        cfg = CONFIGS[key]
        env_name, T, B, algo, S, K = cfg["env"], cfg["T"], cfg["B"], cfg["algo"], cfg["S"], cfg["K"]
        d, nu, A, _ = ENV_SHAPES[env_name]
        model = synthetic_state_dict(d, nu, S, env=env_name, algo=algo).to(f"cuda:{local}")
        torch.cuda.synchronize()
        t_first = time.perf_counter()
        planner = nlc.MPPIDelay(
            nlc.NLDynamics(model, 0.05), nlc.EnvCost(env_name), d, nlc.noise_sigma(nu), num_samples=K, horizon=T, device="cpu",
            compute_device=f"cuda:{local}", lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox",
            seed=0, U_init=torch.zeros(T, nu, dtype=torch.float64), store_rollouts=True, planner_options=planner_options)
        state = nlc.initial_state(env_name, torch.Generator().manual_seed(0))
        ab = torch.zeros(B, nu, dtype=torch.float64)

        def step(ab):
            a = planner.command(state, ab)
            ab = torch.roll(ab, -1, dims=0)
            ab[-1] = a.cpu()
            return ab

        ab = step(ab)  # (the action is on the host when command() returns: device="cpu")
        first_ms = (time.perf_counter() - t_first) * 1e3
        t_end = time.perf_counter() + (0.8 if algo == "dehoog" else 0.1)
        while time.perf_counter() < t_end:
            ab = step(ab)
        for _ in range(warmup):
            ab = step(ab)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ab = step(ab)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        windows = K * T
        flops = (flops_gru_needed_per_window(HIDDEN // 2, nu, B) + flops_rollout_needed_per_sample_step(HIDDEN, d, S)) * windows
        kern, ab = profiled_pass(planner, step, ab, steps)
        out[key] = dict(name=cfg["name"], workload=f"{env_name} (nx={d}, nu={nu}), K={K}, H={T}, action_buffer_size={B}, h={HIDDEN} S={S} {algo}",
                        value=steps / el, unit="planning steps/s", ms_per_step=el / steps * 1e3, steps=steps, warmup=warmup,
                        store_rollouts=True, rollout_body=planner.rollout_body, gpus_the_config_is_worded_for=cfg["gpus"],
                        first_command_ms=first_ms,
                        roofline_step_frac=flops / (el / steps) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                        roofline=dominant_kernel_roofline(kern, d, nu, B, S, int(planner.ctx.get_stat("model_nt3")), windows),
                        kernels_per_command_ms={k: v["per_command_ms"] for k, v in kern.items()})
        del planner, model
        torch.cuda.empty_cache()
    return out


def sliced_encoder_child(args, local, timeout_s=240):
    """`encoder_int8_sliced` measured by a CHILD process (same interpreter, `--sliced-encoder-child`), started right behind the
    headline's steps: the section is experimental, and a fault, an exception or a hang in it must not cost the headline line."""
    cmd = [sys.executable, os.path.abspath(__file__), "--sliced-encoder-child", "--steps", str(args.steps), "--warmup", str(args.warmup)]
    env = dict(os.environ, LOCAL_RANK=str(local))
    for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # polled, with a progress marker every few seconds: under a launcher this worker is itself watched (bench_supervisor.
    # run_watched allows ~100 s between markers) -- a slow or hung child must cost this section, not the headline line
    with tempfile.TemporaryFile() as out_f, tempfile.TemporaryFile() as err_f:
        proc = subprocess.Popen(cmd, stdout=out_f, stderr=err_f, env=env, start_new_session=True)
        t0 = time.time()
        while proc.poll() is None and time.time() - t0 < timeout_s:
            time.sleep(0.5)
            if int(time.time() - t0) % 5 == 0:
                mark("sliced_encoder_running")
        if proc.poll() is None:
            kill_group(proc)
            return dict(error=f"the section's process did not finish within {timeout_s} s")
        out_f.seek(0)
        err_f.seek(0)
        stdout, stderr, rc = out_f.read().decode(), err_f.read().decode(errors="replace"), proc.returncode
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    if rc != 0 or not lines:
        return dict(error=f"the section's process exited with {rc}", stderr_tail=stderr[-400:])
    return json.loads(lines[-1])


def sliced_encoder_section(nlc, local, steps, warmup):
    """BASELINE configs[1] once more with planner option gru_gemm = 1: the GRU encoder's hidden-state GEMMs as int8-sliced fixed-point
    products on the INT8 matrix pipe (csrc/kernels_gru_i8.hip: 54-bit fixed point in seven signed digits, exact integer
    accumulation, FP64 recombination) instead of FP64 MFMAs -- same fenced loop as the headline, the per-kernel averages from a
    second pass, and how far the two encoders' latents are apart on 65 536 random action windows."""
    cfg = CONFIGS["1"]
    env_name, T, B, S, K = cfg["env"], cfg["T"], cfg["B"], cfg["S"], cfg["K"]
    d, nu, A, _ = ENV_SHAPES[env_name]
    model = synthetic_state_dict(d, nu, S, env=env_name).to(f"cuda:{local}")
    planner = nlc.MPPIDelay(
        nlc.NLDynamics(model, 0.05), nlc.EnvCost(env_name), d, nlc.noise_sigma(nu), num_samples=K, horizon=T, device="cpu",
        compute_device=f"cuda:{local}", lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox",
        seed=0, U_init=torch.zeros(T, nu, dtype=torch.float64), store_rollouts=True, planner_options={"gru_gemm": 1})
    state = nlc.initial_state(env_name, torch.Generator().manual_seed(0))
    ab = torch.zeros(B, nu, dtype=torch.float64)

    def step(ab):
        a = planner.command(state, ab)
        ab = torch.roll(ab, -1, dims=0)
        ab[-1] = a.cpu()
        return ab

    t_end = time.perf_counter() + 0.3
    ab = step(ab)
    while time.perf_counter() < t_end:
        ab = step(ab)
    for _ in range(warmup):
        ab = step(ab)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ab = step(ab)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    kern_full, ab = profiled_pass(planner, step, ab, max(steps // 2, 2))
    kern = {k: v["avg_ms"] for k, v in kern_full.items()}
    uses = planner.ctx.get_stat("gru_gemm")
    g = torch.Generator().manual_seed(11)
    win = ((torch.rand(65536, B, nu, dtype=torch.float64, generator=g) * 2 - 1) * A).to(f"cuda:{local}")
    mctx = model.hip_ctx(torch.device(f"cuda:{local}"))
    with torch.no_grad():
        mctx.set_option("gru_gemm", 0)
        lat_f64 = model.encode_actions(win)
        mctx.set_option("gru_gemm", 1)
        lat_i8 = model.encode_actions(win)
        mctx.set_option("gru_gemm", 0)
    gru_flops = flops_gru_needed_per_window(HIDDEN // 2, nu, B) * K * T
    enc_s = kern.get("gru_encode_kernel", float("nan")) * 1e-3
    # the sliced encoder's own roofline: what it ISSUES.  Per 16-window tile: 12 gate tiles of layer 1's input side in GRU step 0,
    # 36 in every later step (W_hh0, W_ih1, W_hh1), 34 v_mfma_i32_16x16x64_i8 each (digit pairs of level >= 5 of 7 x 7); the layer-0
    # input GEMM (3 g/16 per step) and linear_out (g/4) stay v_mfma_f64_16x16x4.  One i8 MFMA = 16 x 16 x 64 x 2 integer ops.
    g_h = HIDDEN // 2
    tiles = K * T / 16.0
    i8_mfma = tiles * (12 + 36 * (B - 1)) * 34
    f64_mfma = tiles * (B * 3 * g_h // 16 + g_h // 4)
    info = planner.ctx.device_info()
    simds = info["num_cus"] * 4
    clk = info["clock_mhz"] * 1e6
    i8_roofline = dict(
        bound="int8 mfma issue + fp64 valu (see note)", kernel="gru_encode_i8_kernel", avg_launch_ms=enc_s * 1e3,
        i8_mfma_per_launch=i8_mfma, f64_mfma_per_launch=f64_mfma, int8_tops_issued=i8_mfma * 32768 / enc_s / 1e12,
        # one i8 16x16x64 MFMA occupies a SIMD's matrix pipe for 8 clocks at two wavefronts per SIMD (tools/ubench_i8emu.hip:
        # 9.0 measured), an FP64 16x16x4 for 64
        matrix_pipe_busy_frac=(i8_mfma * 8 + f64_mfma * 64) / simds / (enc_s * clk),
        fp64_equivalent_tflops=gru_flops / enc_s / 1e12, frac_of_fp64_mfma_peak=gru_flops / enc_s / 1e12 / FP64_MFMA_PEAK_TFLOPS,
        note="fp64_equivalent_tflops counts the algorithm's FP64 flops (SURVEY 8d) over the launch's time and may pass the FP64-MFMA "
             "peak: the products run on the INT8 pipe.  The kernel is bound by FP64 VALU issue beside the MFMAs (recombination, gate "
             "math, digit cut: ~4 500 VALU per tile-step against 1 224 MFMAs; profiles/r5_i8_gemm.md), clock from nlc_device_info")
    out = dict(planner_option="gru_gemm=1", encoder_kernel="gru_encode_i8_kernel" if uses else "gru_encode_kernel (option not taken)",
               dtype="int8x7-sliced 54-bit fixed point (hidden-state GEMMs), FP64 elsewhere; FP64-equivalent error bound",
               value=steps / el, unit="planning steps/s", ms_per_step=el / steps * 1e3, steps=steps, warmup=warmup, store_rollouts=True,
               kernels_avg_ms=kern, roofline=i8_roofline if uses else None,
               latents_max_abs_diff_vs_fp64_encoder=float((lat_i8 - lat_f64).abs().max()),
               latents_max_abs=float(lat_f64.abs().max()),
               encoder_fp64_equivalent_tflops=gru_flops / enc_s / 1e12,
               note="EXPERIMENTAL and not the headline: `value` above is measured with the FP64-MFMA encoder.  Operands: GRU states and "
                    "row-scaled weights as 54-bit fixed point; error against the exact product within 5 x 2^-53 of the row's sum of "
                    "|w h| for ordinary state magnitudes, as the FP64 MFMA chain's, and never above 2^-48 of the row scale "
                    "(tools/i8gemm_check.hip, tools/i8_adversarial.py: profiles/r6_i8_adversarial.json).  The rollout launch behind this "
                    "encoder runs at a ~5 % lower clock than behind the FP64 one (power management, profiles/r6_rollout_giveback.md)")
    del planner, model
    torch.cuda.empty_cache()
    # the option on two more BASELINE configs (short fenced loops as in other_configs): configs[4] (de Hoog, S = 33: the encoder is
    # 2.99 of its 6.3 ms) and configs[3] (acrobot: two action dims, K = 262144, T = 60)
    out["other_configs"] = {k: v for k, v in other_configs_section(nlc, local, steps=10, warmup=2, keys=("4", "3"),
                                                                   planner_options={"gru_gemm": 1}).items()}
    return out


def standalone_ilt_section(nlc, local, pmc_name, pj):
    """Stand-alone ILT kernels at N = 16384 * 40 points, d = 5 (the headline shape): Fourier S = 17 (HBM-bound), de Hoog
    S = 33, fixed Talbot S = 17, Fourier backward."""
    d = 5
    N = K_SAMPLES * HORIZON
    g = torch.Generator(device="cuda").manual_seed(1)
    theta = (torch.rand(N, d, S_TERMS, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * np.pi
    phi = (torch.rand(N, d, S_TERMS, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * (np.pi / 2)
    tt = torch.full((N,), 0.125, dtype=torch.float64, device="cuda")
    from neurallaplacecontrol_amd.laplace import default_ctx

    ictx = default_ctx(local)

    def heat(call, ms=60.0):
        """Untimed launches for `ms` before a timed window, like the planner's --preheat-ms: a stream that starts on a quiet chip
        runs through a power-management transient 2-5 ms after its first launch (the launches of that phase take 20-30 % longer,
        then the duration settles: profiles/r6_ilt_burst_trace.txt) -- a 20-launch window right behind three warm-up calls sits in it."""
        t_end = time.perf_counter() + ms * 1e-3
        while time.perf_counter() < t_end:
            for _ in range(8):
                call()
            torch.cuda.synchronize()  # (the host enqueues far faster than these kernels run: keep the clock on GPU time)

    heat(lambda: nlc.ilt_reconstruct(theta, phi, tt))
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
    heat(lambda: nlc.ilt_reconstruct(theta, phi, tt, "dehoog"))
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
    # ... and the ablation's other side at the same term count: the Fourier series with 33 terms (the row kernel with ONE tile in
    # flight per wavefront: two region pairs of 64 x 33 doubles do not fit four wavefronts' LDS)
    heat(lambda: nlc.ilt_reconstruct(theta, phi, tt))
    ictx.profile_reset()
    ictx.profile(True)
    for _ in range(10):
        nlc.ilt_reconstruct(theta, phi, tt)
    torch.cuda.synchronize()
    ictx.profile(False)
    p = ictx.profile_read()["ilt_fourier_kernel"]
    ms5 = p["total_ms"] / p["launches"]
    ilt["fourier33"] = dict(bound="hbm", kernel="ilt_fourier_kernel (rows, S = 33)", avg_launch_ms=ms5, points=N, algorithmic_bytes=nb2,
                            achieved=nb2 / (ms5 * 1e-3) / 1e9, unit="GB/s", frac=nb2 / (ms5 * 1e-3) / 1e9 / HBM_PEAK_GBS, traffic=None)
    del theta, phi
    # fixed Talbot at the Fourier kernel's shape: the same coalesced stream with the algorithm's per-term phase / weight
    theta = (torch.rand(N, d, S_TERMS, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * np.pi
    phi = (torch.rand(N, d, S_TERMS, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * (np.pi / 2) * 0.9
    heat(lambda: nlc.ilt_reconstruct(theta, phi, tt, "fixed_tablot"))
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
    # ONE forward, then the backward launches alone (retain_graph): a forward launch per iteration would mix 13 launches on other
    # inputs into the rocprofv3 average of the forward kernel above, which has to agree with `avg_launch_ms`
    xb = nlc.ilt_reconstruct(theta, phi, tt)
    heat(lambda: torch.autograd.grad(xb, (theta, phi), gx, retain_graph=True))
    ictx.profile_reset()
    ictx.profile(True)
    for it in range(10):
        torch.autograd.grad(xb, (theta, phi), gx, retain_graph=True)
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
    return ilt


if __name__ == "__main__":
    main()
