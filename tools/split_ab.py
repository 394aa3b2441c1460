"""Same-box A/B of library builds that differ in the latency-split bodies' compile-time knobs (tools/build_split_variants.sh):
per build, in a process of its own, the workloads those bodies serve --

  fused2048   one 8-GPU shard of the headline config on the fused one-launch body (K = 2048, T = 40)
  split2048   the same shard on GRU launch + latency-split rollout launch (rollout_variant 2)
  split8192   K = 8192 (one 2-GPU shard), two launches
  cfg5        BASELINE configs[4]: de Hoog S = 33, K = 16384 (per-step representation launch on the split tile)

-- each with ms per command (wall, fenced), the per-kernel hipEvent averages and a checksum of the actions of the first
commands (the knobs must not change a bit).  Builds are interleaved `--rounds` times.

    python tools/split_ab.py [--rounds 2] [--only fused2048,cfg5] tools/_ab/libnlc_base.so tools/_ab/libnlc_p3.so ..."""
import argparse, hashlib, json, os, subprocess, sys, time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKLOADS = {
    "fused2048": dict(K=2048, algo="fourier", S=17, opts={"rollout_variant": 3}, steps=200),
    "split2048": dict(K=2048, algo="fourier", S=17, opts={"rollout_variant": 2}, steps=200),
    "split8192": dict(K=8192, algo="fourier", S=17, opts={"rollout_variant": 2}, steps=100),
    "cfg5": dict(K=16384, algo="dehoog", S=33, opts={"dehoog_chain": 0, "dehoog_streams": 2}, steps=40),
}


def child(lib, only):
    sys.path.insert(0, REPO)
    from neurallaplacecontrol_amd import _lib
    _lib.use_library(lib)
    import torch
    import bench
    import neurallaplacecontrol_amd as nlc

    out = {}
    d, nu, A, T = 5, 1, 3.0, 40
    for name, w in WORKLOADS.items():
        if only and name not in only:
            continue
        model = bench.synthetic_state_dict(d, nu, w["S"], algo=w["algo"]).to("cuda")
        p = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(bench.ENV), d, nlc.noise_sigma(nu), w["K"], T, "cpu", lambda_=1.0,
                          u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox", seed=0,
                          U_init=torch.zeros(T, nu, dtype=torch.float64), planner_options=w["opts"])
        st, ab = nlc.initial_state(bench.ENV, torch.Generator().manual_seed(0)), torch.zeros(4, nu, dtype=torch.float64)
        h = hashlib.sha256()
        with torch.no_grad():
            for i in range(12):
                a = p.command(st, ab)
                if i < 6:
                    h.update(a.numpy().tobytes())
                    h.update(p.cost_total.cpu().numpy().tobytes())
                ab = torch.roll(ab, -1, 0); ab[-1] = a
            t_end = time.perf_counter() + 0.3
            while time.perf_counter() < t_end:
                p.command(st, ab)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(w["steps"]):
                a = p.command(st, ab)
                ab = torch.roll(ab, -1, 0); ab[-1] = a
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / w["steps"] * 1e3
            p.ctx.profile_reset(); p.ctx.profile(True)
            for _ in range(w["steps"] // 2):
                p.command(st, ab)
            torch.cuda.synchronize()
            p.ctx.profile(False)
            prof = {k: round(v["total_ms"] / max(v["launches"], 1), 5) for k, v in p.ctx.profile_read().items()}
        out[name] = dict(ms_per_command=round(ms, 4), body=p.rollout_body, kernels_avg_ms=prof, checksum=h.hexdigest()[:16])
        del p, model
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2], [x for x in sys.argv[3].split(",") if x] if len(sys.argv) > 3 else [])
        sys.exit(0)
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--only", default="")
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    rows = []
    for rnd in range(a.rounds):
        for lib in a.libs:
            res = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", os.path.abspath(lib), a.only],
                                 stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
            line = [ln for ln in res.stdout.decode().splitlines() if ln.startswith("{")]
            if res.returncode != 0 or not line:
                rows.append(dict(lib=os.path.basename(lib), round=rnd, error=f"exit {res.returncode}"))
                print(rows[-1], file=sys.stderr, flush=True)
                continue
            r = json.loads(line[-1])
            rows.append(dict(lib=os.path.basename(lib), round=rnd, **r))
            print(os.path.basename(lib), rnd, {k: (v["ms_per_command"], v["checksum"]) for k, v in r.items()}, file=sys.stderr, flush=True)
    # summary: best (minimum) ms per workload and build, checksums must agree across builds
    summary, sums = {}, {}
    for r in rows:
        for k, v in r.items():
            if isinstance(v, dict) and "ms_per_command" in v:
                cur = summary.setdefault(k, {}).get(r["lib"])
                summary[k][r["lib"]] = v["ms_per_command"] if cur is None else min(cur, v["ms_per_command"])
                sums.setdefault(k, set()).add(v["checksum"])
    print(json.dumps(dict(best_ms_per_command=summary, bit_identical_across_builds={k: len(v) == 1 for k, v in sums.items()}, rows=rows),
                     indent=1))
