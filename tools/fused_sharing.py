"""the fused one-launch planner body when the GPU is NOT the planner's alone.

The reference's own deployment is a multiprocessing.Pool(12) of evaluation workers on one GPU, one MPPIDelay of K = 1000,
T = 40 each (run_exp_multi.py:145-165, config.py:52).  `rollout_variant` auto picks the fused body for every K <= 4096, and
that body "assumes the device to itself" (include/nlc.h): its rollout workgroups wait for encoder workgroups of the same
launch.  Here P planner processes (P <= 6: the GPU pool's process guard) share cuda:0, every one counting from its FIRST
command, and each reports

  * the body its first and its last command ran on (nlc_get_stat "rollout_body"), fused_timeouts / fused_fallbacks and the
    index of the last command that saw a give-up -- did the fused body survive the sharing, or did the ctx silently fall
    back to the two-launch body?
  * planning steps/s and the slowest single command (a give-up costs `fused_spin_limit` polls);
  * a checksum of its actions, so that the bodies can be compared bit for bit across runs.

    python tools/fused_sharing.py [--procs 1,3,6] [--bodies auto,2,3] [--commands 300] [--K 1000] [--T 40] [--host-spin 1]
One JSON document on stdout (profiles/r5_fused_sharing.json)."""
import argparse, hashlib, json, os, sys, time
import multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def worker(idx, P, K, T, opts, commands, barrier, q):
    import torch
    import bench
    import neurallaplacecontrol_amd as nlc

    torch.set_num_threads(1)
    d, nu = 5, 1
    model = bench.synthetic_state_dict(d, nu, bench.S_TERMS).to("cuda:0")
    state = nlc.initial_state(bench.ENV, torch.Generator().manual_seed(idx))
    p = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(bench.ENV), d, nlc.noise_sigma(nu), num_samples=K, horizon=T,
                      device="cpu", compute_device="cuda:0", lambda_=1.0, u_min=torch.tensor(-3.0), u_max=torch.tensor(3.0),
                      u_scale=3.0, noise_rng="philox", seed=idx, U_init=torch.zeros(T, nu, dtype=torch.float64),
                      planner_options=opts)
    ab = torch.zeros(4, nu, dtype=torch.float64)
    torch.cuda.synchronize()
    barrier.wait()  # everybody's FIRST command starts here: nothing is warmed up, nothing is swallowed
    h = hashlib.sha256()
    lat, bodies = [], []
    t0 = time.perf_counter()
    for i in range(commands):
        tc = time.perf_counter()
        a = p.command(state, ab)
        lat.append(time.perf_counter() - tc)
        if i < 3 or i == commands - 1:
            bodies.append(p.rollout_body)
        ab = torch.roll(ab, -1, 0); ab[-1] = a
        h.update(a.numpy().tobytes())
    wall = time.perf_counter() - t0
    lat_sorted = sorted(lat)
    q.put(dict(idx=idx, commands=commands, wall_s=round(wall, 4), steps_per_s=round(commands / wall, 1),
               body_first=bodies[0], body_last=bodies[-1], fused_timeouts=p.fused_timeouts, fused_fallbacks=p.fused_fallbacks,
               last_giveup_command=int(p.ctx.get_stat("last_giveup_command")),
               first_command_ms=round(lat[0] * 1e3, 3), median_command_ms=round(lat_sorted[len(lat) // 2] * 1e3, 4),
               slowest_command_ms=round(lat_sorted[-1] * 1e3, 3), slowest_command_index=lat.index(lat_sorted[-1]),
               commands_over_50ms=sum(1 for x in lat[1:] if x > 0.05), actions_sha256=h.hexdigest()[:16]))
    barrier.wait()


def run(P, K, T, opts, commands):
    ctx = mp.get_context("spawn")
    barrier, q = ctx.Barrier(P), ctx.Queue()
    procs = [ctx.Process(target=worker, args=(i, P, K, T, opts, commands, barrier, q)) for i in range(P)]
    for pr in procs:
        pr.start()
    res = sorted((q.get(timeout=600) for _ in range(P)), key=lambda r: r["idx"])
    for pr in procs:
        pr.join(timeout=60)
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", default="1,3,6")
    ap.add_argument("--bodies", default="auto,2,3")
    ap.add_argument("--K", type=int, default=1000)  # config.py:52 mppi_roll_outs
    ap.add_argument("--T", type=int, default=40)
    ap.add_argument("--commands", type=int, default=300)
    ap.add_argument("--host-spin", type=int, default=1)
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE")
    a = ap.parse_args()
    extra = {kv.split("=", 1)[0]: float(kv.split("=", 1)[1]) for kv in a.opt}
    rows, solo = [], {}
    for body in a.bodies.split(","):
        for P in [int(x) for x in a.procs.split(",")]:
            opts = dict({"host_spin": a.host_spin}, **extra)
            if body != "auto":
                opts["rollout_variant"] = int(body)
            res = run(P, a.K, a.T, opts, a.commands)
            for r in res:  # the same seed planned alone (P = 1 covers idx 0) or under another body must give the same actions
                solo.setdefault(r["idx"], r["actions_sha256"])
            row = dict(body=body, procs=P, aggregate_steps_per_s=round(sum(r["steps_per_s"] for r in res), 1),
                       fused_timeouts=sum(r["fused_timeouts"] for r in res), fused_fallbacks=sum(r["fused_fallbacks"] for r in res),
                       bodies_first=sorted({str(r["body_first"]) for r in res}), bodies_last=sorted({str(r["body_last"]) for r in res}),
                       slowest_command_ms=max(r["slowest_command_ms"] for r in res),
                       actions_equal_first_seen=all(r["actions_sha256"] == solo[r["idx"]] for r in res), per_process=res)
            rows.append(row)
            print({k: v for k, v in row.items() if k != "per_process"}, file=sys.stderr, flush=True)
    print(json.dumps(dict(what="fused one-launch planner body under GPU sharing: P planner processes (K, T below) on one MI355X, counted "
                               "from each process's FIRST command", K=a.K, T=a.T, commands=a.commands, host_spin=a.host_spin,
                          options=extra, rows=rows), indent=1))
