"""`host_spin` under contention.  The reference harness fans its evaluations out over a
multiprocessing.Pool(12) (run_exp_multi.py:145), one planner per worker, all on one GPU.  P planner processes share cuda:0
here (P <= 6: the GPU pool's process guard allows six processes on a card) and each runs the harness's control loop
(command -> roll the action buffer) for a fixed wall time, with host_spin = 0 (hipStreamSynchronize), 1 (spin on the pinned
sequence word until it changes) and 2 (sleep through the predicted wait -- the shortest of the last eight -- then spin).
Every row also says which rollout body the processes ran on and whether a fused launch ever gave up (nlc_get_stat), counted
from each process's FIRST command (the 30 warm-up commands are included in those counters): the spin comparison is not
confounded by a silent fall-back to the two-launch body (tools/fused_sharing.py measures that question itself).
Reported per setting: aggregate planning steps/s, the host CPU seconds the workers burned per wall second (user + system,
from os.times) and -- with `--cpus C` -- the same with the workers confined to C cores (sched_setaffinity), i.e. the
oversubscribed host a Pool(12) on a small node is.

    python tools/spin_contention.py [--procs 1,3,6] [--K 1000] [--T 40] [--seconds 4] [--cpus 0]"""
import argparse, json, os, sys, time
import multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def worker(idx, P, K, T, opts, seconds, cpus, barrier, q):
    if cpus:
        os.sched_setaffinity(0, set(range(cpus)))
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
    for _ in range(30):
        a = p.command(state, ab)
    torch.cuda.synchronize()
    barrier.wait()
    c0, t0, n = os.times(), time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        a = p.command(state, ab)
        ab = torch.roll(ab, -1, 0); ab[-1] = a
        n += 1
    wall = time.perf_counter() - t0
    c1 = os.times()
    q.put(dict(idx=idx, steps=n, wall=wall, cpu=(c1.user - c0.user) + (c1.system - c0.system), body=p.rollout_body,
               fused_timeouts=p.fused_timeouts, fused_fallbacks=p.fused_fallbacks))
    barrier.wait()


def run(P, K, T, opts, seconds, cpus):
    ctx = mp.get_context("spawn")
    barrier, q = ctx.Barrier(P), ctx.Queue()
    procs = [ctx.Process(target=worker, args=(i, P, K, T, opts, seconds, cpus, barrier, q)) for i in range(P)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=300) for _ in range(P)]
    for pr in procs:
        pr.join(timeout=60)
    agg = sum(r["steps"] / r["wall"] for r in res)
    cpu = sum(r["cpu"] / r["wall"] for r in res)
    return dict(procs=P, K=K, T=T, options=opts, cpus=cpus or None, steps_per_s=round(agg, 1),
                ms_per_step_per_proc=round(1e3 * P / agg, 4), host_cpu_cores_busy=round(cpu, 2))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", default="1,3,6")
    ap.add_argument("--K", type=int, default=1000)  # config.py: mppi_roll_outs default
    ap.add_argument("--T", type=int, default=40)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--cpus", default="0,2")
    ap.add_argument("--modes", default="0,1,2")
    a = ap.parse_args()
    out = []
    for cpus in [int(c) for c in a.cpus.split(",")]:
        for P in [int(x) for x in a.procs.split(",")]:
            for mode in a.modes.split(","):
                opts = {} if mode == "auto" else {"host_spin": int(mode)}
                r = run(P, a.K, a.T, opts, a.seconds, cpus)
                r["mode"] = mode
                out.append(r)
                print(r, file=sys.stderr, flush=True)
    print(json.dumps(dict(what="host_spin under contention: P planner processes on one GPU", usable_cpus=len(os.sched_getaffinity(0)), rows=out)))
