"""The N > 1 plumbing of bench.py (moved out of it in round 6; entry point and CLI are bench.py's, unchanged).

Nothing here touches the GPU or imports the library.  `bench.py --gpus N` without a launcher starts its own ranks
(`self_launch`); under a launcher every rank process is a SUPERVISOR (`supervise`) that runs the measurement in a fresh child
(`bench.py ... --worker`) under a progress watchdog (`run_watched`: markers written by `mark()`), agrees with the other ranks'
supervisors on the outcome through the launcher's TCP store (`SupervisorStore`), and falls back from the library-owned RCCL
communicator to torch.distributed's collective -- or measures both -- as bench.py's docstring describes.
"""

import json
import os
import signal
import subprocess
import sys
import tempfile
import time

# the script the supervisors start their children from: bench.py, beside this file
BENCH_PY = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench.py")


def free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


# ------------------------------------------------------------------------------------------------ watchdog plumbing
PROGRESS_ENV = "NLC_BENCH_PROGRESS_FILE"


def mark(phase):
    """Worker side of the watchdog: append a progress marker to the file the supervisor watches (no-op without one)."""
    path = os.environ.get(PROGRESS_ENV)
    if path:
        with open(path, "a") as f:
            f.write(f"{phase} {time.time():.3f}\n")


def kill_group(proc, grace_s=5.0):
    """End a child started with start_new_session=True together with everything it started: SIGTERM to the process
    group, SIGKILL after `grace_s`.  Only the exact group this process created is signalled."""
    if proc.poll() is not None:
        return
    try:
        os.killpg(proc.pid, signal.SIGTERM)
    except ProcessLookupError:
        return
    t0 = time.time()
    while proc.poll() is None and time.time() - t0 < grace_s:
        time.sleep(0.05)
    if proc.poll() is None:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        proc.wait()


def run_watched(cmd, env, init_s, step_s, total_s, peer_failed=None, stdout=subprocess.PIPE):
    """Run `cmd` as a child in its own process group under a progress watchdog.  The child appends markers to a file
    (mark()); it is killed when it has written none for `init_s` seconds at the start, no NEW one for `step_s` seconds
    afterwards, when `total_s` is exceeded, or as soon as `peer_failed()` says another rank's child is gone (its peers
    would only wait in a collective for a rank that will never come).  Returns (status, returncode, stdout bytes, last
    marker) with status 'ok' | 'failed' | 'timeout' | 'peer'."""
    with tempfile.TemporaryDirectory(prefix="nlcbench_") as tmp:
        prog = os.path.join(tmp, "progress")
        open(prog, "w").close()
        env = dict(env, **{PROGRESS_ENV: prog})
        out_path = os.path.join(tmp, "stdout")
        with open(out_path, "wb") as out_f:
            proc = subprocess.Popen(cmd, stdout=out_f if stdout == subprocess.PIPE else stdout, env=env, start_new_session=True)
            t0 = last_change = time.time()
            last_size, status = 0, None
            while proc.poll() is None:
                time.sleep(0.2)
                now = time.time()
                size = os.path.getsize(prog)
                if size != last_size:
                    last_size, last_change = size, now
                budget = init_s if last_size == 0 else step_s
                if now - last_change > budget or now - t0 > total_s:
                    status = "timeout"
                elif peer_failed is not None and peer_failed():
                    status = "peer"
                if status:
                    kill_group(proc)
                    break
            rc = proc.wait()
        data = open(out_path, "rb").read()
        lines = open(prog).read().split("\n")
        last = next((ln.split()[0] for ln in reversed(lines) if ln.strip()), None)
    if status is None:
        status = "ok" if rc == 0 else "failed"
    return status, rc, data, last


def last_json_line(data):
    lines = [ln for ln in data.decode(errors="replace").splitlines() if ln.strip().startswith("{")]
    return lines[-1] if lines else None


def watchdog_budgets(args):
    step_s = args.watchdog_step_s if args.watchdog_step_s > 0 else 90.0 + 0.25 * (args.steps + args.warmup)
    init_s = args.watchdog_init_s
    return init_s, step_s, init_s + 12 * step_s


def self_launch(n, argv, result_fd, args):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    as a child process (one rank per GPU; every rank is a supervisor, see supervise()), pass its stderr through, write its
    LAST stdout line -- rank 0's JSON line -- to the saved stdout, and return its exit code.  The child runs under a
    watchdog of its own (the supervisors' budgets for two attempts plus slack): on expiry its process group is killed
    and the exit code is non-zero -- this process never touched the GPU and never execs."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), BENCH_PY] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL between the ranks of one node)
    env.setdefault("OMP_NUM_THREADS", "8")
    init_s, step_s, total_s = watchdog_budgets(args)
    outer = 2 * total_s + 120.0
    # the launcher itself writes no progress markers: only the total budget applies
    status, rc, data, _ = run_watched(cmd, env, outer, outer, outer)
    line = last_json_line(data)
    if line:
        os.write(result_fd, (line + "\n").encode())
    if status == "timeout":
        sys.stderr.write(f"bench.py: the launched ranks did not finish within {outer:.0f} s; their process group was killed\n")
        return 124
    if not line and rc == 0:
        sys.stderr.write("bench.py: the launched ranks printed no JSON line\n")
        return 1
    return rc


class SupervisorStore:
    """The supervisors' own key space on the launcher's TCP store (torch.distributed.run hosts it at MASTER_ADDR:MASTER_PORT
    and sets TORCHELASTIC_USE_AGENT_STORE; under any other launcher rank 0's supervisor hosts it and the children are told to
    connect as clients).  CPU only: a supervisor never touches the GPU."""

    def __init__(self, rank, world, timeout_s):
        from datetime import timedelta

        import torch.distributed as dist

        addr, port = os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"])
        self.agent_hosts = os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True"
        self.tcp = dist.TCPStore(addr, port, None, (not self.agent_hosts) and rank == 0, timedelta(seconds=timeout_s),
                                 wait_for_workers=False, multi_tenant=True)
        self.store = dist.PrefixStore(f"nlcbench_sup/{os.environ.get('TORCHELASTIC_RUN_ID', 'run')}", self.tcp)
        self.rank, self.world = rank, world

    def set(self, key, value):
        self.store.set(key, str(value))

    def has(self, key):
        try:
            return bool(self.store.check([key]))
        except Exception:
            return False

    def get(self, key, timeout_s):
        from datetime import timedelta

        self.store.wait([key], timedelta(seconds=timeout_s))
        return self.store.get(key).decode()

    def gather(self, prefix, timeout_s):
        return [self.get(f"{prefix}/r{r}", timeout_s) for r in range(self.world)]


def supervise(args, argv, result_fd):
    """One rank of an N > 1 run under a launcher (RANK set): never touches the GPU.  Runs the measurement as a fresh child
    process (`--worker`) under the progress watchdog, agrees with the other ranks' supervisors on the outcome, and -- see the
    module docstring -- falls back from the library-owned collective to torch.distributed's, or measures both."""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    init_s, step_s, total_s = watchdog_budgets(args)
    sup = SupervisorStore(rank, world, total_s + 60.0)
    base_env = dict(os.environ)  # (the store at MASTER_PORT outlives every attempt: children join it as clients, init_pg)

    attempts_log = []

    def attempt(idx, mode, extra=()):
        env = dict(base_env)
        cmd = [sys.executable, BENCH_PY] + list(argv) + ["--worker", "--attempt", str(idx), "--collective", mode]
        cmd += list(extra)
        status, rc, data, last = run_watched(cmd, env, init_s, step_s, total_s, peer_failed=lambda: sup.has(f"a{idx}/fail"))
        if status != "ok":
            sup.set(f"a{idx}/fail", rank)  # peers: stop waiting in a collective for this rank
            sys.stderr.write(f"bench.py supervisor rank {rank}: attempt {idx} ({mode}) {status} (exit {rc}, last marker {last})\n")
        sup.set(f"a{idx}/r{rank}", json.dumps(dict(status=status, rc=rc, last=last)))
        results = [json.loads(v) for v in sup.gather(f"a{idx}", total_s + 60.0)]
        attempts_log.append((mode, results))
        ok = all(r["status"] == "ok" for r in results)
        line = last_json_line(data) if rank == 0 else None
        if rank == 0 and ok and line is None:
            ok = False
            results[0] = dict(status="failed", rc=rc, last="no JSON line")
        # rank 0 has the last word (it alone sees whether a line was printed)
        if rank == 0:
            sup.set(f"a{idx}/verdict", int(ok))
        ok = sup.get(f"a{idx}/verdict", 120.0) == "1"
        return ok, line, results

    def reason_of(results, mode):
        bad = [f"rank {r}: {v['status']} (exit {v['rc']}, last progress marker {v['last']})" for r, v in enumerate(results)
               if v["status"] != "ok"]
        return f"--collective {mode}: " + "; ".join(bad) if bad else f"--collective {mode}: rank 0 printed no result line"

    first = args.collective
    ok, line, results = attempt(0, first)
    final, fallback_reason, also = None, None, None
    if ok:
        final = json.loads(line) if rank == 0 else None
        # both modes in one record: when the library's collective produced the line, measure torch.distributed's as well
        want_also = False
        if rank == 0:
            native = bool((final.get("config") or {}).get("native_collective"))
            want_also = bool(native and world > 1 and first == "auto" and not args.no_also_collective and not args.dry_launch)
            sup.set("also", int(want_also))
        want_also = sup.get("also", 120.0) == "1"
        if want_also:
            ok2, line2, results2 = attempt(1, "torch", ("--no-ilt", "--no-cpu-baseline"))
            if rank == 0:
                if ok2:
                    l2 = json.loads(line2)
                    also = dict(collective=l2["config"].get("collective"), value=l2["value"], ms_per_step=l2["ms_per_step"],
                                kernels_avg_ms=l2.get("kernels_avg_ms"), collective_timing=l2["config"].get("collective_timing"),
                                ranks_seen=l2["config"].get("ranks_seen"))
                else:
                    also = dict(error=reason_of(results2, "torch"))
    elif first != "torch" and not args.dry_launch_no_fallback:
        fallback_reason = reason_of(results, first)
        ok, line, results = attempt(1, "torch")
        if ok and rank == 0:
            final = json.loads(line)
    if rank == 0 and final is not None:
        # what the supervisors did: one entry per attempt with every rank's outcome and last progress marker
        final.setdefault("config", {})["supervisor"] = dict(
            watchdog_s=dict(init=init_s, step=step_s, total=total_s), requested_collective=first,
            attempts=[dict(collective=m, ranks=r) for m, r in attempts_log])
        if fallback_reason:
            final.setdefault("config", {})["collective_fallback_reason"] = fallback_reason
        if also is not None:
            final["also_collective"] = also
        os.write(result_fd, (json.dumps(final) + "\n").encode())
    if not ok and rank == 0:
        sys.stderr.write("bench.py: no attempt produced a result: " + reason_of(results, "torch" if fallback_reason else first) + "\n")
    # nobody leaves before everybody has read the last keys (rank 0's supervisor may host the store)
    sup.set(f"done/r{rank}", 1)
    try:
        sup.gather("done", 60.0)
    except Exception:
        pass
    return 0 if ok else 1
