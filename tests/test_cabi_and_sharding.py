"""CPU: the C-ABI library loads and exports every symbol include/nlc.h declares; entry points fail loudly
without a GPU; and the world_size-2 sharded planner path (host logic + collective) with gloo."""

import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def lib():
    from neurallaplacecontrol_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "neurallaplacecontrol_amd", "csrc"), "-j", "8"])
    return _lib.load_library()


def test_header_symbols_exported(lib):
    from neurallaplacecontrol_amd import _lib

    hdr = open(os.path.join(REPO, "include", "nlc.h")).read()
    declared = sorted(set(re.findall(r"\b(nlc_[a-z_A-Z0-9]+)\s*\(", hdr)))
    assert declared == sorted(_lib.SYMBOLS), "include/nlc.h and _lib.SYMBOLS disagree"
    for name in declared:
        assert hasattr(lib, name), f"libnlc_hip.so does not export {name}"
    assert lib.nlc_abi_version() == int(re.search(r"#define\s+NLC_ABI_VERSION\s+(\d+)", hdr).group(1))


def test_struct_layouts_match_header():
    """ctypes mirrors must have the sizes the C compiler gives the header structs."""
    from neurallaplacecontrol_amd import _lib

    src = r"""
    #include <stdio.h>
    #include "nlc.h"
    int main(){ printf("%zu %zu %zu %zu %zu %zu\n", sizeof(nlc_ilt_desc), sizeof(nlc_model_desc), sizeof(nlc_mppi_desc),
                       sizeof(nlc_mppi_buffers), sizeof(nlc_rnn_desc), sizeof(nlc_node_desc)); return 0; }
    """
    import tempfile

    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), "-o", os.path.join(td, "s"), os.path.join(td, "s.c")])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(td, "s")]).split()]
    assert sizes == [C.sizeof(_lib.IltDesc), C.sizeof(_lib.ModelDesc), C.sizeof(_lib.MppiDesc), C.sizeof(_lib.MppiBuffers),
                     C.sizeof(_lib.RnnDesc), C.sizeof(_lib.NodeDesc)]


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_fails_loudly_without_gpu(lib):
    from neurallaplacecontrol_amd import _lib

    h = C.c_void_p()
    rc = lib.nlc_create(0, C.byref(h))
    assert rc != 0 and not h.value
    assert b"" != lib.nlc_last_error(None)
    import neurallaplacecontrol_amd as nlc

    with pytest.raises(RuntimeError):
        nlc.ilt_reconstruct(torch.zeros(1, 1, 17).double(), torch.zeros(1, 1, 17).double(), torch.ones(1).double())
    with pytest.raises(RuntimeError):
        nlc.MPPIDelay(nlc.OracleDynamics("oderl-pendulum"), nlc.EnvCost("oderl-pendulum"), 3, nlc.noise_sigma(1), 8, 4)
    with pytest.raises(_lib.NlcError):
        _lib.Ctx(0)
    with pytest.raises(RuntimeError):
        nlc.BatchedEnv("oderl-cartpole", 4)
    d = 3
    for model in (
        nlc.DeltaTRNN(d, 1, hidden_units=64, state_mean=np.zeros(d), state_std=np.ones(d), action_mean=np.array([0]),
                      action_std=np.array([1.0]), normalize=True, normalize_time=True).double(),
        nlc.NODE(d, 1, d, hidden_units=64, state_mean=np.zeros(d), state_std=np.ones(d), action_mean=np.array([0]),
                 action_std=np.array([1.0]), normalize=True, normalize_time=True, augment_dim=1).double(),
    ):
        with torch.no_grad(), pytest.raises(RuntimeError):  # no CPU fallback behind the model mirrors either
            model(torch.zeros(2, d).double(), torch.zeros(2, 4, 1).double(), torch.full((2, 1), 0.05).double())
        with pytest.raises(RuntimeError):  # ... nor in grad mode (training forward needs the GPU as well)
            model(torch.zeros(2, d).double(), torch.zeros(2, 4, 1).double(), torch.full((2, 1), 0.05).double())


def test_product_does_not_import_oracle():
    """The product path must not route through the CPU oracle (it is test infrastructure)."""
    pkg = os.path.join(REPO, "neurallaplacecontrol_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{f} imports the oracle"


def test_bench_weights_equal_oracle_synthetic_weights():
    """bench.py's product-side synthetic model == oracle.make_synthetic_state_dict(tame=True) == reference ctor + shift."""
    sys.path.insert(0, REPO)
    import bench
    from oracle import nl_model as onl

    model = bench.synthetic_state_dict(5, 1, 17)
    st = onl.ENV_STATS["oderl-cartpole"]
    ref = onl.make_synthetic_state_dict(0, 5, 1, 128, 17, st["state_std"], [1.5], tame=True)
    sd = model.state_dict()
    for k, v in ref.items():
        assert torch.equal(sd[k].to(torch.float64), v), k
    # every --config's model: the other envs, and the de Hoog taming of configs[4]
    for key, cfg in bench.CONFIGS.items():
        d, nu, A, std = bench.ENV_SHAPES[cfg["env"]]
        model = bench.synthetic_state_dict(d, nu, cfg["S"], env=cfg["env"], algo=cfg["algo"])
        ref = onl.make_synthetic_state_dict(0, d, nu, 128, cfg["S"], std, [A / 2.0], tame="dehoog" if cfg["algo"] == "dehoog" else True)
        sd = model.state_dict()
        for k, v in ref.items():
            assert torch.equal(sd[k].to(torch.float64), v), (key, k)
        assert onl.ENV_STATS[cfg["env"]]["state_std"] == std and onl.ENV_STATS[cfg["env"]]["act_high"] == A


WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from neurallaplacecontrol_amd.sharding import (shard_range, gather_partials, slice_noise, partial_width, merge_partials_torch,
                                               replicate_from_rank0, check_same_on_all_ranks, share_bytes_from_rank0)
from oracle import envs as oenvs, mppi as omppi
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
env, K, T, A, nu, nx = "oderl-pendulum", 64, 6, 2.0, 1, 3
# ranks that drew DIFFERENT control sequences (badly seeded hosts) are brought to rank 0's; mismatching seeds raise
torch.manual_seed(1000 + rank)
U_own = torch.randn(T, nu, dtype=torch.float64) * 0.3
U = replicate_from_rank0(U_own, dist.group.WORLD)
assert (rank == 0) == bool(torch.equal(U, U_own))
check_same_on_all_ranks((K, T, 7), dist.group.WORLD, "sizes / seed")
# the unique id of the library's own communicator travels from rank 0 through the group the caller has (nlc_comm_init)
uid = bytes(range(128)) if rank == 0 else None
assert share_bytes_from_rank0(uid, 128, dist.group.WORLD) == bytes(range(128))
try:
    check_same_on_all_ranks((K, T, 7 + rank), dist.group.WORLD, "sizes / seed")
    raise SystemExit("seed mismatch not detected")
except ValueError:
    pass
torch.manual_seed(123)                     # same seed on every rank -> same global draw
torch.randn(T, nu, dtype=torch.float64)    # (keeps the draw order of the unsharded reference run below: U, then the noise)
raw = torch.randn(K, T, nu, dtype=torch.float64)
state, ab = oenvs.initial_state(env), torch.zeros(4, nu, dtype=torch.float64)
sig_inv = torch.ones(1, 1, dtype=torch.float64)
k0, kl = shard_range(K, world, rank)
ts = torch.full((kl, 1), 0.05, dtype=torch.float64)
dyn = lambda s, w: oenvs.ORACLE_DYNAMICS[env](s, w, ts, 1)
# per-shard phase 1 (what nlc_mppi_rollout does on each GPU), with the oracle as the arithmetic
out = omppi.mppi_command(U.clone(), state, ab, slice_noise(raw, k0, kl).clone(), dyn, oenvs.RUNNING_COST[env], nx,
                         sig_inv, 1.0, A, torch.tensor(-A), torch.tensor(A))
partials = omppi.shard_partials(out["cost_total"], out["noise"])
assert partials.numel() == partial_width(T, nu)
gathered = gather_partials(partials, torch.empty(world, partials.numel(), dtype=torch.float64), dist.group.WORLD)
beta, eta, dU = omppi.merge_partials(gathered)
# the product's tensor-op merge (planners the HIP kernels are not built for: MPPIDelay._torch_command) against the oracle's
b2, e2, S2 = merge_partials_torch(gathered, 1.0)
assert float(b2) == float(beta) and abs(float(e2) - float(eta)) <= 1e-14 * float(eta)
assert torch.allclose(S2 / e2, dU, rtol=1e-13, atol=1e-15)
Ush = torch.roll(U, -1, 0); Ush[-1] = 0
U_new = Ush + dU.view(T, nu)
torch.save(dict(U=U_new, beta=beta, eta=eta), os.path.join(sys.argv[2], f"r{rank}.pt"))
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world", [2, 8])
def test_multi_rank_gloo_sharded_command(tmp_path, world):
    """world_size 2 and 8 (the driver's scaling run) over gloo: per-shard rollout + all-gather of (beta, eta, S) + merge ==
    unsharded command."""
    from oracle import envs as oenvs
    from oracle import mppi as omppi

    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = str(29531 + world)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1")
    subprocess.check_call(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
         "--master-port", port, str(script), REPO, str(tmp_path)],
        env=env, timeout=600,
    )
    r0 = torch.load(tmp_path / "r0.pt")
    for r in range(1, world):
        assert torch.equal(r0["U"], torch.load(tmp_path / f"r{r}.pt")["U"]), "ranks disagree after the merge"
    # unsharded reference
    envn, K, T, A, nu, nx = "oderl-pendulum", 64, 6, 2.0, 1, 3
    torch.manual_seed(1000)  # the workers end up with RANK 0's control sequence (replicate_from_rank0)
    U = torch.randn(T, nu, dtype=torch.float64) * 0.3
    torch.manual_seed(123)
    torch.randn(T, nu, dtype=torch.float64)
    raw = torch.randn(K, T, nu, dtype=torch.float64)
    ts = torch.full((K, 1), 0.05, dtype=torch.float64)
    full = omppi.mppi_command(U.clone(), oenvs.initial_state(envn), torch.zeros(4, nu, dtype=torch.float64), raw,
                              lambda s, w: oenvs.ORACLE_DYNAMICS[envn](s, w, ts, 1), oenvs.RUNNING_COST[envn], nx,
                              torch.ones(1, 1, dtype=torch.float64), 1.0, A, torch.tensor(-A), torch.tensor(A))
    np.testing.assert_allclose(r0["U"].numpy(), full["U"].numpy(), rtol=1e-12, atol=1e-14)
    assert float(r0["beta"]) == float(full["beta"])
    np.testing.assert_allclose(float(r0["eta"]), float(full["eta"]), rtol=1e-12)


PREHEAT_WORKER = r"""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import bench
dist.init_process_group("gloo")
rank = dist.get_rank()
calls = []
def one_step():
    # every step holds a collective, as a sharded command() does; rank 1 is the slower host
    t = torch.ones(1, dtype=torch.float64)
    dist.all_reduce(t)
    calls.append(float(t))
    time.sleep(0.0005 * (1 + 3 * rank))
n = bench.preheat(one_step, 60.0, dist.group.WORLD, "cpu", chunk=4)
assert n == len(calls) and n >= 4 and n % 4 == 0
assert bench.preheat(one_step, 0.0, dist.group.WORLD, "cpu") == 0
counts = [None] * dist.get_world_size()
dist.all_gather_object(counts, n)
assert len(set(counts)) == 1, counts     # the same number of steps (= collectives) on every rank
dist.barrier()
dist.destroy_process_group()
"""


def test_bench_preheat_takes_the_same_number_of_steps_on_every_rank(tmp_path):
    """bench.py's untimed pre-heat is bounded by wall time, and a sharded command() contains a collective: ranks stopping on
    their own clocks would leave unmatched collectives behind.  Two gloo ranks with different step durations must agree on
    the count."""
    script = tmp_path / "preheat_worker.py"
    script.write_text(PREHEAT_WORKER)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.check_call(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", "29561", str(script), REPO],
        env=env, timeout=600,
    )


def test_shard_range():
    from neurallaplacecontrol_amd.sharding import shard_range

    assert [shard_range(16384, 8, r) for r in (0, 7)] == [(0, 2048), (14336, 2048)]
    with pytest.raises(ValueError):
        shard_range(10, 4, 0)


def test_bench_bare_multi_gpu_invocation_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher (the shape of the driver's N = 1 command) must start its ranks itself:
    --dry-launch goes as far as every rank reporting the environment torch.distributed.run gave it, without a GPU call,
    and the parent relays exactly one JSON line."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-launch"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["dry_launch"] is True and rec["n_gpus"] == 2
    assert [(r["RANK"], r["LOCAL_RANK"], r["WORLD_SIZE"]) for r in rec["ranks"]] == [("0", "0", "2"), ("1", "1", "2")]
    assert all(r["MASTER_ADDR"] == "127.0.0.1" for r in rec["ranks"])
    # the driver's own form (ranks already launched) must not launch again: one rank under torch.distributed.run
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
                          "127.0.0.1", "--master-port", "29577", os.path.join(REPO, "bench.py"), "--gpus", "1", "--dry-launch"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    rec = json.loads([ln for ln in out.stdout.decode().splitlines() if ln.strip().startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["ranks"][0]["RANK"] == "0"


def _bare_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}


def test_bench_watchdog_ends_a_run_whose_rank_never_arrives():
    """a rank that sleeps forever (in every attempt) must not hang `bench.py --gpus N`: each rank's
    supervisor kills its measuring child when the progress markers stop, the fallback attempt ends the same way, and the
    parent exits non-zero well within its own watchdog -- no JSON line, no process left behind."""
    import time

    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-launch", "--test-hang-rank", "1",
                          "--watchdog-step-s", "4", "--watchdog-init-s", "120"], env=_bare_env(), stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=600)
    took = time.time() - t0
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.decode().splitlines() if ln.strip()], out.stdout
    err = out.stderr.decode()
    assert "attempt 0 (auto) timeout" in err and "attempt 1 (torch) timeout" in err, err[-3000:]
    assert took < 120, took  # two attempts of ~4 s of silence each + process start-up, nowhere near the 600 s above


def test_bench_falls_back_to_the_torch_collective_when_the_first_attempt_hangs():
    """... and a hang in the FIRST attempt only (the shape of a library-owned RCCL all-gather misbehaving at G > 1, the path no
    hardware run has exercised) costs the watchdog's patience, not the record: every rank starts a fresh child with
    --collective torch and the one JSON line says why."""
    import json

    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-launch", "--test-hang-rank", "1",
                          "--test-hang-attempts", "1", "--watchdog-step-s", "4", "--watchdog-init-s", "120"], env=_bare_env(),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["attempt"] == 1 and rec["config"]["collective"] == "torch"
    why = rec["config"]["collective_fallback_reason"]
    assert "--collective auto" in why and "rank 1: timeout" in why and "last progress marker pg_ready" in why
    assert [r["RANK"] for r in rec["ranks"]] == ["0", "1"]


def test_bench_run_watched_reports_a_failing_child_and_kills_a_silent_one(tmp_path):
    sys.path.insert(0, REPO)
    import bench

    status, rc, data, last = bench.run_watched([sys.executable, "-c", "import sys; print('{\"a\": 1}'); sys.exit(3)"], dict(os.environ),
                                               30, 30, 60)
    assert (status, rc, last) == ("failed", 3, None) and bench.last_json_line(data) == '{"a": 1}'
    code = ("import os, time\n"
            f"open(os.environ[{bench.PROGRESS_ENV!r}], 'a').write('phase_one 0\\n')\n"
            "time.sleep(600)\n")
    status, rc, data, last = bench.run_watched([sys.executable, "-c", code], dict(os.environ), 30, 1.5, 60)
    assert status == "timeout" and rc != 0 and last == "phase_one"
    # a peer's failure ends the wait at once
    status, rc, data, last = bench.run_watched([sys.executable, "-c", "import time; time.sleep(600)"], dict(os.environ), 30, 30, 60,
                                               peer_failed=lambda: True)
    assert status == "peer"


def test_graft_entry_build_runs(lib):
    """The driver's build check: build() must succeed on a machine without a GPU (make is a no-op when up to date)."""
    sys.path.insert(0, REPO)
    import __graft_entry__ as g

    g.build()


def test_model_from_reference_copies_hyperparameters_buffers_and_weights():
    """NeuralLaplaceModel.from_reference on a reference-shaped module (sub-module names / attributes of w_nl.py:66-115;
    built here with this package's own class, whose layout is the reference's): nothing touches the GPU."""
    import neurallaplacecontrol_amd as nlc

    torch.manual_seed(3)
    ref = nlc.NeuralLaplaceModel(
        5, 1, 5, hidden_units=128, s_recon_terms=17, ilt_algorithm="fourier", encode_obs_time=True,
        state_mean=np.zeros(5), state_std=np.arange(1.0, 6.0), action_mean=np.array([0]), action_std=np.array([1.5]),
        normalize=True, normalize_time=True,
    ).double()
    twin = nlc.NeuralLaplaceModel.from_reference(ref)
    assert twin is not ref and twin.encode_obs_time and twin.action_dim == 1 and twin.hidden_units == 128
    assert twin.s_recon_terms == 17 and twin.normalize and twin.normalize_time and twin.ilt_algorithm == "fourier"
    sd_a, sd_b = ref.state_dict(), twin.state_dict()
    assert list(sd_a) == list(sd_b)
    for k in sd_a:
        assert sd_a[k].dtype == sd_b[k].dtype and torch.equal(sd_a[k], sd_b[k]), k
    assert twin.action_mean.dtype == torch.int64 and twin.dt.dtype == torch.float64  # reference dtypes after .double()
    assert float(twin.dt) == float(np.float32(0.05))


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_plain_c_client_compiles_and_fails_loudly_without_gpu(lib, tmp_path):
    """tests/helpers/cabi_client.c is C99 against include/nlc.h (gcc, no Python): it must build here and report the
    missing device through nlc_last_error (exit code 3), not crash."""
    libdir = os.path.join(REPO, "neurallaplacecontrol_amd")
    # cabi_client: single planner, env step, ILT; cabi_sharded_client: two ctxs walking the sharded protocol (NLC_AGAIN loop)
    for name, code in (("cabi_client", 3), ("cabi_sharded_client", 7)):
        exe = str(tmp_path / name)
        subprocess.check_call(
            ["gcc", "-std=c99", "-Wall", "-Werror", os.path.join(REPO, "tests", "helpers", name + ".c"), "-I", os.path.join(REPO, "include"),
             "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-L" + libdir, "-lnlc_hip", "-L/opt/rocm/lib", "-lamdhip64",
             "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", exe])
        r = subprocess.run([exe], capture_output=True, timeout=120)
        assert r.returncode == code and b"nlc_create" in r.stderr, (name, r.returncode, r.stderr)
