"""GPU parity tests (run with ``-m gpu`` on an MI355X), through the C ABI of libnlc_hip.so via the drop-in Python mirror:
batched episodes (f2), K-sharding over ranks (e): shard merge, two processes on one GPU, the library RCCL communicator, the bench rehearsal.  Helpers and tolerances: tests/gpu_common.py.
"""

import glob
import os

import numpy as np
import pytest
import torch

from gpu_common import *  # noqa: F401,F403
from gpu_common import GOLD, TOL, T64, load_sd, build_model

pytestmark = pytest.mark.gpu


def test_mppi_two_shards_merge_equals_single(nlc):
    """SURVEY §8e on one GPU: two K/2 planners' partials merged through nlc_mppi_finish == one K planner."""
    import ctypes as C

    from neurallaplacecontrol_amd import _lib

    env, K, T, A = "oderl-cartpole", 256, 7, 3.0
    sig = nlc.noise_sigma(1)
    torch.manual_seed(0)
    raw = torch.randn(K, T, 1, dtype=torch.float64)
    U0 = torch.randn(T, 1, dtype=torch.float64) * 0.2
    st, ab = nlc.initial_state(env), torch.randn(4, 1, dtype=torch.float64)

    def planner(**kw):
        return nlc.MPPIDelay(nlc.OracleDynamics(env, 0.05, 1), nlc.EnvCost(env), 5, sig, K, T, "cpu", lambda_=1.0,
                             u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, U_init=U0.clone(), **kw)

    full = planner()
    full.noise_dist = type("R", (), {"sample": staticmethod(lambda shape: raw)})()
    a_full = full.command(st, ab)
    shards = []
    for r in range(2):
        p = planner()
        p.K_local, p.k_offset = K // 2, r * (K // 2)
        p.G, p.rank = 1, 0  # run phase 1 stand-alone; merge by hand below
        p.noise_dist = type("R", (), {"sample": staticmethod(lambda shape: raw)})()
        p.command(st, ab)  # fills partials (and applies a local-only update we overwrite next)
        shards.append(p)
    gathered = torch.stack([s._partials for s in shards]).contiguous()
    for r, p in enumerate(shards):
        p.U = torch.roll(U0, -1, 0).index_fill(0, torch.tensor([T - 1]), 0.0)  # U after the shift, before update
        act = torch.empty(1, dtype=torch.float64)
        p.ctx.check(p.ctx.lib.nlc_mppi_finish(p.ctx.h, _lib.ptr(gathered), 2, r, C.byref(p._buf), _lib.ptr(act)))
        np.testing.assert_allclose(act.numpy(), a_full.numpy(), **TOL)
        np.testing.assert_allclose(p.U.numpy(), full.U.numpy(), **TOL)
        np.testing.assert_allclose(p.omega.numpy(), full.omega[r * (K // 2) : (r + 1) * (K // 2)].numpy(), **TOL)


def _spawn_worker(q):
    import torch as _t

    import neurallaplacecontrol_amd as n

    m = n.MPPIDelay(n.OracleDynamics("oderl-pendulum", 0.05, 0), n.EnvCost("oderl-pendulum"), 3, n.noise_sigma(1), 128, 5,
                    "cpu", u_scale=2.0, U_init=_t.zeros(5, 1, dtype=_t.float64), noise_rng="philox", seed=3)
    q.put(m.command(n.initial_state("oderl-pendulum"), _t.zeros(4, 1, dtype=_t.float64)).tolist())


def test_spawned_worker_creates_its_own_ctx(nlc):
    """The harness fans out with multiprocessing 'spawn' (run_exp_multi.py:145,207): HIP is initialised lazily in
    the worker, and the Philox stream makes the result identical to the parent's."""
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_spawn_worker, args=(q,))
    p.start()
    child = q.get(timeout=180)
    p.join(timeout=60)
    assert p.exitcode == 0
    m = nlc.MPPIDelay(nlc.OracleDynamics("oderl-pendulum", 0.05, 0), nlc.EnvCost("oderl-pendulum"), 3, nlc.noise_sigma(1),
                      128, 5, "cpu", u_scale=2.0, U_init=torch.zeros(5, 1, dtype=torch.float64), noise_rng="philox", seed=3)
    mine = m.command(nlc.initial_state("oderl-pendulum"), torch.zeros(4, 1, dtype=torch.float64)).tolist()
    assert child == mine


@pytest.mark.parametrize("env,delay", [("oderl-cartpole", 2), ("oderl-pendulum", 0), ("oderl-acrobot", 3)])
def test_batched_planner_oracle_dynamics_equals_single_planners(nlc, env, delay):
    """Collector shape (K = 1000 is ragged against every block size): bit-identical to E single planners."""
    bat = _batched_vs_singles(nlc, lambda: nlc.OracleDynamics(env, 0.05, delay), env, E=5, K=1000, T=12)
    assert bat.U.shape[0] == 5 and bat.noise.shape[:2] == (5, 1000)


def test_batched_planner_oracle_options_and_per_sample_state(nlc):
    _batched_vs_singles(nlc, lambda: nlc.OracleDynamics("oderl-acrobot", 0.05, 1), "oderl-acrobot", E=3, K=70, T=5,
                        per_sample=True, sample_null_action=True, noise_abs_cost=True)


@pytest.mark.parametrize("algo,S,K", [("fourier", 17, 100), ("fourier", 17, 8200), ("dehoog", 17, 72), ("fixed_tablot", 17, 72),
                                      ("stehfest", 8, 100)])
@pytest.mark.fp64_bit_identity
def test_batched_planner_nl_dynamics_equals_single_planners(nlc, algo, S, K):
    """NL dynamics: K = 100 makes the 16-sample MFMA tiles straddle episodes; 8200 takes the wave-per-tile kernel."""
    from oracle import nl_model as onl

    env, d, nu, A = "oderl-cartpole", 5, 1, 3.0
    st = onl.ENV_STATS[env]
    sd = onl.make_synthetic_state_dict(4, d, nu, 128, S, st["state_std"], [A / 2], tame=True)
    model = build_model(nlc, sd, S=S, algo=algo)
    E = 3 if K < 1000 else 2
    _batched_vs_singles(nlc, lambda: nlc.NLDynamics(model, 0.05), env, E=E, K=K, T=6, n_cmd=2)


def test_batched_planner_philox_streams_and_device_inputs(nlc):
    """Device RNG: episode 0 continues the single planner's stream, other episodes draw different noise; states and
    action buffers handed over as device tensors give the same result as host tensors; reset(env_ids) is per episode."""
    from neurallaplacecontrol_amd.planners.mppi_batch import BatchedMPPIDelay

    env, E, K, T, A = "oderl-cartpole", 4, 512, 10, 3.0
    sig = nlc.noise_sigma(1)
    kw = dict(lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox", seed=11)
    U0 = torch.zeros(E, T, 1, dtype=torch.float64)
    states = torch.stack([_state(nlc, env, e) for e in range(E)])
    ab = torch.zeros(E, 4, 1, dtype=torch.float64)
    mk = lambda dev: BatchedMPPIDelay(nlc.OracleDynamics(env, 0.05, 2), nlc.EnvCost(env), 5, sig, E, K, T, dev,  # noqa: E731
                                      U_init=U0.clone(), **kw)
    host, devp = mk("cpu"), mk("cuda")
    a_h = host.command(states, ab)
    a_d = devp.command(states.cuda(), ab.cuda())
    assert a_d.is_cuda and torch.equal(a_h, a_d.cpu())
    single = nlc.MPPIDelay(nlc.OracleDynamics(env, 0.05, 2), nlc.EnvCost(env), 5, sig, K, T, "cpu", U_init=U0[0].clone(), **kw)
    a_s = single.command(states[0], ab[0])
    assert torch.equal(a_s, a_h[0]) and torch.equal(single.noise, host.noise[0])
    n = host.noise
    assert not torch.equal(n[0], n[1]) and not torch.equal(n[1], n[2])
    # U = 0 and bounds +-1 in normalised units: the bounded noise is N(0,1) clipped to [-1, 1] (std 0.718)
    assert abs(float(n.mean())) < 0.02 and abs(float(n.std()) - 0.718) < 0.02 and float(n.abs().max()) <= 1.0
    U_before = host.U.clone()
    torch.manual_seed(5)
    host.reset([1, 3])
    U_after = host.U
    assert torch.equal(U_after[0], U_before[0]) and torch.equal(U_after[2], U_before[2])
    assert not torch.equal(U_after[1], U_before[1]) and not torch.equal(U_after[3], U_before[3])
    host.reset()
    assert host.U.shape == (E, T, 1)


def test_batched_planner_rejects_unsupported(nlc):
    from neurallaplacecontrol_amd.planners.mppi_batch import BatchedMPPIDelay

    sig = nlc.noise_sigma(1)
    with pytest.raises(NotImplementedError):
        BatchedMPPIDelay(lambda s, a: s, lambda s, a: s.sum(1), 5, sig, 4, 64, 5, "cpu")
    b = BatchedMPPIDelay(nlc.OracleDynamics("oderl-cartpole", 0.05, 0), nlc.EnvCost("oderl-cartpole"), 5, sig, 4, 64, 5, "cpu")
    with pytest.raises(ValueError):
        b.command(torch.zeros(5, dtype=torch.float64), torch.zeros(4, 4, 1, dtype=torch.float64))
    with pytest.raises(ValueError):
        b.command(torch.zeros(4, 5, dtype=torch.float64), torch.zeros(4, 1, dtype=torch.float64))


def test_planners_sharing_a_model_are_independent_and_track_weight_updates(nlc):
    """Each planner owns its ctx (U, folded bias): interleaving two planners over ONE model changes nothing, and a
    planner picks up new weights (load_state_dict) on its next command, carrying its U over."""
    from oracle import nl_model as onl

    env, d, nu, A, K, T = "oderl-cartpole", 5, 1, 3.0, 64, 5
    st = onl.ENV_STATS[env]
    sd1 = onl.make_synthetic_state_dict(5, d, nu, 128, 17, st["state_std"], [A / 2], tame=True)
    sd2 = onl.make_synthetic_state_dict(6, d, nu, 128, 17, st["state_std"], [A / 2], tame=True)
    g = torch.Generator().manual_seed(9)
    raws = [torch.randn(K, T, nu, dtype=torch.float64, generator=g) for _ in range(3)]
    Ua, Ub = torch.randn(T, nu, dtype=torch.float64, generator=g) * 0.2, torch.randn(T, nu, dtype=torch.float64, generator=g)
    state, ab = _state(nlc, env, 1), torch.zeros(4, nu, dtype=torch.float64)
    kw = dict(lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A)

    def planner(model, U0, draws):
        p = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, nlc.noise_sigma(nu), K, T, "cpu",
                          U_init=U0.clone(), **kw)
        p.noise_dist = _Replay(*[r.clone() for r in draws])
        return p

    with torch.no_grad():
        alone = planner(build_model(nlc, sd1), Ua, raws)
        a1, a2 = alone.command(state, ab).clone(), alone.command(state, ab).clone()
        U_after2 = alone.U.clone()
        shared = build_model(nlc, sd1)
        pa, pb = planner(shared, Ua, raws), planner(shared, Ub, raws)
        assert pa.ctx is not pb.ctx
        b1 = pa.command(state, ab)
        pb.command(state, ab)
        b2 = pa.command(state, ab)
        assert torch.equal(a1, b1) and torch.equal(a2, b2) and torch.equal(pa.U, U_after2)
        shared.load_state_dict(sd2)  # new weights, same module
        b3 = pa.command(state, ab)
        fresh = planner(build_model(nlc, sd2), U_after2, raws[2:])
        assert torch.equal(b3, fresh.command(state, ab))
        assert not torch.equal(b3, alone.command(state, ab))


_TWO_RANK_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import neurallaplacecontrol_amd as n
dist.init_process_group("gloo")            # both ranks share cuda:0 here; bench.py uses "nccl" (= RCCL), one GPU per rank
rank = dist.get_rank()
sd = torch.load(os.path.join(sys.argv[2], "sd.pt"))
d, nu, A, K, T = 5, 1, 3.0, 1024, 8
import numpy as np
model = n.NeuralLaplaceModel(d, nu, d, hidden_units=128, s_recon_terms=17, ilt_algorithm="fourier", state_mean=np.zeros(d),
                             state_std=np.ones(d), action_mean=np.array([0]), action_std=np.array([1.0]),
                             normalize=True, normalize_time=True).double()
model.load_state_dict(sd)
model = model.cuda()
kw = dict(state_mean=np.zeros(d), state_std=np.ones(d), action_mean=np.array([0]), action_std=np.array([1.0]),
          normalize=True, normalize_time=True)
rnn = n.DeltaTRNN(d, nu, hidden_units=64, **kw).double()
rnn.load_state_dict(torch.load(os.path.join(sys.argv[2], "sd_rnn.pt")))
node = n.NODE(d, nu, d, hidden_units=64, augment_dim=1, **kw).double()
node.load_state_dict(torch.load(os.path.join(sys.argv[2], "sd_node.pt")))
out = {}
for name, dyn in (("nl", n.NLDynamics(model, 0.05)), ("oracle", n.OracleDynamics("oderl-cartpole", 0.05, 2)),
                  ("dtrnn", n.NLDynamics(rnn.cuda(), 0.05)), ("node", n.NLDynamics(node.cuda(), 0.05))):
    p = n.MPPIDelay(dyn, n.EnvCost("oderl-cartpole"), d, n.noise_sigma(nu), K, T, "cuda", lambda_=1.0,
                    u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, U_init=torch.zeros(T, nu, dtype=torch.float64),
                    noise_rng="philox", seed=21, process_group=dist.group.WORLD)
    assert p.K_local == K // 2 and p.k_offset == rank * (K // 2)
    state, ab = torch.tensor([0.01, 0.0, -1.0, 0.02, 0.0], dtype=torch.float64), torch.zeros(4, nu, dtype=torch.float64)
    with torch.no_grad():
        acts = [p.command(state, ab).cpu() for _ in range(3)]
    out[name] = dict(acts=torch.stack(acts), U=p.U.cpu(), noise=p.noise.cpu(), omega=p.omega.cpu())
# rollout_samples > 1 under the group: the variance term is a statistic of the WHOLE population (two small all-reduces)
p = n.MPPIDelay(n.OracleDynamics("oderl-cartpole", 0.05, 1), n.EnvCost("oderl-cartpole"), d, n.noise_sigma(nu), K, T, "cuda",
                lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, U_init=torch.zeros(T, nu, dtype=torch.float64),
                noise_rng="philox", seed=22, process_group=dist.group.WORLD, rollout_samples=3, rollout_var_cost=0.7,
                rollout_var_discount=0.9)
with torch.no_grad():
    a = p.command(state, ab).cpu()
out["varcost"] = dict(acts=a, cost=p.cost_total.cpu())
# a planner the HIP kernels are not built for (nu = 3): tensor ops on the GPU, the same (beta, eta, S) partials and ONE all-gather
import warnings
gen = torch.Generator().manual_seed(5)
Wx3, Wu3 = torch.randn(4, 4, generator=gen, dtype=torch.float64) * 0.3, torch.randn(3, 4, generator=gen, dtype=torch.float64) * 0.5
dyn3 = lambda s, w: s + 0.05 * (torch.tanh(s @ Wx3.to(s.device)) + (0.7 * w[:, -1, :] + 0.3 * w[:, 0, :]) @ Wu3.to(s.device))
cost3 = lambda s, u: (s ** 2).sum(dim=1) + 0.01 * (u ** 2).sum(dim=1)
torch.manual_seed(77)                      # the same global draw on every rank; each keeps its slice
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    p = n.MPPIDelay(dyn3, cost3, 4, n.noise_sigma(3), 512, 6, "cpu", lambda_=1.0, u_min=torch.tensor(-2.0), u_max=torch.tensor(2.0),
                    u_scale=2.0, U_init=torch.zeros(6, 3, dtype=torch.float64), process_group=dist.group.WORLD)
assert p.torch_path and p.K_local == 256
st3, ab3 = torch.linspace(-0.5, 0.5, 4, dtype=torch.float64), torch.zeros(4, 3, dtype=torch.float64)
acts3 = torch.stack([p.command(st3, ab3).cpu() for _ in range(3)])
out["torch_path"] = dict(acts=acts3, U=p.U.cpu(), omega=p.omega.cpu())
torch.save(out, os.path.join(sys.argv[2], f"r{rank}.pt"))
dist.destroy_process_group()
"""


_NATIVE_COLLECTIVE_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import neurallaplacecontrol_amd as n
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))   # RCCL, one rank (one GPU on this box)
sd = torch.load(os.path.join(sys.argv[2], "sd.pt"))
d, nu, A, K, T = 5, 1, 3.0, 2048, 12
import numpy as np
model = n.NeuralLaplaceModel(d, nu, d, hidden_units=128, s_recon_terms=17, ilt_algorithm="fourier", state_mean=np.zeros(d),
                             state_std=np.ones(d), action_mean=np.array([0]), action_std=np.array([1.0]),
                             normalize=True, normalize_time=True).double()
model.load_state_dict(sd)
model = model.cuda()
def planner(pg, native):
    return n.MPPIDelay(n.NLDynamics(model, 0.05), n.EnvCost("oderl-cartpole"), d, n.noise_sigma(nu), K, T, "cuda", lambda_=1.0,
                       u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, U_init=torch.zeros(T, nu, dtype=torch.float64),
                       noise_rng="philox", seed=5, process_group=pg, planner_options={"native_collective": native})
ps = [planner(None, 0), planner(dist.group.WORLD, 0), planner(dist.group.WORLD, 1)]
assert ps[2].native_collective and not ps[1].native_collective and not ps[0].native_collective
state, ab = torch.tensor([0.01, 0.0, -1.0, 0.02, 0.0], dtype=torch.float64), torch.zeros(4, nu, dtype=torch.float64)
with torch.no_grad():
    for step in range(4):
        acts = [p.command(state, ab) for p in ps]
        assert torch.equal(acts[0], acts[1]) and torch.equal(acts[0], acts[2]), (step, acts)
        assert torch.equal(ps[0].U, ps[2].U) and torch.equal(ps[0].omega, ps[2].omega)
        ab = torch.roll(ab, -1, 0); ab[-1] = acts[0].cpu()
    ps[2].ctx.profile(True)
    ps[2].command(state, ab); torch.cuda.synchronize()
    ps[2].ctx.profile(False)
    assert "rccl_all_gather" in ps[2].ctx.profile_read()
# a second communicator on a fresh ctx, and the error paths of the C ABI
import ctypes as C
c = n._lib.Ctx(0)
try:
    c.check(c.lib.nlc_comm_init(c.h, 3, 2, C.c_char_p(b"x" * 128)))
    raise SystemExit("bad rank accepted")
except n._lib.NlcError as e:
    assert e.code == -1
c.comm_init(0, 1, c.comm_unique_id())
c.check(c.lib.nlc_comm_destroy(c.h))
dist.destroy_process_group()
open(os.path.join(sys.argv[2], "ok"), "w").write("ok")
"""


def test_native_collective_one_rank_rccl(nlc, tmp_path):
    """include/nlc.h's own communicator (nlc_comm_unique_id / nlc_comm_init; nlc_mppi_finish with gathered_dev == NULL
    runs ncclAllGather on the command's stream): a one-rank RCCL group on this box's one GPU.  The planner with the
    native collective, the one with torch.distributed's and the one without a group return bit-identical actions, U and
    omega over consecutive commands."""
    import subprocess
    import sys

    from oracle import nl_model as onl

    repo = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    st = onl.ENV_STATS["oderl-cartpole"]
    torch.save(onl.make_synthetic_state_dict(8, 5, 1, 128, 17, st["state_std"], [1.5], tame=True), tmp_path / "sd.pt")
    script = tmp_path / "worker.py"
    script.write_text(_NATIVE_COLLECTIVE_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    subprocess.check_call(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
         "--master-port", "29547", str(script), repo, str(tmp_path)], env=env, timeout=600)
    assert (tmp_path / "ok").read_text() == "ok"


def test_two_process_sharded_planner_end_to_end(nlc, tmp_path):
    """`MPPIDelay(process_group=...)` through torch.distributed.run with world_size 2 (both ranks on this one GPU,
    gloo collective): every rank returns the same action, and it equals the unsharded planner's (Philox counters are
    global sample indices, so the draw does not depend on the sharding)."""
    import subprocess
    import sys

    from oracle import nl_model as onl

    repo = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    d, nu, A, K, T = 5, 1, 3.0, 1024, 8
    st = onl.ENV_STATS["oderl-cartpole"]
    sd = onl.make_synthetic_state_dict(8, d, nu, 128, 17, st["state_std"], [A / 2], tame=True)
    torch.save(sd, tmp_path / "sd.pt")
    from oracle import node_model as onode
    from oracle import rnn_model as ornn

    sd_rnn = ornn.make_synthetic_state_dict(8, d, nu, 64, st["state_std"], [A / 2])
    sd_node = onode.make_synthetic_state_dict(8, d, nu, 64, 1, st["state_std"], [A / 2])
    torch.save(sd_rnn, tmp_path / "sd_rnn.pt")
    torch.save(sd_node, tmp_path / "sd_node.pt")
    script = tmp_path / "worker.py"
    script.write_text(_TWO_RANK_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    subprocess.check_call(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", "29541", str(script), repo, str(tmp_path)], env=env, timeout=600)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    model = build_model(nlc, sd)
    state, ab = torch.tensor([0.01, 0.0, -1.0, 0.02, 0.0], dtype=torch.float64), torch.zeros(4, nu, dtype=torch.float64)
    for name, dyn in (("nl", nlc.NLDynamics(model, 0.05)), ("oracle", nlc.OracleDynamics("oderl-cartpole", 0.05, 2)),
                      ("dtrnn", nlc.NLDynamics(build_rnn(nlc, sd_rnn, 64), 0.05)),
                      ("node", nlc.NLDynamics(build_node(nlc, sd_node, 64, 1), 0.05))):
        assert torch.equal(r0[name]["acts"], r1[name]["acts"]) and torch.equal(r0[name]["U"], r1[name]["U"])
        p = nlc.MPPIDelay(dyn, nlc.EnvCost("oderl-cartpole"), d, nlc.noise_sigma(nu), K, T, "cpu", lambda_=1.0,
                          u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A,
                          U_init=torch.zeros(T, nu, dtype=torch.float64), noise_rng="philox", seed=21)
        with torch.no_grad():
            acts = torch.stack([p.command(state, ab) for _ in range(3)])
        np.testing.assert_allclose(r0[name]["acts"].numpy(), acts.numpy(), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(r0[name]["U"].numpy(), p.U.numpy(), rtol=1e-10, atol=1e-12)
        # the shards hold the two halves of the unsharded planner's last draw and weights
        both = torch.cat((r0[name]["noise"], r1[name]["noise"]))
        np.testing.assert_allclose(both.numpy(), p.noise.numpy(), rtol=0, atol=1e-12)
        np.testing.assert_allclose(torch.cat((r0[name]["omega"], r1[name]["omega"])).numpy(), p.omega.numpy(),
                                   rtol=1e-9, atol=1e-15)
    # rollout_samples = 3 with a variance cost: sharded == unsharded (the reference's statistic over all K samples)
    p = nlc.MPPIDelay(nlc.OracleDynamics("oderl-cartpole", 0.05, 1), nlc.EnvCost("oderl-cartpole"), d, nlc.noise_sigma(nu), K,
                      T, "cpu", lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A,
                      U_init=torch.zeros(T, nu, dtype=torch.float64), noise_rng="philox", seed=22, rollout_samples=3,
                      rollout_var_cost=0.7, rollout_var_discount=0.9)
    a = p.command(state, ab)
    np.testing.assert_allclose(r0["varcost"]["acts"].numpy(), a.numpy(), rtol=1e-10, atol=1e-12)
    both = torch.cat((r0["varcost"]["cost"], r1["varcost"]["cost"]))
    np.testing.assert_allclose(both.numpy(), p.cost_total.numpy(), rtol=1e-11, atol=1e-11)
    plain = nlc.MPPIDelay(nlc.OracleDynamics("oderl-cartpole", 0.05, 1), nlc.EnvCost("oderl-cartpole"), d, nlc.noise_sigma(nu), K,
                          T, "cpu", lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A,
                          U_init=torch.zeros(T, nu, dtype=torch.float64), noise_rng="philox", seed=22)
    plain.command(state, ab)
    shift = p.cost_total - plain.cost_total
    assert float(shift.min()) > 1e-6 and float(shift.max() - shift.min()) < 1e-9  # one constant, as in the reference
    # the tensor-op planner (nu = 3) sharded over the two ranks == the same planner alone, under the same seed
    gen = torch.Generator().manual_seed(5)
    Wx3, Wu3 = torch.randn(4, 4, generator=gen, dtype=torch.float64) * 0.3, torch.randn(3, 4, generator=gen, dtype=torch.float64) * 0.5
    dyn3 = lambda s, w: s + 0.05 * (torch.tanh(s @ Wx3.to(s.device)) + (0.7 * w[:, -1, :] + 0.3 * w[:, 0, :]) @ Wu3.to(s.device))  # noqa: E731
    cost3 = lambda s, u: (s ** 2).sum(dim=1) + 0.01 * (u ** 2).sum(dim=1)  # noqa: E731
    torch.manual_seed(77)
    with pytest.warns(UserWarning, match="PyTorch-ROCm tensor ops"):
        p3 = nlc.MPPIDelay(dyn3, cost3, 4, nlc.noise_sigma(3), 512, 6, "cpu", lambda_=1.0, u_min=torch.tensor(-2.0),
                           u_max=torch.tensor(2.0), u_scale=2.0, U_init=torch.zeros(6, 3, dtype=torch.float64))
    st3, ab3 = torch.linspace(-0.5, 0.5, 4, dtype=torch.float64), torch.zeros(4, 3, dtype=torch.float64)
    acts3 = torch.stack([p3.command(st3, ab3) for _ in range(3)])
    assert torch.equal(r0["torch_path"]["acts"], r1["torch_path"]["acts"]) and torch.equal(r0["torch_path"]["U"], r1["torch_path"]["U"])
    np.testing.assert_allclose(r0["torch_path"]["acts"].numpy(), acts3.numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(r0["torch_path"]["U"].numpy(), p3.U.numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(torch.cat((r0["torch_path"]["omega"], r1["torch_path"]["omega"])).numpy(), p3.omega.numpy(),
                               rtol=1e-9, atol=1e-15)


_TIMEOUT_RANK_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import neurallaplacecontrol_amd as n
import numpy as np
dist.init_process_group("gloo")            # the ranks share cuda:0; the partial rows travel over gloo
rank, world = dist.get_rank(), dist.get_world_size()
sd = torch.load(os.path.join(sys.argv[2], "sd.pt"))
d, nu, A, K, T = 5, 1, 3.0, int(sys.argv[3]), 12
bad_rank = int(sys.argv[4])
model = n.NeuralLaplaceModel(d, nu, d, hidden_units=128, s_recon_terms=17, ilt_algorithm="fourier", state_mean=np.zeros(d),
                             state_std=np.ones(d), action_mean=np.array([0]), action_std=np.array([1.0]),
                             normalize=True, normalize_time=True).double()
model.load_state_dict(sd)
model = model.cuda()
opts = {"rollout_variant": 3, "fused_inline": int(sys.argv[5])}
if rank == bad_rank:                       # one encoder tile of THIS rank is never published: its chain gives up
    opts.update({"fused_test_drop_tile": 37, "fused_spin_limit": 3000})
p = n.MPPIDelay(n.NLDynamics(model, 0.05), n.EnvCost("oderl-cartpole"), d, n.noise_sigma(nu), K, T, "cpu",
                compute_device="cuda:0", lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A,
                U_init=torch.zeros(T, nu, dtype=torch.float64), noise_rng="philox", seed=31,
                process_group=dist.group.WORLD, planner_options=opts)
state, ab = torch.tensor([0.01, 0.0, -1.0, 0.02, 0.0], dtype=torch.float64), torch.zeros(4, nu, dtype=torch.float64)
acts, fused_first = [], None
with torch.no_grad():
    for i in range(3):
        p.ctx.profile_reset(); p.ctx.profile(True)
        a = p.command(state, ab)
        p.ctx.profile(False)
        prof = p.ctx.profile_read()
        if i == 0:
            fused_first = prof.get("nl_plan_fused_kernel", {}).get("launches", 0), prof.get("nl_rollout_kernel", {}).get("launches", 0)
        acts.append(a.clone())
        ab = torch.roll(ab, -1, 0); ab[-1] = a
torch.save(dict(acts=torch.stack(acts), U=p.U.cpu(), cost=p.cost_total.cpu(), fused_first=fused_first,
                last_fused="nl_plan_fused_kernel" in prof, body=p.rollout_body, timeouts=p.fused_timeouts,
                fallbacks=p.fused_fallbacks, giveup_at=int(p.ctx.get_stat("last_giveup_command"))),
           os.path.join(sys.argv[2], f"r{rank}.pt"))
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world,K,bad_rank,inline", [(2, 1024, 1, 3), (4, 2048, 2, 3), (2, 1024, 0, 2), (2, 1024, 1, 0)])
def test_sharded_fused_timeout_is_recovered_on_every_rank(nlc, tmp_path, world, K, bad_rank, inline):
    """VERDICT r3 weak 6 / ADVICE r3: a fused-body time-out on ONE rank of a K-sharded planner.  The rank that gave up marks its
    partial row (eta = -1); after the all-gather merge_kernel on every rank sees the mark, skips the update and tells its
    host; every rank re-runs the command on the two-launch body, the partials are gathered again (NLC_AGAIN: the collective
    is the caller's here) and merged -- no rank is left waiting in a collective, every rank returns the same action, and it
    is the unsharded planner's.  The rank that timed out stays on the two-launch body, its peers keep the fused one.
    ADVICE r4 (medium): the same when the weights are folded OUTSIDE the launch (fused_inline 0 / 2): the weight kernels read the
    launch's give-up word and mark the row, so the peers learn of the loss there too."""
    import subprocess
    import sys

    from oracle import nl_model as onl

    repo = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    d, nu, A, T = 5, 1, 3.0, 12
    st = onl.ENV_STATS["oderl-cartpole"]
    sd = onl.make_synthetic_state_dict(8, d, nu, 128, 17, st["state_std"], [A / 2], tame=True)
    torch.save(sd, tmp_path / "sd.pt")
    script = tmp_path / "worker.py"
    script.write_text(_TIMEOUT_RANK_WORKER)
    port = str(29551 + world)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    subprocess.check_call(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
         "--master-port", port, str(script), repo, str(tmp_path), str(K), str(bad_rank), str(inline)], env=env, timeout=600)
    rs = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    for r in range(1, world):
        assert torch.equal(rs[0]["acts"], rs[r]["acts"]) and torch.equal(rs[0]["U"], rs[r]["U"]), r
    assert bool(torch.isfinite(rs[0]["acts"]).all())
    for r in range(world):
        # first command: one fused launch everywhere, then the re-run's rollout launch everywhere
        assert rs[r]["fused_first"] == (1, 1), (r, rs[r]["fused_first"])
        assert rs[r]["last_fused"] == (r != bad_rank), r
        # nlc_get_stat (ABI v9): the give-up is visible -- one launch lost on the bad rank, one re-run everywhere, in command 0
        assert rs[r]["timeouts"] == (1 if r == bad_rank else 0) and rs[r]["fallbacks"] == 1 and rs[r]["giveup_at"] == 0, rs[r]
        assert rs[r]["body"] == ("latency-split" if r == bad_rank else "fused"), rs[r]["body"]
    model = build_model(nlc, sd)
    p = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost("oderl-cartpole"), d, nlc.noise_sigma(nu), K, T, "cpu", lambda_=1.0,
                      u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, U_init=torch.zeros(T, nu, dtype=torch.float64),
                      noise_rng="philox", seed=31, planner_options={"rollout_variant": 2})
    state, ab = torch.tensor([0.01, 0.0, -1.0, 0.02, 0.0], dtype=torch.float64), torch.zeros(4, nu, dtype=torch.float64)
    acts = []
    with torch.no_grad():
        for _ in range(3):
            a = p.command(state, ab)
            acts.append(a.clone())
            ab = torch.roll(ab, -1, 0)
            ab[-1] = a
    np.testing.assert_allclose(rs[0]["acts"].numpy(), torch.stack(acts).numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(rs[0]["U"].numpy(), p.U.numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(torch.cat([r["cost"] for r in rs]).numpy(), p.cost_total.numpy(), rtol=1e-12, atol=1e-12)


def test_batched_planner_rollout_samples_per_episode_variance(nlc):
    """rollout_samples > 1 in BatchedMPPIDelay: episode e gets ITS population's variance term, exactly what a single
    MPPIDelay with the same options computes for it (reference mppi_delay.py:291-292, 310)."""
    env, K, T, E, A = "oderl-pendulum", 200, 7, 3, 2.0
    kw = dict(lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, rollout_samples=2, rollout_var_cost=0.5,
              rollout_var_discount=0.8)
    torch.manual_seed(9)
    raw = torch.randn(E, K, T, 1, dtype=torch.float64)
    U0 = torch.randn(E, T, 1, dtype=torch.float64) * 0.3
    states = torch.stack([nlc.initial_state(env) + 0.05 * e for e in range(E)])
    abs_ = torch.randn(E, 4, 1, dtype=torch.float64)
    b = nlc.BatchedMPPIDelay(nlc.OracleDynamics(env, 0.05, 1), nlc.EnvCost(env), 3, nlc.noise_sigma(1), E, K, T, "cpu",
                             U_init=U0.clone(), **kw)
    b.noise_dist = type("R", (), {"sample": staticmethod(lambda shape: raw)})()
    acts = b.command(states, abs_)
    for e in range(E):
        p = nlc.MPPIDelay(nlc.OracleDynamics(env, 0.05, 1), nlc.EnvCost(env), 3, nlc.noise_sigma(1), K, T, "cpu",
                          U_init=U0[e].clone(), **kw)
        p.noise_dist = type("R", (), {"sample": staticmethod(lambda shape, e=e: raw[e])})()
        a = p.command(states[e], abs_[e])
        assert torch.equal(a, acts[e])
        np.testing.assert_allclose(b.cost_total[e].numpy(), p.cost_total.numpy(), rtol=1e-13, atol=1e-13)


def test_batched_planner_acrobot_nl_u_per_command(nlc):
    """nu = 2 NL dynamics, two actions per command, K ragged against the 16-sample tiles."""
    from oracle import nl_model as onl

    env, d, nu, A = "oderl-acrobot", 6, 2, 5.0
    st = onl.ENV_STATS[env]
    sd = onl.make_synthetic_state_dict(14, d, nu, 128, 17, st["state_std"], [A / 2], tame=True)
    model = build_model(nlc, sd)
    from neurallaplacecontrol_amd.planners.mppi_batch import BatchedMPPIDelay

    E, K, T = 3, 50, 5
    g = torch.Generator().manual_seed(77)
    U0 = torch.randn(E, T, nu, dtype=torch.float64, generator=g) * 0.3
    raw = torch.randn(E, K, T, nu, dtype=torch.float64, generator=g)
    states = torch.stack([_state(nlc, env, e) for e in range(E)])
    ab = torch.randn(E, 4, nu, dtype=torch.float64, generator=g)
    kw = dict(lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, u_per_command=2)
    bat = BatchedMPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, nlc.noise_sigma(nu), E, K, T, "cpu",
                           U_init=U0.clone(), **kw)
    bat.noise_dist = _Replay(raw.clone())
    with torch.no_grad():
        act = bat.command(states, ab)
        assert act.shape == (E, 2, nu)
        for e in range(E):
            m = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, nlc.noise_sigma(nu), K, T, "cpu",
                              U_init=U0[e].clone(), **kw)
            m.noise_dist = _Replay(raw[e].clone())
            assert torch.equal(m.command(states[e], ab[e]), act[e])
            assert torch.equal(m.U, bat.U[e])


@pytest.mark.parametrize("ranks,samples", [(2, 16384), (4, 8192)])
def test_bench_ranks_rehearsed_on_one_gpu(ranks, samples):
    """The whole N > 1 flow of bench.py on the 1-GPU box: `python bench.py --gpus N` starts its own ranks, each plans its K / N
    shard on cuda:0 (the ranks talk over gloo: RCCL refuses two ranks per device), rank-consistent pre-heat, timed steps
    between barriers, MAX over ranks, ONE JSON line from rank 0, orderly teardown.  The numbers mean nothing; the run must
    end with exit code 0 and a well-formed line.  Two ranks: the latency-split body (8192 samples each).  Four ranks of 2048
    samples: the one-launch fused body on every rank, as on the 8-GPU node -- four is what the GPU pool's process guard
    allows beside this test process and the launcher (at most six processes on a card; five ranks were killed by it), so
    the shard SIZE of the 8-GPU run is kept and the rank count is not."""
    import json
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", str(ranks), "--rehearse-on-one-gpu", "--steps", "12",
                          "--warmup", "2", "--preheat-ms", "40", "--no-ilt", "--no-cpu-baseline", "--samples", str(samples)],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == ranks and rec["steps"] == 12 and rec["value"] > 0 and rec["scaling"] == "strong"
    assert rec["config"]["samples_per_gpu"] == samples // ranks and "gloo" in rec["config"]["collective"]
    if samples // ranks <= 4096:
        assert "nl_plan_fused_kernel" in rec["kernels_avg_ms"], "a 2048-sample shard plans on the one-launch fused body"
    else:
        assert "nl_rollout_kernel" in rec["kernels_avg_ms"]
    # round 5: what the ranks saw -- the group size, one entry per rank with its body, its give-ups and its kernel averages
    seen = rec["config"]["ranks_seen"]
    assert seen["torch_world"] == ranks and seen["library_comm_world"] == 0 and len(seen["per_rank"]) == ranks
    assert sorted(r["rank"] for r in seen["per_rank"]) == list(range(ranks))
    for r in seen["per_rank"]:
        assert r["rollout_body"] == ("fused" if samples // ranks <= 4096 else "latency-split"), r
        assert r["fused_timeouts"] == 0 and r["fused_fallbacks"] == 0, "ranks sharing one GPU must not lose fused launches silently"
        assert r["collective_timing"]["torch_all_gather_avg_ms"] > 0 and r["kernels_avg_ms"]
    assert rec["config"]["commit"] and rec["config"]["library_build"]["csrc_sha"]


def _bench_line(args, timeout=900):
    import json
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         timeout=timeout)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_bench_config3_shards_rehearsed_on_one_gpu():
    """BASELINE configs[3] as it is worded -- acrobot, nu = 2, T = 60, 32 768 samples per rank, 122 doubles per
    rank in the all-gather -- through bench.py's own N > 1 flow.  The GPU pool's process guard allows four ranks on one card (at
    most six processes beside this one and the launcher), so four of the eight 32 768-sample shards run (`--samples 131072`: the
    shard SIZE, body and payload of the 8-GPU run); the eight-shard merge itself is test_config3_eight_shards_merge_equals_the_
    whole_population below.  Same command count on both runs (no time-boxed pre-heat), device Philox noise (G-invariant):
    the sharded run's action equals the one-GPU whole-population action."""
    common = ["--config", "3", "--samples", "131072", "--steps", "3", "--warmup", "1", "--preheat-ms", "0", "--no-ilt", "--no-cpu-baseline"]
    rec = _bench_line(["--gpus", "4", "--rehearse-on-one-gpu"] + common)
    assert rec["n_gpus"] == 4 and rec["steps"] == 3 and rec["value"] > 0 and rec["config"]["samples_per_gpu"] == 32768
    assert "acrobot" in rec["config"]["workload"] and "H=60" in rec["config"]["workload"] and rec["config"]["baseline_config"] == "3"
    seen = rec["config"]["ranks_seen"]
    assert seen["torch_world"] == 4 and sorted(r["rank"] for r in seen["per_rank"]) == [0, 1, 2, 3]
    for r in seen["per_rank"]:
        assert r["rollout_body"] == "wave-per-tile" and r["collective_timing"]["doubles_per_rank"] == 2 + 60 * 2, r
    one = _bench_line(["--gpus", "1"] + common)
    assert one["n_gpus"] == 1 and one["config"]["samples_per_gpu"] == 131072
    np.testing.assert_allclose(rec["config"]["last_action"], one["config"]["last_action"], rtol=0, atol=1e-10)


def test_config3_eight_shards_merge_equals_the_whole_population(nlc):
    """BASELINE configs[3]'s sharding at its own numbers on one GPU, in one process: eight planners each own 32 768 of the
    262 144 acrobot samples (nu = 2, T = 60, NL dynamics, device Philox noise keyed by the GLOBAL sample index), their eight
    (beta_r, eta_r, S_r) rows -- 122 doubles each, what the RCCL all-gather carries -- go through nlc_mppi_finish(G = 8) on every
    shard: action and U equal the one-GPU whole-population planner's to 1e-10 (planners/mppi_delay.py:210-216)."""
    import ctypes as C

    from neurallaplacecontrol_amd import _lib
    from oracle import nl_model as onl

    env, K, T, G = "oderl-acrobot", 262144, 60, 8
    st = onl.ENV_STATS[env]
    d, nu, A = st["d"], st["nu"], st["act_high"]
    sd = onl.make_synthetic_state_dict(0, d, nu, 128, 17, st["state_std"], [A / 2], tame=True)
    model = build_model(nlc, sd)
    g = torch.Generator().manual_seed(3)
    U0 = torch.randn(T, nu, dtype=torch.float64, generator=g) * 0.2
    state = nlc.initial_state(env)
    ab = (torch.rand(4, nu, dtype=torch.float64, generator=g) * 2 - 1) * A

    def planner():
        return nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, nlc.noise_sigma(nu), K, T, "cpu", lambda_=1.0,
                             u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, U_init=U0.clone(), noise_rng="philox", seed=5,
                             store_rollouts=False)

    full = planner()
    a_full = full.command(state, ab)
    U_full, omega_full = full.U.clone(), full.omega.cpu()
    del full
    torch.cuda.empty_cache()
    shards = []
    for r in range(G):
        p = planner()
        p.K_local, p.k_offset = K // G, r * (K // G)  # phase 1 of rank r stand-alone (G = 1 inside); merged by hand below
        p.command(state, ab)
        assert p._partials.numel() == 2 + T * nu == 122 and p.rollout_body == "wave-per-tile"
        shards.append(p)
    gathered = torch.stack([s._partials for s in shards]).contiguous()
    U_shift = torch.roll(U0, -1, 0)
    U_shift[-1] = 0
    for r, p in enumerate(shards):
        p.U = U_shift
        act = torch.empty(nu, dtype=torch.float64)
        p.ctx.check(p.ctx.lib.nlc_mppi_finish(p.ctx.h, _lib.ptr(gathered), G, r, C.byref(p._buf), _lib.ptr(act)))
        np.testing.assert_allclose(act.numpy(), a_full.numpy(), rtol=0, atol=1e-10)
        np.testing.assert_allclose(p.U.numpy(), U_full.numpy(), rtol=0, atol=1e-10)
        np.testing.assert_allclose(p.omega.cpu().numpy(), omega_full[r * (K // G) : (r + 1) * (K // G)].numpy(), rtol=1e-9, atol=1e-18)
    del shards
    torch.cuda.empty_cache()


def _independent_planner(idx, K, T, commands, barrier, q, opts):
    """One of the reference's evaluation workers (run_exp_multi.py:145-165): its own process, its own planner, cuda:0."""
    import torch

    import bench
    import neurallaplacecontrol_amd as nlc

    torch.set_num_threads(1)
    d, nu = 5, 1
    model = bench.synthetic_state_dict(d, nu, bench.S_TERMS).to("cuda:0")
    state = nlc.initial_state(bench.ENV, torch.Generator().manual_seed(idx))
    p = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(bench.ENV), d, nlc.noise_sigma(nu), num_samples=K, horizon=T,
                      device="cpu", compute_device="cuda:0", lambda_=1.0, u_min=torch.tensor(-3.0), u_max=torch.tensor(3.0),
                      u_scale=3.0, noise_rng="philox", seed=100 + idx, U_init=torch.zeros(T, nu, dtype=torch.float64),
                      planner_options=opts)
    ab = torch.zeros(4, nu, dtype=torch.float64)
    if barrier is not None:
        barrier.wait()  # the FIRST commands of all workers start together: nothing warmed up, nothing swallowed
    acts, first_body = [], None
    with torch.no_grad():
        for i in range(commands):
            a = p.command(state, ab)
            if i == 0:
                first_body = p.rollout_body
            acts.append(a.clone())
            ab = torch.roll(ab, -1, 0)
            ab[-1] = a
    out = dict(idx=idx, acts=torch.stack(acts).numpy(), first_body=first_body, last_body=p.rollout_body, timeouts=p.fused_timeouts,
               fallbacks=p.fused_fallbacks, lost=int(p.ctx.get_stat("fused_lost")))
    if q is None:
        return out
    q.put(out)
    barrier.wait()


def test_independent_planner_processes_sharing_one_gpu_keep_the_fused_body(nlc):
    """the reference's deployment is many evaluation workers on ONE GPU, one MPPIDelay of K = 1000, T = 40
    each (run_exp_multi.py:145-165, config.py:52), and `rollout_variant` auto gives every one of them the fused one-launch
    body, which waits inside the launch for workgroups of the same launch.  Four such processes, 200 commands each, counted
    from the first command: every action equals the same planner's solo run bit for bit, no fused launch gave up, nobody was
    moved off the fused body behind the caller's back (measured at 1 / 3 / 6 processes: profiles/r5_fused_sharing.json)."""
    import multiprocessing as mp

    P, K, T, commands = 4, 1000, 40, 200
    ctx = mp.get_context("spawn")
    barrier, q = ctx.Barrier(P), ctx.Queue()
    procs = [ctx.Process(target=_independent_planner, args=(i, K, T, commands, barrier, q, {})) for i in range(P)]
    for pr in procs:
        pr.start()
    res = sorted((q.get(timeout=600) for _ in range(P)), key=lambda r: r["idx"])
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    for r in res:
        assert (r["first_body"], r["last_body"]) == ("fused", "fused"), r
        assert (r["timeouts"], r["fallbacks"], r["lost"]) == (0, 0, 0), "a fused launch gave up while the GPU was shared"
        # the same planner alone on the GPU, on the two-launch body: the bodies are bit-identical, sharing changed nothing
        solo = _independent_planner(r["idx"], K, T, commands, None, None, {"rollout_variant": 2})
        assert solo["last_body"] == "latency-split"
        assert np.array_equal(solo["acts"], r["acts"]), r["idx"]
