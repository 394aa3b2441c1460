"""Host-side logic of the Python mirror that needs no GPU."""

import os
import numpy as np
import pytest
import torch
import torch.nn as nn


def _nl_model():
    import neurallaplacecontrol_amd as nlc

    d, nu = 3, 1
    return nlc.NeuralLaplaceModel(
        d, nu, d, hidden_units=128, s_recon_terms=17, ilt_algorithm="fourier", state_mean=np.zeros(d),
        state_std=np.ones(d), action_mean=np.array([0]), action_std=np.array([1.0]), normalize=True, normalize_time=True,
    ).double()


def _rnn_model():
    from neurallaplacecontrol_amd.rnn_model import DeltaTRNN

    return DeltaTRNN(3, 1, 64, state_mean=np.zeros(3), state_std=np.ones(3), action_mean=np.array([0]),
                     action_std=np.array([1.0]), normalize=True, normalize_time=True).double()


def _node_model():
    from neurallaplacecontrol_amd.node_model import NODE

    return NODE(3, 1, 3, state_mean=np.zeros(3), state_std=np.ones(3), action_mean=np.array([0]), action_std=np.array([1.0]),
                normalize=True, normalize_time=True).double()


@pytest.mark.parametrize("make", [_nl_model, _rnn_model, _node_model])
def test_weights_key_detects_every_kind_of_weight_change(make):
    """The planner re-uploads weights when ``_weights_key()`` changes (``MPPIDelay._ensure_configured``).  Round 1 cached
    the tensor list and missed replaced parameters / buffers; every kind of change must move the key now."""
    try:
        m = make()
    except TypeError as e:  # constructor signature of a twin differs: not what this test is about
        pytest.skip(str(e))
    k0 = m._weights_key()
    assert m._weights_key() == k0  # stable while nothing changes
    first = next(m.parameters())
    with torch.no_grad():
        first.mul_(2.0)  # in-place write under no_grad
    k1 = m._weights_key()
    assert k1 != k0
    lin = next(mod for mod in m.modules() if isinstance(mod, nn.Linear))
    lin.weight = nn.Parameter(lin.weight.detach().clone() * 3.0)  # parameter object replaced
    k2 = m._weights_key()
    assert k2 != k1
    m.state_std = torch.full_like(m.state_std, 2.0)  # buffer replaced by attribute assignment
    k3 = m._weights_key()
    assert k3 != k2
    m.load_state_dict(m.state_dict())  # copy_ into the same storage: versions move
    k4 = m._weights_key()
    assert k4 != k3
    m.float().double()  # _apply re-allocates
    k5 = m._weights_key()
    assert k5 != k4
    # documented limitation: a write through .data is invisible (own version counter) -> explicit mark
    next(m.parameters()).data.mul_(0.5)
    assert m._weights_key() == k5
    m.mark_weights_dirty()
    assert m._weights_key() != k5


# --------------------------------------------------------------------------- literal harness closures
def _harness_style_closures(model, env, ts_pred, device="cpu", encode_obs_time=False, action_buffer_size=4, model_name="nl",
                            state_constraint=False, change_goal=False):
    """Closures of the SHAPE the reference harness builds (mppi_with_model.py:103-122, 145-171): same free variables."""

    def dynamics(state, perturbed_action, encode_obs_time=encode_obs_time, action_buffer_size=action_buffer_size,
                 model_name=model_name):
        if encode_obs_time and model_name == "nl":
            perturbed_action = torch.cat(
                (perturbed_action, torch.flip(torch.arange(action_buffer_size, device=device), (0,))
                 .view(1, action_buffer_size, 1).repeat(perturbed_action.shape[0], 1, 1)), dim=2)
        state_diff_pred = model(state, perturbed_action, ts_pred)
        return state + state_diff_pred

    def running_cost(state, action):
        if state_constraint:
            reward = env.diff_obs_reward_(state, exp_reward=False, state_constraint=state_constraint) + env.diff_ac_reward_(action)
        elif change_goal:
            reward = env.diff_obs_reward_(state, exp_reward=False, change_goal=change_goal) + env.diff_ac_reward_(action)
        else:
            reward = env.diff_obs_reward_(state, exp_reward=False) + env.diff_ac_reward_(action)
        return -reward

    return dynamics, running_cost


def test_recognise_literal_harness_closures_without_gpu():
    """Free-variable inspection only (no GPU, nothing executed): a model + constant ts_pred -> NLDynamics candidate, an
    oracle partial -> OracleDynamics, an env on the default reward branch -> EnvCost; anything else -> no candidate."""
    import functools

    import numpy as np

    import neurallaplacecontrol_amd as nlc
    from neurallaplacecontrol_amd import _recognise as R

    model = nlc.NeuralLaplaceModel(5, 1, 5, hidden_units=128, s_recon_terms=17, ilt_algorithm="fourier",
                                   state_mean=np.zeros(5), state_std=np.ones(5), action_mean=np.array([0]),
                                   action_std=np.array([1.5]), normalize=True, normalize_time=True).double()

    class CTCartpole:  # stand-in with the two methods the closure calls (class name as in envs/oderl/envs/ctcartpole.py)
        def diff_obs_reward_(self, s, exp_reward=False, **kw):
            return -s.pow(2).sum(-1)

        def diff_ac_reward_(self, a):
            return -0.01 * a.pow(2).sum(-1)

    ts_pred = torch.tensor(0.05, dtype=torch.double).view(1, 1).repeat(64, 1)
    dyn, cost = _harness_style_closures(model, CTCartpole(), ts_pred)
    cd, cc = R.candidate_dynamics(dyn), R.candidate_cost(cost)
    assert isinstance(cd, nlc.NLDynamics) and cd.model is model and cd.ts_pred == 0.05
    assert isinstance(cc, nlc.EnvCost) and cc.env_name == "oderl-cartpole"
    # non-default reward branches are not the plain env cost; a varying ts_pred is not a constant prediction time
    _, cost_sc = _harness_style_closures(model, CTCartpole(), ts_pred, state_constraint=True)
    assert R.candidate_cost(cost_sc) is None
    dyn_var, _ = _harness_style_closures(model, CTCartpole(), torch.linspace(0.01, 0.05, 64, dtype=torch.double).view(-1, 1))
    assert R.candidate_dynamics(dyn_var) is None
    # two models in the closure: ambiguous -> no candidate; plain lambdas without free variables -> none
    other = model

    def two(state, w, m1=None):
        return model(state, w, ts_pred) + other(state, w, ts_pred) * 0

    assert R.candidate_dynamics(two) is None
    assert R.candidate_dynamics(lambda s, w: s) is None and R.candidate_cost(lambda s, u: s.sum(-1)) is None
    # oracle partial, as mppi_with_model.py:129-143 builds it

    def cartpole_dynamics_dt_delay(state, perturbed_action, ts, delay, friction=False):
        return state

    od = R.candidate_dynamics(functools.partial(cartpole_dynamics_dt_delay, ts=ts_pred, delay=2, friction=True))
    assert isinstance(od, nlc.OracleDynamics) and (od.env_name, od.ts, od.delay, od.friction) == ("oderl-cartpole", 0.05, 2, True)
    assert R.candidate_dynamics(functools.partial(cartpole_dynamics_dt_delay, ts=ts_pred, delay=2, extra=1)) is None

    def some_other_name(state, perturbed_action, ts, delay, friction=False):
        return state

    assert R.candidate_dynamics(functools.partial(some_other_name, ts=ts_pred, delay=0)) is None
    # a probe cannot see logic outside the probed region, so a closure must also have the harness closure's
    # STRUCTURE (names, free variables, constants).  Same free variables, but a clamp / a threshold / a helper call / extra
    # state-dependent logic -> no candidate, whatever a probe would say.
    env = CTCartpole()

    def clamped(state, perturbed_action):
        return torch.clamp(state + model(state, perturbed_action, ts_pred), -10.0, 10.0)

    def thresholded(state, perturbed_action):
        out = state + model(state, perturbed_action, ts_pred)
        return out * (out.abs() < 1000.0)

    def wrong_args(x, w):
        return x + model(x, w, ts_pred)

    def cost_plus_barrier(state, action):
        return -(env.diff_obs_reward_(state, exp_reward=False) + env.diff_ac_reward_(action)) + torch.relu(state[:, 0] - 2.0)

    def cost_scaled(state, action):
        return -(env.diff_obs_reward_(state, exp_reward=False) + 0.5 * env.diff_ac_reward_(action))

    # operators and small integers need no names -- the gate fingerprints the OPERATIONS too
    def masked_store(state, perturbed_action):
        out = state + model(state, perturbed_action, ts_pred)
        out[out[:, 0] > 2] = 2
        return out

    def minus(state, perturbed_action):
        return state - model(state, perturbed_action, ts_pred)

    def small_mask(state, perturbed_action):
        out = state + model(state, perturbed_action, ts_pred)
        return out * (out < 3)

    def twice(state, perturbed_action):
        return state + model(state, perturbed_action, ts_pred) + model(state, perturbed_action, ts_pred)

    def sliced(state, perturbed_action):
        return state + model(state, perturbed_action[:, 1:], ts_pred)

    def inplace(state, perturbed_action):
        state += model(state, perturbed_action, ts_pred)
        return state

    def cost_not_negated(state, action):
        return env.diff_obs_reward_(state, exp_reward=False) + env.diff_ac_reward_(action)

    def cost_minus(state, action):
        return -(env.diff_obs_reward_(state, exp_reward=False) - env.diff_ac_reward_(action))

    def cost_first_dim(state, action):
        return -(env.diff_obs_reward_(state, exp_reward=False) + env.diff_ac_reward_(action))[0]

    for f in (clamped, thresholded, wrong_args, masked_store, minus, small_mask, twice, sliced, inplace):
        assert not R.has_harness_structure(f, "dynamics") and R.candidate_dynamics(f) is None, f.__name__
    for f in (cost_plus_barrier, cost_scaled, cost_not_negated, cost_minus, cost_first_dim):
        assert not R.has_harness_structure(f, "cost") and R.candidate_cost(f) is None, f.__name__
    assert R.has_harness_structure(dyn, "dynamics") and R.has_harness_structure(cost, "cost")
    # the encode_obs_time variant of the same closure (its time-channel branch taken or not) is the same code object
    dyn_t, _ = _harness_style_closures(model, CTCartpole(), ts_pred, encode_obs_time=True)
    assert R.has_harness_structure(dyn_t, "dynamics")


def test_twin_of_a_foreign_model_follows_its_weight_updates():
    """a closure over an instance of the REFERENCE's model class is planned through a converted twin (a weight
    snapshot).  refresh_twin re-copies the weights whenever the source's (data_ptr, _version) key moved: load_state_dict, an
    in-place optimizer-style write and a replaced buffer all reach the twin; an untouched source costs one key comparison."""
    import numpy as np

    import neurallaplacecontrol_amd as nlc
    from neurallaplacecontrol_amd import _recognise as R

    ours = nlc.NeuralLaplaceModel(3, 1, 3, hidden_units=64, s_recon_terms=9, ilt_algorithm="fourier", state_mean=np.zeros(3),
                                  state_std=np.ones(3), action_mean=np.array([0]), action_std=np.array([1.0]), normalize=True,
                                  normalize_time=True).double()

    # a foreign class with the reference's name, sub-modules and attributes (w_nl.py:66-115): built from ours
    NeuralLaplaceModel = type("NeuralLaplaceModel", (torch.nn.Module,), {})
    src = NeuralLaplaceModel()
    src.action_encoder, src.laplace_rep_func = ours.action_encoder, ours.laplace_rep_func
    for name in ("state_mean", "state_std", "action_mean", "action_std", "dt"):
        src.register_buffer(name, getattr(ours, name).clone())
    for name in ("output_dim", "latent_dim", "s_recon_terms", "ilt_algorithm", "encode_obs_time", "normalize", "normalize_time"):
        setattr(src, name, getattr(ours, name))
    twin = R._model_twin(src)
    assert isinstance(twin, nlc.NeuralLaplaceModel) and twin is not src and twin.__dict__["_twin_source"] is src
    w_src = src.laplace_rep_func.linear_tanh_stack[0].weight
    w_twin = twin.laplace_rep_func.linear_tanh_stack[0].weight
    assert torch.equal(w_src, w_twin) and w_src.data_ptr() != w_twin.data_ptr()
    key0 = twin._weights_key()
    R.refresh_twin(twin)
    assert twin._weights_key() == key0  # nothing moved: no copy
    with torch.no_grad():
        w_src.mul_(1.5)  # an optimizer step
    assert not torch.equal(w_src, w_twin)
    R.refresh_twin(twin)
    assert torch.equal(w_src, w_twin) and twin._weights_key() != key0  # ... and the planner will re-upload
    sd = {k: v * 0.5 for k, v in src.state_dict().items()}
    src.load_state_dict(sd)
    R.refresh_twin(twin)
    assert torch.equal(w_src, w_twin)
    src.state_std = torch.full((3,), 2.0, dtype=torch.float64)  # a replaced buffer
    R.refresh_twin(twin)
    assert torch.equal(twin.state_std, src.state_std)
    # a write through .data moves no version counter -- the per-tensor content check (every
    # TWIN_CONTENT_CHECK_EVERY looks) catches it, refresh_twin(force=True) (= MPPIDelay.refresh_model()) at once
    w_src.data.mul_(0.25)
    R.refresh_twin(twin)
    assert not torch.equal(w_src, w_twin)  # not seen by the key
    for _ in range(R.TWIN_CONTENT_CHECK_EVERY):
        R.refresh_twin(twin)
    assert torch.equal(w_src, w_twin)
    w_src.data.add_(1.0)
    key1 = twin._weights_key()
    R.refresh_twin(twin, force=True)
    assert torch.equal(w_src, w_twin) and twin._weights_key() != key1


def test_no_spill_reload_behind_an_exec_empty_loop_exit_in_the_dehoog_backward_kernel(tmp_path):
    """Round 5: ROCm 7.2 compiled a per-lane copy loop of ilt_dehoog_bwd_kernel into a loop that leaves EXEC empty and placed the
    reload of a spilled loop invariant behind it, before EXEC is restored -- a no-op reload, a division by garbage, a memory
    fault from 1025 blocks on.  tools/scan_exec_hazard.py finds that signature in the assembly; the kernel as written now
    (wave-uniform trip counts) must scan clean, and the scanner must still recognise the pattern (synthetic snippet)."""
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(repo, "tools"))
    import scan_exec_hazard as sc

    bad = tmp_path / "bad.s"
    bad.write_text("""kern:
.LBB0_1:
\tglobal_store_dwordx2 v[8:9], v[6:7], off
\ts_andn2_b64 exec, exec, s[6:7]
\ts_cbranch_execnz .LBB0_1
\ts_branch .LBB0_2
.LBB0_2:
\tscratch_load_dwordx2 v[12:13], off, off offset:8 ; 8-byte Folded Reload
.LBB0_3:
\ts_or_b64 exec, exec, s[94:95]
\ts_endpgm
""")
    hits = [h for h in sc.scan(str(bad)) if h[4]]
    assert len(hits) == 1 and "offset:8" in hits[0][2]
    asm = tmp_path / "dhb.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", str(asm),
                           os.path.join(repo, "neurallaplacecontrol_amd", "csrc", "kernels_dehoog_bwd.hip")],
                          stderr=subprocess.DEVNULL, timeout=900)
    assert [h for h in sc.scan(str(asm)) if h[4]] == []


def test_bench_quotes_pmc_traffic_only_from_the_library_s_own_sources(monkeypatch):
    """`roofline.traffic` comes from a committed PMC summary; a summary collected from OTHER kernel sources
    than the library in use was built from (content hash recorded by __graft_entry__.build() / tools/collect_profiles.sh) is not
    quoted: traffic null + the reason, an error under --strict-pmc; one that names kernels the library does not have fails loudly."""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import sys

    sys.path.insert(0, repo)
    import bench

    pj = {"_meta": {"commit": "abc1234", "csrc_sha": "1111", "device": "MI355X", "date": "2026-10-05",
                    "kernel_names": {"gru_encode": ["void nlc::gru_encode_kernel<64>"]}},
          "gru_encode": {"hbm_bytes_per_launch": 123.0}}
    monkeypatch.setattr(bench, "build_info", lambda: {"commit": "def5678", "csrc_sha": "1111"})
    traffic, src = bench.pmc_traffic("x.json", pj, "gru_encode_kernel")
    assert traffic == 123.0 and "abc1234" in src and "= this build" in src
    monkeypatch.setattr(bench, "build_info", lambda: {"commit": "def5678", "csrc_sha": "2222"})
    traffic, src = bench.pmc_traffic("x.json", pj, "gru_encode_kernel")
    assert traffic is None and src.startswith("STALE") and "1111" in src and "2222" in src
    monkeypatch.setattr(bench, "STRICT_PMC", True)
    with pytest.raises(RuntimeError, match="STALE"):
        bench.pmc_traffic("x.json", pj, "gru_encode_kernel")
    monkeypatch.setattr(bench, "STRICT_PMC", False)
    pj["_meta"]["kernel_names"]["gru_encode"] = ["void nlc::some_other_kernel"]
    with pytest.raises(RuntimeError, match="stale PMC summary"):
        bench.pmc_traffic("x.json", pj, "gru_encode_kernel")
    assert bench.pmc_traffic("x.json", pj, "nl_rollout_kernel") == (None, None)  # no such entry: nothing to quote
    # the hash itself: content of csrc + nlc.h, independent of .git
    import __graft_entry__ as g

    h = g.csrc_sha()
    assert len(h) == 16 and h == g.csrc_sha()


def test_bench_sliced_encoder_section_failure_does_not_cost_the_line():
    """bench.py measures `encoder_int8_sliced` (experimental) in a child process: when that process fails -- here: no GPU -- the
    section becomes an error record and the caller goes on to print its line."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    args = bench.parse_args(["--steps", "2", "--warmup", "1"])
    rec = bench.sliced_encoder_child(args, 0, timeout_s=300)
    assert isinstance(rec, dict) and "error" in rec and "value" not in rec
