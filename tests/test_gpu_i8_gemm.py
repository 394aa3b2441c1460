"""GPU tests (``-m gpu``) of the experimental int8-sliced encoder (``gru_gemm = 1``: csrc/kernels_gru_i8.hip, nlc_i8gemm.h): the
GRU's hidden-state GEMMs as fixed-point digit products on the INT8 matrix pipe.  Bars: the unit check against exact integer /
long-double arithmetic; the REAL reference ReverseGRUEncoder's outputs (G2) at the suite's tolerance; the FP64-MFMA kernel's
latents to 1e-12; the planner's actions to 1e-9."""
import os
import subprocess

import numpy as np
import pytest
import torch

from gpu_common import *  # noqa: F401,F403
from gpu_common import GOLD, TOL, T64, load_sd, build_model

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sliced_gemm_unit_check_against_exact_arithmetic():
    """tools/i8gemm_check.hip: digits of rint(x 2^54) byte for byte, the i8 MFMA's operand / accumulator maps, and 51 200 outputs of
    sliced 16 x 64 x 16 GEMM tiles against the exact product (the FP64 fused-multiply-add chain beside them)."""
    exe = os.path.join(REPO, "tools", "i8gemm_check.bin")
    if not os.path.exists(exe):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-I", os.path.join(REPO, "neurallaplacecontrol_amd", "csrc"),
                               os.path.join(REPO, "tools", "i8gemm_check.hip"), "-o", exe])
    res = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    text = res.stdout.decode()
    assert res.returncode == 0 and text.strip().endswith("OK"), text
    assert "0 wrong bytes" in text and "row 4 q + r: 0 wrong" in text, text


@pytest.mark.parametrize("env", ["cartpole", "pendulum", "acrobot"])
def test_sliced_encoder_vs_reference_golden(nlc, env):
    """G2 again: the int8-sliced encoder against the REAL reference ReverseGRUEncoder (nn.GRU) outputs."""
    g = np.load(f"{GOLD}/g2_stages_{env}.npz")
    sd = load_sd(g)
    model = build_model(nlc, sd)
    ctx = model.hip_ctx(torch.device("cuda:0"))
    ctx.set_option("gru_gemm", 1)
    ctx.set_option("gru_coop", 0)  # (few windows: auto would pick the cooperative FP64 form)
    assert ctx.get_stat("gru_gemm") == 1
    win = T64(g["gru_in"]) * sd["action_std"] + sd["action_mean"]
    with torch.no_grad():
        got = model.encode_actions(win.cuda()).cpu()
    np.testing.assert_allclose(got.numpy(), g["gru_out"], **TOL)


@pytest.mark.parametrize("nu,B,N", [(1, 4, 70000), (2, 4, 4097), (1, 1, 1000), (1, 6, 2049), (2, 3, 17)])
def test_sliced_encoder_agrees_with_fp64_encoder(nlc, nu, B, N):
    """Random windows (ragged N, one to six GRU steps, one and two action dims): latents of the two encoders to 1e-12 -- measured
    3e-16 -- and both against the torch float64 modules at the suite's tolerance."""
    import oracle.nl_model as onl

    d = 5 if nu == 1 else 6
    sd = onl.make_synthetic_state_dict(3, d, nu, 128, 17)
    model = build_model(nlc, sd)
    ctx = model.hip_ctx(torch.device("cuda:0"))
    g = torch.Generator().manual_seed(N + B)
    win = (torch.rand(N, B, nu, dtype=torch.float64, generator=g) * 2 - 1) * 3.0
    with torch.no_grad():
        ctx.set_option("gru_coop", 0)
        ctx.set_option("gru_gemm", 0)
        f64 = model.encode_actions(win.cuda()).cpu()
        ctx.set_option("gru_gemm", 1)
        i8 = model.encode_actions(win.cuda()).cpu()
    assert torch.isfinite(i8).all()
    np.testing.assert_allclose(i8.numpy(), f64.numpy(), rtol=0, atol=1e-12)
    ref = onl.gru_encoder(sd, (win[:512] - sd["action_mean"]) / sd["action_std"])
    np.testing.assert_allclose(i8[:512].numpy(), ref.numpy(), **TOL)


@pytest.mark.fp64_bit_identity
def test_planner_with_sliced_encoder(nlc):
    """The two-launch planner (K = 8192: GRU launch + split rollout) with either encoder: the same actions to 1e-9 over three
    commands; the option reaches the launch (stat), the FP64 planner is untouched by it."""
    import bench

    d, nu, A, T, K = 5, 1, 3.0, 12, 8192
    model = bench.synthetic_state_dict(d, nu, 17).to("cuda")

    def run(opts):
        p = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(bench.ENV), d, nlc.noise_sigma(nu), K, T, "cpu", lambda_=1.0,
                          u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox", seed=3,
                          U_init=torch.zeros(T, nu, dtype=torch.float64), planner_options=dict(opts, rollout_variant=2))
        st, ab = nlc.initial_state(bench.ENV, torch.Generator().manual_seed(0)), torch.zeros(4, nu, dtype=torch.float64)
        acts = []
        with torch.no_grad():
            for _ in range(3):
                act = p.command(st, ab)
                acts.append(act.clone())
                ab = torch.roll(ab, -1, 0)
                ab[-1] = act
        return torch.stack(acts), p.ctx.get_stat("gru_gemm"), p.cost_total.cpu().clone()

    a0, s0, c0 = run({})
    a1, s1, c1 = run({"gru_gemm": 1})
    assert (s0, s1) == (0, 1)
    np.testing.assert_allclose(a1.numpy(), a0.numpy(), rtol=0, atol=1e-9)
    np.testing.assert_allclose(c1.numpy(), c0.numpy(), rtol=1e-9, atol=1e-9)


def test_non_finite_weights_keep_the_fp64_encoder(nlc):
    """Fixed point has no NaN / infinity: a model with a non-finite GRU weight does not take the option -- the stat says so, and the
    launch is the FP64 kernel's (same bits with the option on and off)."""
    import oracle.nl_model as onl

    sd = onl.make_synthetic_state_dict(5, 5, 1, 128, 17)
    sd["action_encoder.gru.weight_hh_l1"] = sd["action_encoder.gru.weight_hh_l1"].clone()
    sd["action_encoder.gru.weight_hh_l1"][7, 3] = float("nan")
    model = build_model(nlc, sd)
    ctx = model.hip_ctx(torch.device("cuda:0"))
    win = torch.randn(64, 4, 1, dtype=torch.float64)
    with torch.no_grad():
        off = model.encode_actions(win.cuda()).cpu()
        ctx.set_option("gru_gemm", 1)
        assert ctx.get_stat("gru_gemm") == 0
        on = model.encode_actions(win.cuda()).cpu()
    assert torch.equal(torch.nan_to_num(on, nan=7.0), torch.nan_to_num(off, nan=7.0))


@pytest.mark.parametrize("env", ["cartpole", "pendulum"])
def test_sliced_encoder_with_time_channel_vs_reference_golden(nlc, env):
    """G5b with the int8-sliced encoder: an encode_obs_time model (GRU input nu + 1) -- forward() on explicit windows, and the
    two-launch planner, whose encode launch builds the harness closure's constant time channel itself (mppi_with_model.py:110-119)
    -- against the reference's outputs."""
    from gpu_common import check_command_steps

    g = np.load(f"{GOLD}/g5_nl_obs_time_{env}.npz")
    sd = load_sd(g)
    d, nu, K, T, A = int(g["d"]), int(g["nu"]), int(g["K"]), int(g["T"]), float(g["A"])
    m = nlc.NeuralLaplaceModel(
        d, nu, d, hidden_units=128, s_recon_terms=17, ilt_algorithm="fourier", encode_obs_time=True,
        state_mean=np.zeros(d), state_std=np.ones(d), action_mean=np.array([0] * nu), action_std=np.array([1.0]),
        normalize=True, normalize_time=True,
    ).double()
    m.load_state_dict(sd)
    m = m.cuda()
    ctx = m.hip_ctx(torch.device("cuda:0"))
    ctx.set_option("gru_gemm", 1)
    ctx.set_option("gru_coop", 0)
    with torch.no_grad():
        out = m(T64(g["fwd_obs"]).cuda(), T64(g["fwd_window"]).cuda(), T64(g["fwd_ts"]).cuda())
        np.testing.assert_allclose(out.cpu().numpy(), g["fwd_out"], **TOL)
        planners = []

        def make(U0):
            p = nlc.MPPIDelay(
                nlc.NLDynamics(m, 0.05), nlc.EnvCost("oderl-" + env), d, nlc.noise_sigma(nu), num_samples=K, horizon=T, device="cpu",
                lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, U_init=U0,
                planner_options={"gru_gemm": 1, "gru_coop": 0, "rollout_variant": 2},
            )
            planners.append(p)
            return p

        check_command_steps(nlc, g, make)
        assert planners[0].ctx.get_stat("gru_gemm") == 1 and planners[0].rollout_body != "fused"


def test_sliced_gemm_adversarial_rows_against_exact_rational_arithmetic(tmp_path):
    """12 288 outputs of sliced tiles on rows built to break a fixed-point product -- cancelling sums, a 2^+-40
    spread inside a row (large weights on small states), h = +-1 exactly with |w| on the row scale, every |h| < 2^-54, tiny states
    among ordinary ones -- against python `fractions` (exact).  The guarantee is absolute: |err| <= 2^-54 (sum |w| + s_m sum |h|) +
    5 x 2^-53 sum |w h| everywhere, never above 2^-48 of the row scale; for ordinary state magnitudes the FP64 chain's relative
    bound (5 x 2^-53 sum |w h|) holds too.  (A NaN / infinite window entry is the encoder's business: next test.)"""
    import sys

    sys.path.insert(0, os.path.join(REPO, "tools"))
    import i8_adversarial as adv

    dump = str(tmp_path / "adv.bin")
    adv.run_dump(dump)
    table = adv.check(adv.read_dump(dump))
    assert sum(r["outputs"] for r in table) >= 10000
    for fam, row in enumerate(table):
        assert row["outputs"] > 0, row
        assert row["max_of_bound"] <= 1.0, row
        assert row["max_abs_err_over_scale"] <= 2.0**-48, row
        if fam in adv.RELATIVE_FAMILIES:
            assert row["max_rel_units"] <= 5.0, row


def test_sliced_encoder_gives_nan_latents_for_a_non_finite_window(nlc):
    """fixed point has no NaN / infinity -- a window with a non-finite action must not come back as finite garbage (and
    the FP64 kernels' v_min / v_max clamps used to swallow a NaN too).  Every encoder form -- sliced, FP64 wave-sized, FP64
    cooperative -- marks such a window and returns NaN for both of its latents, as nn.GRU does for a NaN entry (w_nl.py:14-29);
    every other window of the launch is untouched."""
    import oracle.nl_model as onl

    sd = onl.make_synthetic_state_dict(3, 5, 1, 128, 17)
    model = build_model(nlc, sd)
    ctx = model.hip_ctx(torch.device("cuda:0"))
    g = torch.Generator().manual_seed(5)
    N, B = 4096 + 37, 4
    win = (torch.rand(N, B, 1, dtype=torch.float64, generator=g) * 2 - 1) * 3.0
    clean = win.clone()
    bad = {7: float("nan"), 64: float("inf"), 1000: float("-inf"), N - 1: float("nan")}
    for i, (w, v) in enumerate(bad.items()):
        win[w, i % B, 0] = v
    with torch.no_grad():
        ctx.set_option("gru_coop", 0)
        ctx.set_option("gru_gemm", 0)
        f64 = model.encode_actions(win.cuda()).cpu()
        f64_clean = model.encode_actions(clean.cuda()).cpu()
        ctx.set_option("gru_coop", 1)
        coop = model.encode_actions(win.cuda()).cpu()
        ctx.set_option("gru_coop", 0)
        ctx.set_option("gru_gemm", 1)
        i8 = model.encode_actions(win.cuda()).cpu()
        i8_clean = model.encode_actions(clean.cuda()).cpu()
    rows = torch.tensor(sorted(bad))
    keep = torch.ones(N, dtype=torch.bool)
    keep[rows] = False
    for got, ref in ((i8, i8_clean), (f64, f64_clean), (coop, f64_clean)):
        assert torch.isnan(got[rows]).all(), got[rows]
        assert torch.equal(got[keep], ref[keep]) and torch.isfinite(got[keep]).all()
    want = onl.gru_encoder(sd, (win[rows] - sd["action_mean"]) / sd["action_std"])  # the torch modules (= the reference's op sequence)
    assert torch.isnan(want[[0, 3]]).all(), "nn.GRU propagates a NaN entry"
