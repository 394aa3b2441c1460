#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference); the fixtures it writes are
data (inputs + expected outputs), never reference source.  Re-run:

    python tests/golden/make_golden.py

Fixtures:
  g1_mppi_<env>_d<delay>.npz  reference MPPIDelay + reference oracle.*_dynamics_dt_delay +
                              reference env reward methods           (pins a1-a4, a10-a12)
  g2_stages_<env>.npz         reference ReverseGRUEncoder / LaplaceRepresentationFunc I/O
                              (pins a7, a8 against real nn.GRU / the real module)
  g3_nl_<env>.npz             reference NeuralLaplaceModel.forward + reference MPPIDelay.command
                              with NL dynamics; ONLY laplace_reconstruct is the build's
                              restatement (torchlaplace is absent: parity unpinned for a9)
  g4_ilt_known.npz            analytic Laplace pairs + mpmath.invertlaplace(method='dehoog')
  g5_collector_<env>.npz      dataset-collector call pattern: reference MPPIDelay(encode_obs_time=True) with the
                              rolling time-stamp column in action_buffer (mppi_delay.py:261-287) + reference
                              oracle dynamics; records action_buffer before/after (the reference adds dt to the
                              caller's time column in place)
  g6_full_cfg2.npz            BASELINE configs[1] at FULL size: reference MPPIDelay + reference NeuralLaplaceModel,
                              K=16384, T=40, cartpole, two consecutive commands from torch.manual_seed(6) (the noise is
                              NOT stored: the test replays the seed, so the generator consumption order is pinned too);
                              stored: U, action, cost_total, omega, a strided subset of the states
  g7_full_<cfg>.npz           BASELINE configs[0] (cartpole, K=1024, T=20: the reference's own CPU-runnable case), configs[2] (pendulum, K=65536, T=40, 5-row action buffer) and configs[3]
                              (acrobot, K=262144, T=60) at FULL size, one reference command each, seed replay as in g6;
                              stored: U, action, beta/eta, and strided subsets of cost_total / omega / states / noise
  g8_cost_variants.npz        cartpole, reference MPPIDelay with the harness running_cost's non-default branches
                              (state_constraint / change_goal / change_goal_flipped, mppi_with_model.py:146-162) evaluated
                              by the REAL env class, NL and oracle dynamics, plus a terminal_state_cost
  g14_width_h<h>_<env>.npz    other hidden widths with the real reference classes: hidden_units 64 (class default, 33 terms)
                              and 256: GRU encoder, representation module, forward, two MPPIDelay commands
  g13_rollout_samples_<env>.npz  reference MPPIDelay(rollout_samples=3, rollout_var_cost=0.7, rollout_var_discount=0.9),
                              oracle dynamics, two commands (mppi_delay.py:291-292, 310)
  g5_nl_obs_time_<env>.npz    encode_obs_time NL model (GRU input nu+1) behind the harness closure that appends the
                              constant time channel B-1..0 (mppi_with_model.py:110-119) + reference MPPIDelay
"""

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.path.insert(0, REPO)

from oracle import envs as oenvs  # noqa: E402
from oracle import ilt as oilt  # noqa: E402
from oracle import nl_model as onl  # noqa: E402


def install_stubs():
    """Minimal stand-ins so the reference modules import without gym/torchdiffeq/torchlaplace."""
    tl = types.ModuleType("torchlaplace")
    tl.laplace_reconstruct = oilt.laplace_reconstruct  # the ONLY non-reference arithmetic in G3
    sys.modules["torchlaplace"] = tl

    gym = types.ModuleType("gym")

    class Env:
        pass

    gym.Env = Env
    spaces = types.ModuleType("gym.spaces")

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shape if shape else np.shape(low)).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=dtype), shape if shape else np.shape(high)).copy()
            self.shape = self.low.shape

    spaces.Box = Box
    gym.spaces = spaces
    gutils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")
    seeding.np_random = lambda seed=None: (np.random.RandomState(seed), seed)
    gutils.seeding = seeding
    gym.utils = gutils
    sys.modules.update({"gym": gym, "gym.spaces": spaces, "gym.utils": gutils, "gym.utils.seeding": seeding})
    tde = types.ModuleType("torchdiffeq")
    tde.odeint = None
    sys.modules["torchdiffeq"] = tde
    pk = types.ModuleType("TorchDiffEqPack")
    pko = types.ModuleType("TorchDiffEqPack.odesolver")
    pko.odesolve = None
    pk.odesolver = pko
    sys.modules.update({"TorchDiffEqPack": pk, "TorchDiffEqPack.odesolver": pko})


def load_reference_modules():
    """Import reference modules by file path so they do not clash with this repo's ``oracle`` package."""
    import importlib.util

    install_stubs()
    sys.argv = [sys.argv[0]]
    sys.path.insert(0, REF)

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    ref_oracle = load("ref_oracle", f"{REF}/oracle.py")
    import config as ref_config  # noqa: F401
    from planners.mppi_delay import MPPIDelay

    w_nl = load("ref_w_nl", f"{REF}/w_nl.py")
    from envs.oderl.envs import CTAcrobot, CTCartpole, CTPendulum

    envs = {
        "oderl-cartpole": lambda: CTCartpole(dt=0.05, obs_trans=True, device="cpu", solver="euler", friction=False),
        "oderl-pendulum": lambda: CTPendulum(dt=0.05, obs_trans=True, device="cpu", solver="euler"),
        "oderl-acrobot": lambda: CTAcrobot(dt=0.05, obs_trans=True, device="cpu", solver="euler"),
    }
    dyn = {
        "oderl-cartpole": ref_oracle.cartpole_dynamics_dt_delay,
        "oderl-pendulum": ref_oracle.pendulum_dynamics_dt_delay,
        "oderl-acrobot": ref_oracle.acrobot_dynamics_dt_delay,
    }
    return MPPIDelay, w_nl, envs, dyn


def noise_sigma(nu, sigma=1.0):
    # mppi_with_model.py:66-70
    g = sigma**2
    return torch.ones((nu, nu), dtype=torch.double) * 0.5 * g + torch.eye(nu, dtype=torch.double) * (g - 0.5 * g)


def np_(x):
    return x.detach().cpu().numpy().astype(np.float64)


def capture_command(mppi, state, action_buffer):
    """Run reference command() while recording U_before and the raw noise draw."""
    U_before = mppi.U.clone()
    rng = torch.random.get_rng_state()
    raw = mppi.noise_dist.sample((mppi.K, mppi.T))  # same draw command() will make
    torch.random.set_rng_state(rng)
    action = mppi.command(state, action_buffer)
    return dict(
        U_before=np_(U_before),
        noise_raw=np_(raw),
        action=np_(action),
        U_after=np_(mppi.U),
        cost_total=np_(mppi.cost_total),
        omega=np_(mppi.omega),
        noise=np_(mppi.noise),
        perturbed_action=np_(mppi.perturbed_action),
        states=np_(mppi.states),
        actions=np_(mppi.actions),
    )


def make_g1(MPPIDelay, envs, dyn):
    K, T, B = 64, 8, 4
    for env_name, mk in envs.items():
        env = mk()
        nx, nu, A = oenvs.OBS_DIM[env_name], oenvs.ACT_DIM[env_name], oenvs.ACTION_HIGH[env_name]
        assert float(env.action_space.high[0]) == A
        ts_pred = torch.full((K, 1), 0.05, dtype=torch.double)

        def running_cost(state, action, env=env):
            return -(env.diff_obs_reward_(state, exp_reward=False) + env.diff_ac_reward_(action))

        for delay in range(4):
            torch.manual_seed(100 + delay)
            from functools import partial

            dynamics = partial(dyn[env_name], ts=ts_pred, delay=delay, friction=False)
            mppi = MPPIDelay(
                dynamics,
                running_cost,
                nx,
                noise_sigma(nu),
                num_samples=K,
                horizon=T,
                device="cpu",
                lambda_=1.0,
                u_min=torch.tensor(-A),
                u_max=torch.tensor(A),
                u_scale=A,
            )
            state = oenvs.initial_state(env_name, seed=delay)
            action_buffer = (torch.rand(B, nu, dtype=torch.double) - 0.5) * A
            out = {}
            # two consecutive commands: the second exercises the U shift of a non-trivial U
            for step in range(2):
                c = capture_command(mppi, state.numpy(), action_buffer)
                for k, v in c.items():
                    out[f"s{step}_{k}"] = v
                out[f"s{step}_state"] = np_(state)
                out[f"s{step}_action_buffer"] = np_(action_buffer)
                state = mppi.states[0, 0].clone()  # any plausible next state
                action_buffer = torch.roll(action_buffer, -1, dims=0)
                action_buffer[-1] = torch.as_tensor(c["action"])
            np.savez_compressed(
                f"{HERE}/g1_mppi_{env_name.split('-')[1]}_d{delay}.npz",
                K=K, T=T, B=B, delay=delay, nx=nx, nu=nu, A=A, **out,
            )
            print("g1", env_name, delay, "action", out["s1_action"])


def build_ref_model(w_nl, env_name, seed, S=17, algo="fourier", h=128):
    st = onl.ENV_STATS[env_name]
    torch.manual_seed(seed)
    model = w_nl.NeuralLaplaceModel(
        st["d"],
        st["nu"],
        st["d"],
        hidden_units=h,
        s_recon_terms=S,
        ilt_algorithm=algo,
        encode_obs_time=False,
        state_mean=np.zeros(st["d"]),
        state_std=np.array(st["state_std"]),
        action_mean=np.array([0] * st["nu"]),
        action_std=np.array([st["act_high"] / 2.0]),
        normalize=True,
        normalize_time=True,
    ).double()
    return model


def make_g2_g3(MPPIDelay, w_nl, envs):
    for env_name, mk in envs.items():
        short = env_name.split("-")[1]
        st = onl.ENV_STATS[env_name]
        d, nu, A = st["d"], st["nu"], st["act_high"]
        S = 17
        model = build_ref_model(w_nl, env_name, seed=0)
        sd = {k: np_(v) for k, v in model.state_dict().items()}
        # the oracle's synthetic-weight builder must reproduce the reference constructor's weights
        mine = onl.make_synthetic_state_dict(
            0, d, nu, 128, S, state_std=st["state_std"], action_std=[A / 2.0]
        )
        for k in sd:
            assert np.array_equal(sd[k], np_(mine[k])), f"synthetic weights differ from reference ctor: {k}"
        torch.manual_seed(7)
        N, B = 48, 4
        with torch.no_grad():
            # ---- G2: stage I/O
            win = torch.randn(N, B, nu, dtype=torch.double)
            p_action = model.action_encoder(win)
            rep_in = torch.randn(N, 2 * S + d + 2, dtype=torch.double)
            theta, phi = model.laplace_rep_func(rep_in)
            np.savez_compressed(
                f"{HERE}/g2_stages_{short}.npz",
                d=d, nu=nu, S=S, h=128,
                gru_in=np_(win), gru_out=np_(p_action),
                rep_in=np_(rep_in), rep_theta=np_(theta), rep_phi=np_(phi),
                **{f"w::{k}": v for k, v in sd.items()},
            )
            # ---- G3: model forward + command with NL dynamics, on the "trained-like" tamed
            # weights (oracle/nl_model.py PHI_BIAS_SHIFT: the raw random model is chaotic)
            model.laplace_rep_func.linear_tanh_stack[4].bias[d * S :] += onl.PHI_BIAS_SHIFT
            sd = {k: np_(v) for k, v in model.state_dict().items()}
            obs = torch.randn(N, d, dtype=torch.double) * torch.tensor(st["state_std"])
            window = (torch.rand(N, B, nu, dtype=torch.double) * 2 - 1) * A
            ts = torch.full((N, 1), 0.05, dtype=torch.double)
            fwd = model(obs, window, ts)
            fwd33 = None
            out = dict(fwd_obs=np_(obs), fwd_window=np_(window), fwd_ts=np_(ts), fwd_out=np_(fwd))
            K, T = 64, 8
            ts_pred = torch.full((K, 1), 0.05, dtype=torch.double)
            env = mk()

            def dynamics(state, perturbed_action):
                return state + model(state, perturbed_action, ts_pred)

            def running_cost(state, action, env=env):
                return -(env.diff_obs_reward_(state, exp_reward=False) + env.diff_ac_reward_(action))

            torch.manual_seed(11)
            mppi = MPPIDelay(
                dynamics, running_cost, d, noise_sigma(nu), num_samples=K, horizon=T, device="cpu",
                lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A,
            )
            state = oenvs.initial_state(env_name, seed=3)
            action_buffer = torch.zeros(B, nu, dtype=torch.double)
            for step in range(2):
                c = capture_command(mppi, state.numpy(), action_buffer)
                for k, v in c.items():
                    out[f"s{step}_{k}"] = v
                out[f"s{step}_state"] = np_(state)
                out[f"s{step}_action_buffer"] = np_(action_buffer)
                state = mppi.states[0, 0].clone()
                action_buffer = torch.roll(action_buffer, -1, dims=0)
                action_buffer[-1] = torch.as_tensor(c["action"])
            # de Hoog / S=33 variant of the forward (cfg5)
            model33 = build_ref_model(w_nl, env_name, seed=1, S=33, algo="dehoog")
            fwd33 = model33(obs, window, ts)
            sd33 = {k: np_(v) for k, v in model33.state_dict().items()}
            np.savez_compressed(
                f"{HERE}/g3_nl_{short}.npz",
                d=d, nu=nu, S=S, h=128, K=K, T=T, B=B, A=A,
                fwd33_out=np_(fwd33),
                **out,
                **{f"w::{k}": v for k, v in sd.items()},
                **{f"w33::{k}": v for k, v in sd33.items()},
            )
            print("g3", env_name, "action", out["s1_action"], "fwd[0]", out["fwd_out"][0])


def make_g5(MPPIDelay, w_nl, envs, dyn):
    from functools import partial

    K, T, B, dt = 64, 8, 4, 0.05
    for env_name, delay in (("oderl-cartpole", 2), ("oderl-acrobot", 1), ("oderl-pendulum", 0)):
        short = env_name.split("-")[1]
        env = envs[env_name]()
        nx, nu, A = oenvs.OBS_DIM[env_name], oenvs.ACT_DIM[env_name], oenvs.ACTION_HIGH[env_name]
        ts_pred = torch.full((K, 1), dt, dtype=torch.double)

        def running_cost(state, action, env=env):
            return -(env.diff_obs_reward_(state, exp_reward=False) + env.diff_ac_reward_(action))

        # ---- (a) collector: planner-side rolling time stamps, oracle dynamics (mppi_dataset_collector.py:166-180,228-231)
        torch.manual_seed(300 + delay)
        mppi = MPPIDelay(
            partial(dyn[env_name], ts=ts_pred, delay=delay, friction=False), running_cost, nx, noise_sigma(nu),
            num_samples=K, horizon=T, device="cpu", lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A),
            u_scale=A, encode_obs_time=True, dt=dt,
        )
        state = oenvs.initial_state(env_name, seed=5)
        action_buffer = torch.zeros((B, nu + 1), dtype=torch.double)
        action_buffer[:, nu:] = (torch.flip(torch.arange(4), (0,)) * dt).view(-1, 1)
        action_buffer[:, :nu] = (torch.rand(B, nu, dtype=torch.double) - 0.5) * A
        out = {}
        for step in range(2):
            out[f"s{step}_action_buffer"] = np_(action_buffer)
            c = capture_command(mppi, state.numpy(), action_buffer)
            out[f"s{step}_action_buffer_after"] = np_(action_buffer)
            for k, v in c.items():
                out[f"s{step}_{k}"] = v
            out[f"s{step}_state"] = np_(state)
            state = mppi.states[0, 0].clone()
            # get_action_with_encode_obs_time-like hand-over: roll, append [action, 0], age the time stamps
            action_buffer = torch.roll(action_buffer, -1, dims=0)
            action_buffer[-1, :nu] = torch.as_tensor(c["action"])
            action_buffer[:, nu:] += dt
            action_buffer[-1, nu:] = 0
        np.savez_compressed(f"{HERE}/g5_collector_{short}.npz", K=K, T=T, B=B, delay=delay, nx=nx, nu=nu, A=A, dt=dt, **out)
        print("g5 collector", env_name, "action", out["s1_action"])

        # ---- (b) encode_obs_time NL model behind the harness closure (mppi_with_model.py:103-122)
        if nu != 1:
            continue  # the reference itself cannot run it: action_mean = [0]*nu does not broadcast over nu+1 channels
        st = onl.ENV_STATS[env_name]
        torch.manual_seed(2)
        model = w_nl.NeuralLaplaceModel(
            nx, nu, nx, hidden_units=128, s_recon_terms=17, ilt_algorithm="fourier", encode_obs_time=True,
            state_mean=np.zeros(nx), state_std=np.array(st["state_std"]), action_mean=np.array([0] * nu),
            action_std=np.array([A / 2.0]), normalize=True, normalize_time=True,
        ).double()
        with torch.no_grad():
            model.laplace_rep_func.linear_tanh_stack[4].bias[nx * 17 :] += onl.PHI_BIAS_SHIFT
            sd = {k: np_(v) for k, v in model.state_dict().items()}

            def dynamics(state, perturbed_action, action_buffer_size=B):
                perturbed_action = torch.cat(
                    (
                        perturbed_action,
                        torch.flip(torch.arange(action_buffer_size), (0,)).view(1, action_buffer_size, 1)
                        .repeat(perturbed_action.shape[0], 1, 1),
                    ),
                    dim=2,
                )
                return state + model(state, perturbed_action, ts_pred)

            torch.manual_seed(13)
            mppi = MPPIDelay(
                dynamics, running_cost, nx, noise_sigma(nu), num_samples=K, horizon=T, device="cpu", lambda_=1.0,
                u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A,
            )
            state = oenvs.initial_state(env_name, seed=4)
            action_buffer = (torch.rand(B, nu, dtype=torch.double) - 0.5) * A
            out = {}
            N = 40
            obs = torch.randn(N, nx, dtype=torch.double) * torch.tensor(st["state_std"])
            window = torch.cat(((torch.rand(N, B, nu, dtype=torch.double) * 2 - 1) * A,
                                torch.rand(N, B, 1, dtype=torch.double) * 0.2), dim=2)
            ts = torch.full((N, 1), dt, dtype=torch.double)
            out.update(fwd_obs=np_(obs), fwd_window=np_(window), fwd_ts=np_(ts), fwd_out=np_(model(obs, window, ts)))
            for step in range(2):
                c = capture_command(mppi, state.numpy(), action_buffer)
                for k, v in c.items():
                    out[f"s{step}_{k}"] = v
                out[f"s{step}_state"] = np_(state)
                out[f"s{step}_action_buffer"] = np_(action_buffer)
                state = mppi.states[0, 0].clone()
                action_buffer = torch.roll(action_buffer, -1, dims=0)
                action_buffer[-1] = torch.as_tensor(c["action"])
        np.savez_compressed(f"{HERE}/g5_nl_obs_time_{short}.npz", K=K, T=T, B=B, nx=nx, nu=nu, A=A, S=17, h=128, d=nx,
                            **out, **{f"w::{k}": v for k, v in sd.items()})
        print("g5 nl", env_name, "action", out["s1_action"])


def make_g6(MPPIDelay, w_nl, envs):
    env_name, K, T, B = "oderl-cartpole", 16384, 40, 4
    st = onl.ENV_STATS[env_name]
    d, nu, A = st["d"], st["nu"], st["act_high"]
    model = build_ref_model(w_nl, env_name, seed=0)
    env = envs[env_name]()
    ts_pred = torch.full((K, 1), 0.05, dtype=torch.double)
    with torch.no_grad():
        model.laplace_rep_func.linear_tanh_stack[4].bias[d * 17 :] += onl.PHI_BIAS_SHIFT

        def dynamics(state, perturbed_action):
            return state + model(state, perturbed_action, ts_pred)

        def running_cost(state, action):
            return -(env.diff_obs_reward_(state, exp_reward=False) + env.diff_ac_reward_(action))

        torch.manual_seed(6)
        mppi = MPPIDelay(
            dynamics, running_cost, d, noise_sigma(nu), num_samples=K, horizon=T, device="cpu", lambda_=1.0,
            u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A,
        )
        state = oenvs.initial_state(env_name, seed=0)
        action_buffer = torch.zeros(B, nu, dtype=torch.double)
        out = dict(U0=np_(mppi.U))
        sub = np.arange(0, K, 128)
        for step in range(2):
            action = mppi.command(state.numpy(), action_buffer)
            out[f"s{step}_state"] = np_(state)
            out[f"s{step}_action_buffer"] = np_(action_buffer)
            out[f"s{step}_action"] = np_(action)
            out[f"s{step}_U_after"] = np_(mppi.U)
            out[f"s{step}_cost_total"] = np_(mppi.cost_total)
            out[f"s{step}_omega"] = np_(mppi.omega)
            out[f"s{step}_states_sub"] = np_(mppi.states)[sub]
            out[f"s{step}_noise_sub"] = np_(mppi.noise)[sub]
            state = mppi.states[0, 0].clone()
            action_buffer = torch.roll(action_buffer, -1, dims=0)
            action_buffer[-1] = action
            print("g6 step", step, "action", out[f"s{step}_action"])
    np.savez_compressed(f"{HERE}/g6_full_cfg2.npz", K=K, T=T, B=B, d=d, nu=nu, A=A, seed=6, sub=sub, **out)


def make_g7(MPPIDelay, w_nl, envs, only=None):
    for tag, env_name, K, T, B, seed in (("cfg1", "oderl-cartpole", 1024, 20, 4, 9), ("cfg3", "oderl-pendulum", 65536, 40, 5, 7),
                                         ("cfg4", "oderl-acrobot", 262144, 60, 4, 8)):
        if only is not None and tag not in only:
            continue
        st = onl.ENV_STATS[env_name]
        d, nu, A = st["d"], st["nu"], st["act_high"]
        model = build_ref_model(w_nl, env_name, seed=0)
        env = envs[env_name]()
        ts_pred = torch.full((K, 1), 0.05, dtype=torch.double)
        with torch.no_grad():
            model.laplace_rep_func.linear_tanh_stack[4].bias[d * 17 :] += onl.PHI_BIAS_SHIFT

            def dynamics(state, perturbed_action):
                return state + model(state, perturbed_action, ts_pred)

            def running_cost(state, action):
                return -(env.diff_obs_reward_(state, exp_reward=False) + env.diff_ac_reward_(action))

            torch.manual_seed(seed)
            mppi = MPPIDelay(
                dynamics, running_cost, d, noise_sigma(nu), num_samples=K, horizon=T, device="cpu", lambda_=1.0,
                u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A,
            )
            U0 = np_(mppi.U)
            state = oenvs.initial_state(env_name, seed=1)
            action_buffer = (torch.rand(B, nu, dtype=torch.double) - 0.5) * A
            action = mppi.command(state.numpy(), action_buffer)
            sub = np.arange(0, K, K // 128)
            cost = mppi.cost_total
            np.savez_compressed(
                f"{HERE}/g7_full_{tag}.npz", K=K, T=T, B=B, d=d, nu=nu, A=A, seed=seed, sub=sub, U0=U0,
                state=np_(state), action_buffer=np_(action_buffer), action=np_(action), U_after=np_(mppi.U),
                beta=float(cost.min()), eta=float(mppi.cost_total_non_zero.sum()), cost_sum=float(cost.sum()),
                cost_total_sub=np_(cost)[sub], omega_sub=np_(mppi.omega)[sub], states_sub=np_(mppi.states)[sub],
                noise_sub=np_(mppi.noise)[sub],
            )
            print("g7", tag, "action", np_(action))


def make_g8(MPPIDelay, w_nl, envs, dyn):
    from functools import partial

    env_name, K, T, B = "oderl-cartpole", 64, 8, 4
    st = onl.ENV_STATS[env_name]
    d, nu, A = st["d"], st["nu"], st["act_high"]
    env = envs[env_name]()
    model = build_ref_model(w_nl, env_name, seed=3)
    ts_pred = torch.full((K, 1), 0.05, dtype=torch.double)
    out = {}
    with torch.no_grad():
        model.laplace_rep_func.linear_tanh_stack[4].bias[d * 17 :] += onl.PHI_BIAS_SHIFT
        out.update({f"w::{k}": np_(v) for k, v in model.state_dict().items()})

        def nl_dynamics(state, perturbed_action):
            return state + model(state, perturbed_action, ts_pred)

        variants = {
            "constraint": dict(state_constraint=True),
            "goal": dict(change_goal=True, change_goal_flipped=False),
            "goal_flipped": dict(change_goal=True, change_goal_flipped=True),
        }
        for vname, kw in variants.items():
            def running_cost(state, action, kw=kw):
                return -(env.diff_obs_reward_(state, exp_reward=False, **kw) + env.diff_ac_reward_(action))

            for dname, dynamics in (("nl", nl_dynamics),
                                    ("oracle", partial(dyn[env_name], ts=ts_pred, delay=1, friction=False))):
                torch.manual_seed(40)
                terminal = (lambda states, actions: 0.5 * (states[..., -1, 0] ** 2).reshape(-1)) if vname == "goal" else None
                mppi = MPPIDelay(
                    dynamics, running_cost, d, noise_sigma(nu), num_samples=K, horizon=T, device="cpu", lambda_=1.0,
                    u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, terminal_state_cost=terminal,
                )
                state = oenvs.initial_state(env_name, seed=2)
                action_buffer = (torch.rand(B, nu, dtype=torch.double) - 0.5) * A
                for step in range(2):
                    c = capture_command(mppi, state.numpy(), action_buffer)
                    pre = f"{vname}_{dname}_s{step}_"
                    for k, v in c.items():
                        out[pre + k] = v
                    out[pre + "state"] = np_(state)
                    out[pre + "action_buffer"] = np_(action_buffer)
                    state = mppi.states[0, 0].clone()
                    action_buffer = torch.roll(action_buffer, -1, dims=0)
                    action_buffer[-1] = torch.as_tensor(c["action"])
                print("g8", vname, dname, "action", out[pre + "action"])
    np.savez_compressed(f"{HERE}/g8_cost_variants.npz", K=K, T=T, B=B, d=d, nu=nu, A=A, **out)


def make_g4():
    """ILT known answers: analytic pairs + mpmath de Hoog (degree 16 -> 33 terms)."""
    import mpmath as mp

    mp.mp.dps = 30
    ts = [0.125, 0.5, 1.0, 2.5]
    pairs = {
        "exp_decay": (lambda s: 1 / (s + 1), lambda t: mp.e ** (-t)),
        "cosine": (lambda s: s / (s * s + 4), lambda t: mp.cos(2 * t)),
        "sine_damped": (lambda s: 3 / ((s + mp.mpf("0.5")) ** 2 + 9), lambda t: mp.e ** (-t / 2) * mp.sin(3 * t)),
        "ramp": (lambda s: 1 / (s * s), lambda t: t),
        "delayed_step": (lambda s: mp.e ** (-mp.mpf("0.3") * s) / s, lambda t: mp.mpf(1) if t > 0.3 else mp.mpf(0)),
    }
    out = {"ts": np.array(ts)}
    for name, (F, f) in pairs.items():
        out[f"{name}_exact"] = np.array([float(f(mp.mpf(t))) for t in ts])
        out[f"{name}_mp_dehoog"] = np.array(
            # same abscissa parameters as the restated torchlaplace defaults (alpha=1e-10, tol=1e-9)
            [float(mp.invertlaplace(F, t, method="dehoog", degree=16, alpha=1e-10, tol=1e-9)) for t in ts]
        )
        # F sampled at the query points the oracle/HIP path use (both algorithms), so the
        # device test needs no mpmath: (len(ts), S) real/imag
        for algo, S in (("fourier", 17), ("fourier", 33), ("dehoog", 33), ("dehoog", 17)):
            alpha, tol, scale = oilt.ilt_options(algo)
            sr, si, _, _ = oilt.query_points(torch.tensor(ts), S, alpha, tol, scale)
            vals = [[complex(F(mp.mpc(float(sr[i, k]), float(si[i, k])))) for k in range(S)] for i in range(len(ts))]
            out[f"{name}_{algo}{S}_Fre"] = np.array([[v.real for v in row] for row in vals])
            out[f"{name}_{algo}{S}_Fim"] = np.array([[v.imag for v in row] for row in vals])
    np.savez_compressed(f"{HERE}/g4_ilt_known.npz", **out)
    print("g4 done")


def make_g14(MPPIDelay, w_nl, envs):
    """Other hidden widths with the REAL reference classes: hidden_units = 64 with the class defaults' 33 terms
    (w_nl.py:72-73) on pendulum, hidden_units = 256 / 17 terms on acrobot: GRU encoder (real nn.GRU, hidden 32 / 128),
    representation module, model.forward and two commands of the real MPPIDelay behind the harness closure (only
    laplace_reconstruct is the build's restatement, as in G3)."""
    for env_name, h, S in (("oderl-pendulum", 64, 33), ("oderl-acrobot", 256, 17)):
        st = onl.ENV_STATS[env_name]
        d, nu, A = st["d"], st["nu"], st["act_high"]
        model = build_ref_model(w_nl, env_name, seed=14, S=S, h=h)
        mine = onl.make_synthetic_state_dict(14, d, nu, h, S, state_std=st["state_std"], action_std=[A / 2.0])
        for k, v in model.state_dict().items():
            assert np.array_equal(np_(v), np_(mine[k])), f"synthetic weights differ from reference ctor: {k}"
        torch.manual_seed(15)
        N, B, K, T = 40, 4, 64, 8
        out = {}
        with torch.no_grad():
            win = torch.randn(N, B, nu, dtype=torch.double)
            rep_in = torch.randn(N, 2 * S + d + 2, dtype=torch.double)
            theta, phi = model.laplace_rep_func(rep_in)
            out.update(gru_in=np_(win), gru_out=np_(model.action_encoder(win)), rep_in=np_(rep_in), rep_theta=np_(theta),
                       rep_phi=np_(phi))
            model.laplace_rep_func.linear_tanh_stack[4].bias[d * S :] += onl.PHI_BIAS_SHIFT
            sd = {k: np_(v) for k, v in model.state_dict().items()}
            obs = torch.randn(N, d, dtype=torch.double) * torch.tensor(st["state_std"])
            window = (torch.rand(N, B, nu, dtype=torch.double) * 2 - 1) * A
            ts = torch.full((N, 1), 0.05, dtype=torch.double)
            out.update(fwd_obs=np_(obs), fwd_window=np_(window), fwd_ts=np_(ts), fwd_out=np_(model(obs, window, ts)))
            ts_pred = torch.full((K, 1), 0.05, dtype=torch.double)
            env = envs[env_name]()

            def dynamics(state, perturbed_action):
                return state + model(state, perturbed_action, ts_pred)

            def running_cost(state, action, env=env):
                return -(env.diff_obs_reward_(state, exp_reward=False) + env.diff_ac_reward_(action))

            torch.manual_seed(16)
            mppi = MPPIDelay(dynamics, running_cost, d, noise_sigma(nu), num_samples=K, horizon=T, device="cpu", lambda_=1.0,
                             u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A)
            state = oenvs.initial_state(env_name, seed=4)
            action_buffer = torch.zeros(B, nu, dtype=torch.double)
            for step in range(2):
                c = capture_command(mppi, state.numpy(), action_buffer)
                for k, v in c.items():
                    out[f"s{step}_{k}"] = v
                out[f"s{step}_state"] = np_(state)
                out[f"s{step}_action_buffer"] = np_(action_buffer)
                state = mppi.states[0, 0].clone()
                action_buffer = torch.roll(action_buffer, -1, dims=0)
                action_buffer[-1] = torch.as_tensor(c["action"])
        np.savez_compressed(f"{HERE}/g14_width_h{h}_{env_name.split('-')[1]}.npz", d=d, nu=nu, S=S, h=h, K=K, T=T, B=B, A=A,
                            **out, **{f"w::{k}": v for k, v in sd.items()})
        print("g14", env_name, h, "action", out["s1_action"])


def make_g13(MPPIDelay, envs, dyn):
    """rollout_samples M > 1 with a rollout_var_cost (mppi_delay.py:291-292, 310): pendulum and acrobot, oracle dynamics,
    two consecutive commands.  (The reference does NOT replicate the state M times, so the variance it adds is the
    variance of the running cost OVER THE K SAMPLES -- one number per horizon step, the same for every sample.)"""
    from functools import partial

    K, T, B = 96, 7, 4
    for env_name in ("oderl-pendulum", "oderl-acrobot"):
        env = envs[env_name]()
        nx, nu, A = oenvs.OBS_DIM[env_name], oenvs.ACT_DIM[env_name], oenvs.ACTION_HIGH[env_name]

        def running_cost(state, action, env=env):
            return -(env.diff_obs_reward_(state, exp_reward=False) + env.diff_ac_reward_(action))

        torch.manual_seed(1300)
        dynamics = partial(dyn[env_name], ts=torch.full((K, 1), 0.05, dtype=torch.double), delay=1, friction=False)
        mppi = MPPIDelay(dynamics, running_cost, nx, noise_sigma(nu), num_samples=K, horizon=T, device="cpu", lambda_=1.0,
                         u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, rollout_samples=3, rollout_var_cost=0.7,
                         rollout_var_discount=0.9)
        state = oenvs.initial_state(env_name, seed=5)
        action_buffer = (torch.rand(B, nu, dtype=torch.double) - 0.5) * A
        out = {}
        for step in range(2):
            c = capture_command(mppi, state.numpy(), action_buffer)
            for k, v in c.items():
                out[f"s{step}_{k}"] = v
            out[f"s{step}_state"] = np_(state)
            out[f"s{step}_action_buffer"] = np_(action_buffer)
            state = mppi.states[0, 0].clone()
            action_buffer = torch.roll(action_buffer, -1, dims=0)
            action_buffer[-1] = torch.as_tensor(c["action"])
        np.savez_compressed(f"{HERE}/g13_rollout_samples_{env_name.split('-')[1]}.npz", K=K, T=T, B=B, delay=1, nx=nx, nu=nu,
                            A=A, M=3, var_cost=0.7, var_discount=0.9, **out)
        print("g13", env_name, "action", out["s1_action"])


def main():
    MPPIDelay, w_nl, envs, dyn = load_reference_modules()
    if os.environ.get("NLC_GOLDEN_ONLY") == "g13":
        make_g13(MPPIDelay, envs, dyn)
        return
    if os.environ.get("NLC_GOLDEN_ONLY") == "g14":
        make_g14(MPPIDelay, w_nl, envs)
        return
    make_g1(MPPIDelay, envs, dyn)
    make_g2_g3(MPPIDelay, w_nl, envs)
    make_g4()
    make_g5(MPPIDelay, w_nl, envs, dyn)
    make_g6(MPPIDelay, w_nl, envs)
    make_g8(MPPIDelay, w_nl, envs, dyn)
    make_g7(MPPIDelay, w_nl, envs, only=os.environ.get("NLC_G7_ONLY", "cfg1,cfg3,cfg4").split(","))
    make_g13(MPPIDelay, envs, dyn)
    make_g14(MPPIDelay, w_nl, envs)


if __name__ == "__main__":
    main()
