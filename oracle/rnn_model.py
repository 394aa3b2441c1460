"""Oracle: the Delta-t RNN baseline dynamics model.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.

Restates ``DeltaTRNN`` (``train_utils.py:589-631``; factory ``:56-74``, ``rnn_hidden_units=160``
``config.py:43``) behind the harness dynamics closure ``mppi_with_model.py:103-122``
(``state + model(state, window, ts_pred)``; the closure appends the time channel only for
``model_name == "nl"``, so this model always sees the plain (K, B, nu) action window):

    out, _ = GRU(action_dim -> H, 1 layer, batch_first)(window_n)        # forward order, h0 = 0
    dx     = Linear(H + d + 1 -> d)(cat(out[:, -1], obs_n, ts_n))

Normalisation quirk reproduced on purpose (``train_utils.py:618-626``): the ``else`` that selects the raw
inputs belongs to ``if self.normalize_time``, not to ``if self.normalize``:
  normalize and normalize_time      -> obs/action standardised, ts / (dt * 8)
  normalize_time False (any normalize) -> obs RAW, action / 3, ts RAW
  normalize False, normalize_time True -> the reference raises (batch_obs undefined)

Weights travel as a dict with the reference's ``state_dict`` keys (``gru.weight_ih_l0`` ...,
``linear_out.weight/bias``, buffers ``state_mean/state_std/action_mean/action_std/dt``), float64.
"""

import numpy as np
import torch
import torch.nn as nn


def make_synthetic_state_dict(seed=0, d=5, nu=1, hidden=160, state_std=None, action_std=None, dt=0.05,
                              out_scale=0.2, time_input=True):
    """Seeded synthetic weights in the reference constructor's order (GRU, then linear_out: train_utils.py:610-613).

    ``out_scale`` shrinks ``linear_out`` ("trained-like": a model that predicts small state differences), so a
    T = 40 rollout stays in the envs' state range; parity does not depend on it.
    """
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    gru = nn.GRU(nu, hidden, batch_first=True)
    lin = nn.Linear(hidden + d + (1 if time_input else 0), d)  # the plain RNN baseline has no time input
    torch.random.set_rng_state(gen_state)
    sd = {f"gru.{k}": v for k, v in gru.state_dict().items()}
    sd["linear_out.weight"] = lin.weight
    sd["linear_out.bias"] = lin.bias
    sd = {k: v.detach().to(torch.float64).clone() for k, v in sd.items()}
    sd["linear_out.weight"] *= out_scale  # scaled after model.double(), as the fixture generator does
    sd["linear_out.bias"] *= out_scale
    sd["state_mean"] = torch.zeros(d, dtype=torch.float64)
    sd["state_std"] = torch.as_tensor(state_std if state_std is not None else np.ones(d), dtype=torch.float64)
    sd["action_mean"] = torch.zeros(1, dtype=torch.float64)
    sd["action_std"] = torch.as_tensor(action_std if action_std is not None else [1.0], dtype=torch.float64)
    # the reference registers torch.tensor(0.05) = float32; model.double() widens it (0.05000000074505806)
    sd["dt"] = torch.tensor(dt, dtype=torch.float32).to(torch.float64)
    return sd


def gru_forward_last(sd, x):
    """Last output of a 1-layer batch_first nn.GRU with h0 = 0, explicit gate equations (PyTorch docs):
    r = s(W_ir x + b_ir + W_hr h + b_hr), z likewise, n = tanh(W_in x + b_in + r (W_hn h + b_hn)),
    h' = (1 - z) n + z h.  x: (N, L, nin)."""
    Wih, Whh = sd["gru.weight_ih_l0"], sd["gru.weight_hh_l0"]
    bih, bhh = sd["gru.bias_ih_l0"], sd["gru.bias_hh_l0"]
    H = Whh.shape[1]
    h = torch.zeros(x.shape[0], H, dtype=torch.float64)
    for s in range(x.shape[1]):
        gi = x[:, s, :] @ Wih.T + bih
        gh = h @ Whh.T + bhh
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H : 2 * H] + gh[:, H : 2 * H])
        n = torch.tanh(gi[:, 2 * H :] + r * gh[:, 2 * H :])
        h = (1.0 - z) * n + z * h
    return h


def forward(sd, obs, window, ts_pred, normalize=True, normalize_time=True):
    """``DeltaTRNN.forward`` (train_utils.py:618-631): obs (N, d), window (N, B, nin), ts_pred (N, 1) -> (N, d)."""
    obs = obs.to(torch.float64)
    window = window.to(torch.float64)
    ts = ts_pred.to(torch.float64)
    if normalize:
        batch_obs = (obs - sd["state_mean"]) / sd["state_std"]
        batch_action = (window - sd["action_mean"]) / sd["action_std"]
    if normalize_time:
        if not normalize:
            raise NameError("the reference leaves batch_obs undefined for normalize=False, normalize_time=True")
        ts = ts / (sd["dt"] * 8.0)
    else:
        batch_obs = obs
        batch_action = window / 3.0
    h = gru_forward_last(sd, batch_action)
    feat = torch.cat((h, batch_obs, ts), dim=1)
    return feat @ sd["linear_out.weight"].T + sd["linear_out.bias"]


def forward_rnn(sd, obs, window, normalize=True):
    """``RNN.forward`` (train_utils.py:577-586): the plain baseline -- no time input, ``linear_out`` is (d, H+d), the
    raw-input branch belongs to ``normalize``."""
    obs, window = obs.to(torch.float64), window.to(torch.float64)
    if normalize:
        batch_obs = (obs - sd["state_mean"]) / sd["state_std"]
        batch_action = (window - sd["action_mean"]) / sd["action_std"]
    else:
        batch_obs = obs
        batch_action = window / 3.0
    h = gru_forward_last(sd, batch_action)
    return torch.cat((h, batch_obs), dim=1) @ sd["linear_out.weight"].T + sd["linear_out.bias"]


def make_dynamics_rnn(sd, normalize=True):
    def dynamics(state, window):
        return state + forward_rnn(sd, state, window, normalize)

    return dynamics


def make_dynamics(sd, dt=0.05, normalize=True, normalize_time=True):
    """Harness closure (mppi_with_model.py:103-122): ts_pred = tensor(dt).view(1,1).repeat(K,1) (:74)."""

    def dynamics(state, window):
        ts = torch.full((state.shape[0], 1), dt, dtype=torch.float64)
        return state + forward(sd, state, window, ts, normalize, normalize_time)

    return dynamics
