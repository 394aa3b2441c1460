"""MI355X twin of the reference's Delta-t RNN baseline, ``train_utils.DeltaTRNN`` (``train_utils.py:589-631``;
factory ``get_delta_t_rnn_model`` ``:56-74``, ``rnn_hidden_units=160`` ``config.py:43``): same constructor
arguments, sub-module names and ``state_dict`` keys (``gru.*``, ``linear_out.*``, buffers ``state_mean state_std
action_mean action_std dt``), so checkpoints written by the reference load unchanged.

The plain ``RNN`` baseline (``train_utils.py:550-586``: the same GRU, ``linear_out`` over ``[h | obs]``, no time input,
the un-normalised branch tied to ``normalize``) shares the kernels (``nlc_rnn_desc.time_input = 0``).

``forward`` runs as two HIP launches behind ``nlc_rnn_forward`` (GRU on FP64 matrix cores + hidden part of
``linear_out``; then the state/time part); behind ``NLDynamics`` the planner hoists the GRU out of the horizon loop
(``NLC_DYN_DTRNN``).  The HIP path is inference-only and float64, as the harness uses it
(``mppi_with_model.py:101,319``); in grad mode ``forward`` is the same op sequence on PyTorch-ROCm (trainable).
"""

import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from ._weights import WeightsKeyMixin
from .laplace import compute_device

_BLOB_KEYS = ["gru.weight_ih_l0", "gru.weight_hh_l0", "gru.bias_ih_l0", "gru.bias_hh_l0", "linear_out.weight",
              "linear_out.bias"]


class DeltaTRNN(WeightsKeyMixin, nn.Module):
    _dyn_id = _lib.DYN_DTRNN  # rollout the fused planner selects for NLDynamics(model, dt)

    def __init__(
        self,
        state_dim,
        action_dim,
        hidden_units=64,
        encode_obs_time=False,
        state_mean=None,
        state_std=None,
        action_mean=None,
        action_std=None,
        normalize=False,
        normalize_time=False,
        dt=0.05,
    ):
        super().__init__()
        dimension_in = action_dim + (1 if encode_obs_time else 0)
        self.encode_obs_time = encode_obs_time
        self.state_dim, self.action_dim, self.hidden_units = state_dim, action_dim, hidden_units
        self.gru = nn.GRU(dimension_in, hidden_units, batch_first=True)
        self.linear_out = nn.Linear(hidden_units + state_dim + 1, state_dim)  # + 1: delta t
        self.normalize = normalize
        self.normalize_time = normalize_time
        # same dtypes as the reference (train_utils.py:613-617): dt is float32, action_mean int64
        self.register_buffer("state_mean", torch.tensor(state_mean))
        self.register_buffer("state_std", torch.tensor(state_std))
        self.register_buffer("action_mean", torch.tensor(action_mean))
        self.register_buffer("action_std", torch.tensor(action_std))
        self.register_buffer("dt", torch.tensor(dt))
        self._ctx = None
        self._uploaded_key = None

    @classmethod
    def from_reference(cls, ref):
        """Twin of a loaded reference ``DeltaTRNN`` (same hyper-parameters, buffers and weights, on its device)."""
        first = next(ref.parameters())
        d = ref.linear_out.out_features
        enc = bool(ref.encode_obs_time)
        m = cls(
            d, ref.gru.input_size - int(enc), hidden_units=ref.gru.hidden_size, encode_obs_time=enc,
            state_mean=[0.0] * d, state_std=[1.0] * d, action_mean=[0], action_std=[1.0],
            normalize=ref.normalize, normalize_time=ref.normalize_time,
        ).to(device=first.device, dtype=first.dtype)
        for name in ("state_mean", "state_std", "action_mean", "action_std", "dt"):
            m.register_buffer(name, getattr(ref, name).detach().clone())
        m.load_state_dict(ref.state_dict())
        m.train(ref.training)
        return m

    # ------------------------------------------------------------------ HIP plumbing
    def _weights_key_extra(self):
        return (self.normalize, self.normalize_time)

    def model_desc(self):
        """Resolve the reference's branch structure (train_utils.py:618-626): the raw-input ``else`` belongs to
        ``if self.normalize_time``; normalize=False with normalize_time=True leaves ``batch_obs`` undefined there."""
        d, nin = self.state_dim, self.action_dim + (1 if self.encode_obs_time else 0)
        desc = _lib.RnnDesc()
        desc.d, desc.nin, desc.hidden, desc.time_input = d, nin, self.hidden_units, 1
        f64 = lambda t: t.detach().to("cpu", torch.float64).reshape(-1)  # noqa: E731
        if self.normalize_time:
            if not self.normalize:
                raise NameError("DeltaTRNN(normalize=False, normalize_time=True): the reference's forward fails "
                                "(batch_obs is undefined, train_utils.py:618-631)")
            sm, ss = f64(self.state_mean), f64(self.state_std)
            am = f64(self.action_mean).expand(nin) if self.action_mean.numel() == 1 else f64(self.action_mean)
            a_s = f64(self.action_std).expand(nin) if self.action_std.numel() == 1 else f64(self.action_std)
            if am.numel() != nin or a_s.numel() != nin or sm.numel() != d or ss.numel() != d:
                raise ValueError("normalisation buffers do not broadcast against the model's input dims")
            desc.time_div = float(f64(self.dt)[0] * 8.0)
        else:
            sm, ss = torch.zeros(d, dtype=torch.float64), torch.ones(d, dtype=torch.float64)
            am, a_s = torch.zeros(nin, dtype=torch.float64), torch.full((nin,), 3.0, dtype=torch.float64)
            desc.time_div = 1.0
        for i in range(d):
            desc.state_mean[i], desc.state_std[i] = float(sm[i]), float(ss[i])
        for i in range(nin):
            desc.action_mean[i], desc.action_std[i] = float(am[i]), float(a_s[i])
        return desc

    def upload(self, ctx):
        """Pack the current weights into ``ctx`` (``nlc_set_rnn_model``); returns the key they were taken at."""
        if any(p.dtype != torch.float64 for p in self.parameters()):
            raise NotImplementedError(
                "the HIP path computes in float64 only: call model.double() first (reference: mppi_with_model.py:101)"
            )
        key = self._weights_key()
        sd = self.state_dict()
        blob = torch.cat([sd[k].detach().to("cpu", torch.float64).reshape(-1) for k in _BLOB_KEYS]).contiguous()
        desc = self.model_desc()
        n = ctx.lib.nlc_rnn_blob_size(C.byref(desc))
        if n != blob.numel():
            raise ValueError(f"weight blob has {blob.numel()} doubles, library expects {n}")
        ctx.check(ctx.lib.nlc_set_rnn_model(ctx.h, C.byref(desc), _lib.ptr(blob), blob.numel()))
        return key

    def hip_ctx(self, device=None):
        dev = compute_device(next(self.parameters())) if device is None else torch.device(device)
        if self._ctx is None or self._ctx.device_index != dev.index:
            self._ctx = _lib.Ctx(dev.index)
            self._uploaded_key = None
        if self._weights_key() != self._uploaded_key:
            self._uploaded_key = self.upload(self._ctx)
        return self._ctx

    def _forward_train(self, in_batch_obs, in_batch_action, ts_pred):
        """Grad-mode forward for training (``train_utils.py:388-407``): the reference's op sequence (``:618-631``) on
        PyTorch-ROCm modules; the HIP kernels serve inference / planning."""
        dev = next(self.parameters()).device
        if dev.type != "cuda":  # no CPU path in this package, training included
            raise RuntimeError("training forward: move the model to the GPU first (model.to('cuda'))")
        obs, act = in_batch_obs.to(dev), in_batch_action.to(dev)
        desc = self.model_desc()  # resolves (and rejects) the reference's normalisation branches
        d, nin = self.state_dim, desc.nin
        sm = torch.tensor(list(desc.state_mean)[:d], dtype=obs.dtype, device=dev)
        ss = torch.tensor(list(desc.state_std)[:d], dtype=obs.dtype, device=dev)
        am = torch.tensor(list(desc.action_mean)[:nin], dtype=obs.dtype, device=dev)
        a_s = torch.tensor(list(desc.action_std)[:nin], dtype=obs.dtype, device=dev)
        out, _ = self.gru((act - am) / a_s)
        feats = [out[:, -1, :], (obs - sm) / ss]
        if desc.time_input:
            feats.append(torch.as_tensor(ts_pred).to(dev, obs.dtype).reshape(obs.shape[0], 1) / desc.time_div)
        return self.linear_out(torch.cat(feats, dim=1)).to(in_batch_obs.device)

    def forward(self, in_batch_obs, in_batch_action, ts_pred):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._forward_train(in_batch_obs, in_batch_action, ts_pred)
        if torch.is_grad_enabled():
            raise NotImplementedError(
                "neurallaplacecontrol_amd.DeltaTRNN is inference-only on the HIP path: wrap the call in "
                "torch.no_grad() (as the reference harness does, mppi_with_model.py:319)"
            )
        out_device = in_batch_obs.device
        dev = compute_device(in_batch_obs, in_batch_action, next(self.parameters()))
        ctx = self.hip_ctx(dev)
        obs = in_batch_obs.detach().to(dev, torch.float64).contiguous()
        win = in_batch_action.detach().to(dev, torch.float64).contiguous()
        N, d = obs.shape
        ts = torch.as_tensor(ts_pred).detach().to(dev, torch.float64).reshape(-1).contiguous()
        if ts.numel() != N:
            raise ValueError("ts_pred must hold one prediction time per row (the reference concatenates it per row)")
        out = torch.empty((N, d), dtype=torch.float64, device=dev)
        ws = torch.empty((N, d), dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            ctx.use_torch_stream()
            ctx.check(
                ctx.lib.nlc_rnn_forward(
                    ctx.h, _lib.ptr(obs), _lib.ptr(win), _lib.ptr(ts), N, win.shape[1], _lib.ptr(out), _lib.ptr(ws)
                )
            )
        return out.to(out_device)


class RNN(DeltaTRNN):
    """Twin of ``train_utils.RNN`` (``:550-586``): ``linear_out(cat(gru(actions)[:, -1], obs))``; ``ts_pred`` is
    ignored; ``normalize=False`` selects raw observations and actions / 3."""

    def __init__(self, state_dim, action_dim, hidden_units=64, encode_obs_time=False, state_mean=None, state_std=None,
                 action_mean=None, action_std=None, normalize=False):
        nn.Module.__init__(self)
        self.encode_obs_time = encode_obs_time
        self.state_dim, self.action_dim, self.hidden_units = state_dim, action_dim, hidden_units
        self.gru = nn.GRU(action_dim, hidden_units, batch_first=True)  # the reference ignores encode_obs_time here
        self.linear_out = nn.Linear(hidden_units + state_dim, state_dim)
        self.normalize = normalize
        self.normalize_time = False
        self.register_buffer("state_mean", torch.tensor(state_mean))
        self.register_buffer("state_std", torch.tensor(state_std))
        self.register_buffer("action_mean", torch.tensor(action_mean))
        self.register_buffer("action_std", torch.tensor(action_std))
        self._ctx = None
        self._uploaded_key = None

    @classmethod
    def from_reference(cls, ref):
        first = next(ref.parameters())
        d = ref.linear_out.out_features
        m = cls(d, ref.gru.input_size, hidden_units=ref.gru.hidden_size, encode_obs_time=bool(ref.encode_obs_time),
                state_mean=[0.0] * d, state_std=[1.0] * d, action_mean=[0], action_std=[1.0],
                normalize=ref.normalize).to(device=first.device, dtype=first.dtype)
        for name in ("state_mean", "state_std", "action_mean", "action_std"):
            m.register_buffer(name, getattr(ref, name).detach().clone())
        m.load_state_dict(ref.state_dict())
        m.train(ref.training)
        return m

    def model_desc(self):
        d, nin = self.state_dim, self.action_dim
        desc = _lib.RnnDesc()
        desc.d, desc.nin, desc.hidden, desc.time_input = d, nin, self.hidden_units, 0
        desc.time_div = 1.0
        f64 = lambda t: t.detach().to("cpu", torch.float64).reshape(-1)  # noqa: E731
        if self.normalize:
            sm, ss = f64(self.state_mean), f64(self.state_std)
            am = f64(self.action_mean).expand(nin) if self.action_mean.numel() == 1 else f64(self.action_mean)
            a_s = f64(self.action_std).expand(nin) if self.action_std.numel() == 1 else f64(self.action_std)
            if am.numel() != nin or a_s.numel() != nin or sm.numel() != d or ss.numel() != d:
                raise ValueError("normalisation buffers do not broadcast against the model's input dims")
        else:
            sm, ss = torch.zeros(d, dtype=torch.float64), torch.ones(d, dtype=torch.float64)
            am, a_s = torch.zeros(nin, dtype=torch.float64), torch.full((nin,), 3.0, dtype=torch.float64)
        for i in range(d):
            desc.state_mean[i], desc.state_std[i] = float(sm[i]), float(ss[i])
        for i in range(nin):
            desc.action_mean[i], desc.action_std[i] = float(am[i]), float(a_s[i])
        return desc

    def forward(self, in_batch_obs, in_batch_action, _):
        ts = torch.zeros(in_batch_obs.shape[0], dtype=torch.float64)  # unused by the model (time_input = 0)
        return super().forward(in_batch_obs, in_batch_action, ts)
