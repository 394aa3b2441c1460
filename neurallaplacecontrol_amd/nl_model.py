"""MI355X implementation behind ``neurallaplacecontrol_amd.w_nl`` -- drop-in for the reference's ``w_nl.py``: ``NeuralLaplaceModel`` / ``ReverseGRUEncoder`` /
``LaplaceRepresentationFunc`` with the same constructor arguments, sub-module names and
``state_dict`` keys (``action_encoder.gru.*``, ``action_encoder.linear_out.*``,
``laplace_rep_func.linear_tanh_stack.{0,2,4}.*``, buffers ``state_mean state_std action_mean
action_std dt``: reference ``w_nl.py:14-115``), so checkpoints written by the reference's
``train_utils.py:442,490`` load unchanged.

``NeuralLaplaceModel.forward`` (reference ``w_nl.py:117-145``) runs as two HIP launches behind
``nlc_model_forward``: the GRU encoder on FP64 matrix cores, then representation MLP + sphere map +
Fourier ILT fused (the (N, 2dS) theta/phi tensor never reaches HBM), float64 only (the reference harness calls
``model.double()`` under ``torch.no_grad()``: ``mppi_with_model.py:101,319``).  In grad mode (training,
``train_utils.py:388-407``) ``forward`` runs the reference's op sequence with the GRU / MLP on PyTorch-ROCm and the
line integral -- forward and backward -- in HIP, so the class is trainable on the GPU without torchlaplace.

The reference constructors take ANY ``hidden_units`` / ``state_dim`` (``w_nl.py:67-83``); the MFMA kernels are instantiated
for hidden_units 64 / 128 / 256, state_dim <= 6 and GRU input dim <= 3.  Any other shape is not an error (round 5, SURVEY
8b "falls back to the torch path on UNSUPPORTED"): when ``nlc_set_model`` answers ``NLC_ERR_UNSUPPORTED`` the no-grad forward
runs the same op sequence as the grad-mode one -- GRU and MLP as PyTorch-ROCm ops on the GPU, contour / sphere map / line
integral in the HIP ILT kernels -- with a one-time warning, and a planner over such a model takes its callables path.
"""

import copy
import ctypes as C
import warnings

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._weights import WeightsKeyMixin
from .laplace import compute_device, laplace_reconstruct


class ReverseGRUEncoder(nn.Module):
    """Encodes an observed (action) trajectory, newest element first, into a latent vector (w_nl.py:14-29)."""

    def __init__(self, dimension_in, latent_dim, hidden_units, encode_obs_time=True):
        super().__init__()
        self.encode_obs_time = encode_obs_time
        if self.encode_obs_time:
            dimension_in += 1
        self.gru = nn.GRU(dimension_in, hidden_units, 2, batch_first=True)
        self.linear_out = nn.Linear(hidden_units, latent_dim)
        nn.init.xavier_uniform_(self.linear_out.weight)

    def forward(self, observed_data):
        # stand-alone use (outside NeuralLaplaceModel) goes through PyTorch-ROCm; the fused model path
        # uses the HIP GRU kernel (NeuralLaplaceModel.encode_actions / forward)
        out, _ = self.gru(torch.flip(observed_data, (1,)))
        return self.linear_out(out[:, -1, :])


class LaplaceRepresentationFunc(nn.Module):
    """Sphere-surface model C^{b+k} -> C^{b x d} in Riemann-sphere coordinates (w_nl.py:32-63)."""

    def __init__(self, s_dim, output_dim, latent_dim, hidden_units=64):
        super().__init__()
        self.s_dim = s_dim
        self.output_dim = output_dim
        self.latent_dim = latent_dim
        self.linear_tanh_stack = nn.Sequential(
            nn.Linear(s_dim * 2 + latent_dim, hidden_units),
            nn.Tanh(),
            nn.Linear(hidden_units, hidden_units),
            nn.Tanh(),
            nn.Linear(hidden_units, s_dim * 2 * output_dim),
        )
        for m in self.linear_tanh_stack.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
        self.phi_scale = torch.pi / 2.0 - -torch.pi / 2.0

    def forward(self, i):
        out = self.linear_tanh_stack(i.view(-1, self.s_dim * 2 + self.latent_dim)).view(
            -1, 2 * self.output_dim, self.s_dim
        )
        theta = torch.tanh(out[:, : self.output_dim, :]) * torch.pi
        phi = torch.tanh(out[:, self.output_dim :, :]) * self.phi_scale / 2.0 - torch.pi / 2.0 + self.phi_scale / 2.0
        return theta, phi


# state_dict order of the weight blob nlc_set_model expects (include/nlc.h)
_BLOB_KEYS = [
    "action_encoder.gru.weight_ih_l0",
    "action_encoder.gru.weight_hh_l0",
    "action_encoder.gru.bias_ih_l0",
    "action_encoder.gru.bias_hh_l0",
    "action_encoder.gru.weight_ih_l1",
    "action_encoder.gru.weight_hh_l1",
    "action_encoder.gru.bias_ih_l1",
    "action_encoder.gru.bias_hh_l1",
    "action_encoder.linear_out.weight",
    "action_encoder.linear_out.bias",
    "laplace_rep_func.linear_tanh_stack.0.weight",
    "laplace_rep_func.linear_tanh_stack.0.bias",
    "laplace_rep_func.linear_tanh_stack.2.weight",
    "laplace_rep_func.linear_tanh_stack.2.bias",
    "laplace_rep_func.linear_tanh_stack.4.weight",
    "laplace_rep_func.linear_tanh_stack.4.bias",
]


def cme_reconstruction_terms():
    """Orders for which the CME (concentrated matrix-exponential) inversion method has published parameter sets -- the
    values of the reference's ``config.CME_reconstruction_terms()`` (config.py:278-418), generated from their run
    structure: 3..75, 101, 111..211 step 10, 216, 221, 231..391 step 10, 396, 401, 421..1001 step 20 (136 orders)."""
    t, out = 3, [3]
    for step, count in ((1, 72), (26, 1), (10, 11), (5, 2), (10, 17), (5, 2), (20, 30)):
        for _ in range(count):
            t += step
            out.append(t)
    return np.array(out)


def _cme_terms(s_recon_terms):
    """Term snapping of the reference constructor for ``ilt_algorithm == "cme"`` (w_nl.py:86-88), quirk included: the
    entry two places BEFORE the first order >= s_recon_terms (17 -> 15, 33 -> 31)."""
    terms = cme_reconstruction_terms()
    return int(terms[np.argmin(terms < s_recon_terms) - 2])


class _DeviceCopy(nn.Module):
    """Sub-modules and buffers of a host-resident model on the GPU (the generic forward's operands; re-made when the
    source's weights key moves)."""

    def __init__(self, src, dev):
        super().__init__()
        self.action_encoder = copy.deepcopy(src.action_encoder).to(dev)
        self.laplace_rep_func = copy.deepcopy(src.laplace_rep_func).to(dev)
        for name in ("state_mean", "state_std", "action_mean", "action_std", "dt"):
            self.register_buffer(name, getattr(src, name).detach().to(dev))


class NeuralLaplaceModel(WeightsKeyMixin, nn.Module):
    _dyn_id = _lib.DYN_NL  # rollout the fused planner selects for NLDynamics(model, dt)

    def __init__(
        self,
        state_dim,
        action_dim,
        latent_dim,
        hidden_units=64,
        s_recon_terms=33,
        ilt_algorithm="fourier",
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
        self.ilt_algorithm = ilt_algorithm
        if ilt_algorithm == "cme":
            s_recon_terms = _cme_terms(s_recon_terms)
        action_encoder_latent_dim = 2
        laplace_latent_dim = state_dim + action_encoder_latent_dim
        self.latent_dim = latent_dim
        self.action_dim = action_dim
        self.hidden_units = hidden_units
        self.action_encoder = ReverseGRUEncoder(
            action_dim, action_encoder_latent_dim, hidden_units // 2, encode_obs_time=encode_obs_time
        )
        self.laplace_rep_func = LaplaceRepresentationFunc(
            s_recon_terms, state_dim, laplace_latent_dim, hidden_units=hidden_units
        )
        self.encode_obs_time = encode_obs_time
        self.output_dim = state_dim
        self.normalize = normalize
        self.normalize_time = normalize_time
        self.s_recon_terms = s_recon_terms
        self.ilt_options = None  # optional dict(alpha=, tol=, scale=) override of the torchlaplace defaults
        # same dtypes as the reference (w_nl.py:111-115): torch.tensor(dt) is FLOAT32, so after .double()
        # the time normaliser is float32(0.05) widened; action_mean built from np.array([0]*nu) is int64
        self.register_buffer("state_mean", torch.tensor(state_mean))
        self.register_buffer("state_std", torch.tensor(state_std))
        self.register_buffer("action_mean", torch.tensor(action_mean))
        self.register_buffer("action_std", torch.tensor(action_std))
        self.register_buffer("dt", torch.tensor(dt))
        self._ctx = None
        self._uploaded_key = None
        self._rep_dev = None  # (key, device copy of laplace_rep_func) for the staged path

    @classmethod
    def from_reference(cls, ref):
        """MI355X twin of an instance of the reference's ``w_nl.NeuralLaplaceModel`` (any module with the same
        sub-module names and attributes, e.g. one just loaded by ``train_utils.train_model(retrain=False)``):
        same hyper-parameters, buffers (dtypes included) and weights, on ``ref``'s device.  The harness keeps training
        with its own class and plans with this one::

            model, _ = train_model("nl", env_name, config=config, retrain=False, ...)   # mppi_with_model.py:81-93
            model = neurallaplacecontrol_amd.NeuralLaplaceModel.from_reference(model.double())
        """
        gru = ref.action_encoder.gru
        enc = bool(ref.encode_obs_time)
        first = next(ref.parameters())
        m = cls(
            ref.output_dim, gru.input_size - int(enc), ref.latent_dim, hidden_units=2 * gru.hidden_size,
            s_recon_terms=ref.s_recon_terms, ilt_algorithm=ref.ilt_algorithm, encode_obs_time=enc,
            state_mean=[0.0] * ref.output_dim, state_std=[1.0] * ref.output_dim, action_mean=[0], action_std=[1.0],
            normalize=ref.normalize, normalize_time=ref.normalize_time,
        ).to(device=first.device, dtype=first.dtype)
        for name in ("state_mean", "state_std", "action_mean", "action_std", "dt"):
            m.register_buffer(name, getattr(ref, name).detach().clone())
        m.load_state_dict(ref.state_dict())
        m.train(ref.training)
        return m

    # ------------------------------------------------------------------ HIP plumbing
    def _weights_key_extra(self):
        return (self.normalize, self.normalize_time, self.ilt_algorithm, repr(self.ilt_options) if self.ilt_options is not None else None)

    def model_desc(self):
        d, nin = self.output_dim, self.action_dim + (1 if self.encode_obs_time else 0)
        if d > _lib.NLC_MAX_D or nin > _lib.NLC_MAX_NIN:  # (the descriptor's arrays end there: same answer as the library's)
            raise _lib.NlcError(_lib.NLC_ERR_UNSUPPORTED, f"state_dim {d} / GRU input dim {nin} exceed the descriptor's "
                                                          f"{_lib.NLC_MAX_D} / {_lib.NLC_MAX_NIN}")
        desc = _lib.ModelDesc()
        desc.d, desc.nin, desc.h = d, nin, self.hidden_units
        # fourier: fused forward and planner kernels; dehoog / fixed_tablot / stehfest: staged all-HIP forward (representation
        # kernel -> ILT kernel) and staged planner path
        desc.ilt = _lib.ilt_desc(self.ilt_algorithm, self.s_recon_terms, self.ilt_options)
        f64 = lambda t: t.detach().to("cpu", torch.float64).reshape(-1)  # noqa: E731
        if self.normalize:
            sm, ss = f64(self.state_mean), f64(self.state_std)
            am = f64(self.action_mean).expand(nin) if self.action_mean.numel() == 1 else f64(self.action_mean)
            a_s = f64(self.action_std).expand(nin) if self.action_std.numel() == 1 else f64(self.action_std)
            if am.numel() != nin or a_s.numel() != nin or sm.numel() != d or ss.numel() != d:
                raise ValueError("normalisation buffers do not broadcast against the model's input dims")
            desc.time_div = float(f64(self.dt)[0] * 8.0) if self.normalize_time else 1.0
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
        """Pack the current weights into ``ctx`` (``nlc_set_model``); returns the key they were taken at.  Planners
        keep their own ctx (planner state lives there) and call this when the key changes."""
        if any(p.dtype != torch.float64 for p in self.parameters()):
            raise NotImplementedError(
                "the HIP path computes in float64 only: call model.double() first (reference: mppi_with_model.py:101)"
            )
        key = self._weights_key()
        sd = self.state_dict()
        blob = torch.cat([sd[k].detach().to("cpu", torch.float64).reshape(-1) for k in _BLOB_KEYS]).contiguous()
        desc = self.model_desc()
        n = ctx.lib.nlc_model_blob_size(C.byref(desc))
        if n != blob.numel():
            raise ValueError(f"weight blob has {blob.numel()} doubles, library expects {n}")
        ctx.check(ctx.lib.nlc_set_model(ctx.h, C.byref(desc), _lib.ptr(blob), blob.numel()))
        return key

    def hip_ctx(self, device=None):
        """The model's own ``nlc_ctx`` (forward / encode_actions) with its current weights uploaded."""
        dev = compute_device(next(self.parameters())) if device is None else torch.device(device)
        if self._ctx is None or self._ctx.device_index != dev.index:
            self._ctx = _lib.Ctx(dev.index)
            self._uploaded_key = None
        if self._weights_key() != self._uploaded_key:
            self._uploaded_key = self.upload(self._ctx)
        return self._ctx

    @staticmethod
    def _no_grad_only():
        if torch.is_grad_enabled():
            raise NotImplementedError(
                "neurallaplacecontrol_amd.NeuralLaplaceModel is inference-only on the HIP path: "
                "wrap the call in torch.no_grad() (as the reference harness does, mppi_with_model.py:319)"
            )

    def encode_actions(self, in_batch_action):
        """HIP GRU encoder on raw (un-normalised) action windows (N, B, nin) -> (N, 2)  [stage a7]."""
        self._no_grad_only()
        dev = compute_device(in_batch_action, next(self.parameters()))
        ctx = self.hip_ctx(dev)
        win = in_batch_action.detach().to(dev, torch.float64).contiguous()
        if win.dim() == 2:
            win = win.unsqueeze(1)
        N, B, nin = win.shape
        out = torch.empty((N, 2), dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            ctx.use_torch_stream()
            ctx.check(ctx.lib.nlc_gru_encode(ctx.h, _lib.ptr(win), N, B, _lib.ptr(out)))
        return out

    def rep_func_hip(self, i):
        """``self.laplace_rep_func(i)`` on the MFMA kernel (``nlc_rep_func``): rows ``[theta_s | phi_s | p]`` of length
        2S + d + 2 -> ``(theta, phi)`` of shape (N, d, S) each, as ``LaplaceRepresentationFunc.forward`` (w_nl.py:55-63)."""
        self._no_grad_only()
        dev = compute_device(i, next(self.parameters()))
        ctx = self.hip_ctx(dev)
        S, d = self.s_recon_terms, self.output_dim
        rows = i.detach().to(dev, torch.float64).reshape(-1, 2 * S + d + 2).contiguous()
        N = rows.shape[0]
        theta = torch.empty((N, d, S), dtype=torch.float64, device=dev)
        phi = torch.empty((N, d, S), dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            ctx.use_torch_stream()
            ctx.check(ctx.lib.nlc_rep_func(ctx.h, _lib.ptr(rows), N, _lib.ptr(theta), _lib.ptr(phi)))
        return theta, phi

    def _forward_train(self, in_batch_obs, in_batch_action, ts_pred, mod=None):
        """Grad-mode forward (the reference trains through ``model(...)``, ``train_utils.py:388-407``): the op sequence of
        ``w_nl.py:117-145`` with the GRU encoder and the representation MLP on PyTorch-ROCm (their backward is
        autograd's) and the line integral in HIP, forward AND backward (``nlc_ilt_reconstruct_backward``).
        ``mod``: the module whose sub-modules / buffers to use (``_forward_generic``'s device copy); default ``self``."""
        mod = self if mod is None else mod
        dev = compute_device(in_batch_obs, in_batch_action, next(mod.parameters()))
        if next(mod.parameters()).device != dev:
            raise RuntimeError("training forward: move the model to the GPU first (model.to('cuda'))")
        obs = in_batch_obs.to(dev, torch.float64)
        act = in_batch_action.to(dev, torch.float64)
        ts = torch.as_tensor(ts_pred).to(dev, torch.float64)
        if self.normalize:
            batch_obs = (obs - mod.state_mean) / mod.state_std
            batch_action = (act - mod.action_mean) / mod.action_std
            if self.normalize_time:
                ts = ts / (mod.dt * 8.0)
        else:
            batch_obs = obs
            batch_action = act / 3.0
        if batch_action.dim() == 2:
            batch_action = batch_action.unsqueeze(1)
        p = torch.cat((batch_obs, mod.action_encoder(batch_action)), dim=1)
        return torch.squeeze(
            laplace_reconstruct(
                mod.laplace_rep_func, p, ts, recon_dim=self.output_dim, ilt_algorithm=self.ilt_algorithm,
                ilt_reconstruction_terms=self.s_recon_terms, options=self.ilt_options,
            )
        ).to(in_batch_obs.device)

    # ------------------------------------------------------------------ shapes the MFMA kernels are not instantiated for
    def _shape_key(self):
        return (self.hidden_units, self.output_dim, self.action_dim, self.encode_obs_time, self.ilt_algorithm,
                self.s_recon_terms, repr(self.ilt_options))

    def hip_unsupported(self):
        """The library's reason if ``nlc_set_model`` rejects this model's SHAPE (``NLC_ERR_UNSUPPORTED``), else None.  Asked
        once per shape; other failures (a float32 model, a wrong blob) raise as before."""
        hit = getattr(self, "_unsupported", None)
        if hit is not None and hit[0] == self._shape_key():
            return hit[1]
        why = None
        try:
            self.hip_ctx()
        except _lib.NlcError as err:
            if err.code != _lib.NLC_ERR_UNSUPPORTED:
                raise
            why = str(err)
            warnings.warn(f"neurallaplacecontrol_amd.NeuralLaplaceModel: {why} -- this shape runs the reference's op sequence on "
                          "PyTorch-ROCm (GRU / MLP) + the HIP ILT kernels instead of the fused MFMA kernels", stacklevel=3)
        self._unsupported = (self._shape_key(), why)
        return why

    def _forward_generic(self, in_batch_obs, in_batch_action, ts_pred):
        """No-grad forward of a shape the fused kernels do not take: ``_forward_train``'s op sequence (w_nl.py:117-145) under
        ``no_grad`` on the GPU -- through a device copy of the weights if the model itself lives on the host."""
        dev = compute_device(in_batch_obs, in_batch_action, next(self.parameters()))
        mod = self
        if next(self.parameters()).device != dev:
            key = (self._weights_key(), str(dev))
            if getattr(self, "_generic_dev", None) is None or self._generic_dev[0] != key:
                self._generic_dev = (key, _DeviceCopy(self, dev))
            mod = self._generic_dev[1]
        with torch.no_grad():
            return self._forward_train(in_batch_obs, in_batch_action, ts_pred, mod=mod)

    def _constant_ts(self, ts_pred, N):
        """The value of ts_pred if it is ONE query time per row and the same for every row (a Python number, or an
        (N,) / (N, 1) tensor of equal entries -- what the harness closure passes), else None.  Looking into a device
        tensor costs a synchronisation, so the answer is remembered per tensor (storage, version, shape): the harness
        builds ts_pred once and passes the same tensor every horizon step."""
        if not torch.is_tensor(ts_pred):
            try:
                v = float(ts_pred)
            except (TypeError, ValueError):
                return None
            return v if v > 0 else None
        if ts_pred.numel() != N or ts_pred.numel() == 0:
            return None
        key = (ts_pred.data_ptr(), ts_pred._version, tuple(ts_pred.shape), str(ts_pred.device), ts_pred.dtype)
        hit = getattr(self, "_const_ts_cache", None)
        if hit is not None and hit[0] == key:
            return hit[1]
        lo, hi = torch.aminmax(ts_pred.detach())
        lo, hi = float(lo), float(hi)
        val = lo if (lo == hi and lo > 0) else None
        self._const_ts_cache = (key, val, ts_pred)  # (holding the tensor keeps its storage address from being reused)
        return val

    def forward(self, in_batch_obs, in_batch_action, ts_pred):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._forward_train(in_batch_obs, in_batch_action, ts_pred)
        self._no_grad_only()
        if self.hip_unsupported() is not None:
            return self._forward_generic(in_batch_obs, in_batch_action, ts_pred)
        out_device = in_batch_obs.device
        dev = compute_device(in_batch_obs, in_batch_action, next(self.parameters()))
        ctx = self.hip_ctx(dev)
        obs = in_batch_obs.detach().to(dev, torch.float64).contiguous()
        win = in_batch_action.detach().to(dev, torch.float64).contiguous()
        if win.dim() == 2:
            win = win.unsqueeze(1)
        N, d = obs.shape
        t_const = self._constant_ts(ts_pred, N) if self.ilt_algorithm == "fourier" else None
        if t_const is not None:
            # one query time for every row (the harness closure's ts_pred = dt): constant sphere inputs, folded bias
            out = torch.empty((N, d), dtype=torch.float64, device=dev)
            ws = torch.empty(ctx.lib.nlc_model_workspace_bytes(ctx.h, N) // 8, dtype=torch.float64, device=dev)
            with torch.cuda.device(dev):
                ctx.use_torch_stream()
                ctx.check(ctx.lib.nlc_model_forward_const_t(ctx.h, _lib.ptr(obs), _lib.ptr(win), t_const, N, win.shape[1],
                                                            _lib.ptr(out), _lib.ptr(ws)))
            return torch.squeeze(out.view(N, 1, d)).to(out_device)
        ts = torch.as_tensor(ts_pred).detach().to(dev, torch.float64)
        fused = ts.numel() == N  # one query time per row: all-HIP forward for every implemented algorithm
        if fused:
            out = torch.empty((N, d), dtype=torch.float64, device=dev)
            ws = torch.empty(ctx.lib.nlc_model_workspace_bytes(ctx.h, N) // 8, dtype=torch.float64, device=dev)
            with torch.cuda.device(dev):
                ctx.use_torch_stream()
                ctx.check(
                    ctx.lib.nlc_model_forward(
                        ctx.h,
                        _lib.ptr(obs),
                        _lib.ptr(win),
                        _lib.ptr(ts.reshape(-1).contiguous()),
                        N,
                        win.shape[1],
                        _lib.ptr(out),
                        _lib.ptr(ws),
                    )
                )
            return torch.squeeze(out.view(N, 1, d)).to(out_device)
        # several time points per row: HIP GRU -> PyTorch-ROCm MLP (laplace_rep_func) -> HIP ILT
        desc = self.model_desc()
        sm = torch.tensor(list(desc.state_mean)[:d], dtype=torch.float64, device=dev)
        ss = torch.tensor(list(desc.state_std)[:d], dtype=torch.float64, device=dev)
        p = torch.cat(((obs - sm) / ss, self.encode_actions(win)), dim=1)
        rep = self.laplace_rep_func
        if next(rep.parameters()).device != dev:
            key = (self._weights_key(), str(dev))
            if self._rep_dev is None or self._rep_dev[0] != key:
                self._rep_dev = (key, copy.deepcopy(rep).to(dev))
            rep = self._rep_dev[1]
        return torch.squeeze(
            laplace_reconstruct(
                rep,
                p,
                ts / desc.time_div,
                recon_dim=d,
                ilt_algorithm=self.ilt_algorithm,
                ilt_reconstruction_terms=self.s_recon_terms,
                options=self.ilt_options,
            )
        ).to(out_device)


def load_replay_buffer(fn):
    """Same helper as the reference (w_nl.py:148-150)."""
    return np.load(fn, allow_pickle=True).item()
