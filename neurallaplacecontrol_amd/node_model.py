"""MI355X twin of the reference's NODE baseline, ``train_utils.NODE`` / ``xOdeFuncInXAndU``
(``train_utils.py:637-738``; factory ``get_node_model`` ``:101-125``; ``node_hidden_units=270``,
``node_augment_dim=1``, ``node_method="euler"``: ``config.py:40-42``): same constructor arguments, sub-module names and
``state_dict`` keys (``x_ode_func_in_x_and_u.linear_tanh_stack.{0,2,4}.*``, buffers ``state_mean state_std
action_mean action_std dt``), so checkpoints written by the reference load unchanged.

``forward`` is one HIP launch (``nlc_node_forward``): the ODE function's three layers on FP64 matrix cores inside the
fixed-grid Euler loop; behind ``NLDynamics`` the planner runs it inside the horizon loop (``NLC_DYN_NODE``).
``torchdiffeq.odeint(method="euler", options={"step_size": 0.05})`` is restated (grid ``k * step_size``, last point =
the end time): **parity unpinned vs upstream torchdiffeq**, which is absent offline.  The HIP path is inference-only
and float64; in grad mode ``forward`` is the same op sequence in torch ops on PyTorch-ROCm (trainable).
"""

import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from ._weights import WeightsKeyMixin
from .laplace import compute_device

_BLOB_KEYS = [f"x_ode_func_in_x_and_u.linear_tanh_stack.{i}.{p}" for i in (0, 2, 4) for p in ("weight", "bias")]


class xOdeFuncInXAndU(nn.Module):  # noqa: N801  (reference class name, train_utils.py:637)
    def __init__(self, state_dim=4, action_dim=1, nhidden=50, augment_dim=0):
        super().__init__()
        self.linear_tanh_stack = nn.Sequential(
            nn.Linear(state_dim + action_dim + augment_dim, nhidden),
            nn.Tanh(),
            nn.Linear(nhidden, nhidden),
            nn.Tanh(),
            nn.Linear(nhidden, state_dim + augment_dim),
        )
        for m in self.linear_tanh_stack.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
        self.state_dim, self.action_dim, self.augment_dim = state_dim, action_dim, augment_dim
        self.u = None

    def update_u(self, u):
        self.u = u

    def forward(self, t, x):  # stand-alone use goes through PyTorch-ROCm; NODE.forward uses the HIP kernel
        return self.linear_tanh_stack(torch.cat((x, self.u), 1))


class NODE(WeightsKeyMixin, nn.Module):
    _dyn_id = _lib.DYN_NODE  # rollout the fused planner selects for NLDynamics(model, dt)
    step_size = 0.05  # options={"step_size": 0.05} (train_utils.py:722)

    def __init__(
        self,
        state_dim,
        action_dim,
        latent_dim,
        hidden_units=64,
        encode_obs_time=False,
        state_mean=None,
        state_std=None,
        action_mean=None,
        action_std=None,
        normalize=False,
        normalize_time=False,
        method="euler",
        augment_dim=0,
        action_high=1.0,
        dt=0.05,
    ):
        super().__init__()
        if method != "euler":
            raise NotImplementedError("NODE on the HIP path integrates with the fixed-grid Euler solver only "
                                      "(config.py:40 node_method='euler')")
        self.x_ode_func_in_x_and_u = xOdeFuncInXAndU(
            state_dim=state_dim, action_dim=action_dim, augment_dim=augment_dim, nhidden=hidden_units
        )
        self.method = method
        self.state_dim, self.hidden_units = state_dim, hidden_units
        self.augment_dim = augment_dim
        self.action_dim = action_dim
        self.action_high = action_high
        self.normalize = normalize
        self.encode_obs_time = encode_obs_time
        self.normalize_time = normalize_time
        self.register_buffer("state_mean", torch.tensor(state_mean))
        self.register_buffer("state_std", torch.tensor(state_std))
        self.register_buffer("action_mean", torch.tensor(action_mean))
        self.register_buffer("action_std", torch.tensor(action_std))
        self.register_buffer("dt", torch.tensor(dt))
        self._ctx = None
        self._uploaded_key = None

    @classmethod
    def from_reference(cls, ref):
        """Twin of a loaded reference ``NODE`` (same hyper-parameters, buffers and weights, on its device)."""
        first = next(ref.parameters())
        f = ref.x_ode_func_in_x_and_u
        m = cls(
            f.state_dim, f.action_dim, f.state_dim, hidden_units=f.linear_tanh_stack[0].out_features,
            encode_obs_time=ref.encode_obs_time, state_mean=[0.0] * f.state_dim, state_std=[1.0] * f.state_dim,
            action_mean=[0], action_std=[1.0], normalize=ref.normalize, normalize_time=ref.normalize_time,
            method=ref.method, augment_dim=ref.augment_dim, action_high=ref.action_high,
        ).to(device=first.device, dtype=first.dtype)
        for name in ("state_mean", "state_std", "action_mean", "action_std", "dt"):
            m.register_buffer(name, getattr(ref, name).detach().clone())
        m.load_state_dict(ref.state_dict())
        m.train(ref.training)
        return m

    # ------------------------------------------------------------------ HIP plumbing
    def _weights_key_extra(self):
        return (self.normalize, self.normalize_time, self.step_size)

    def model_desc(self):
        d = self.state_dim
        desc = _lib.NodeDesc()
        desc.d, desc.nu, desc.hidden, desc.augment_dim = d, self.action_dim, self.hidden_units, self.augment_dim
        f64 = lambda t: t.detach().to("cpu", torch.float64).reshape(-1)  # noqa: E731
        if self.normalize:
            sm, ss = f64(self.state_mean), f64(self.state_std)
            if sm.numel() != d or ss.numel() != d:
                raise ValueError("normalisation buffers do not match state_dim")
        else:
            sm, ss = torch.zeros(d, dtype=torch.float64), torch.ones(d, dtype=torch.float64)
        desc.time_div = float(f64(self.dt)[0] * 8.0) if self.normalize_time else 1.0
        desc.step_size = float(self.step_size)
        for i in range(d):
            desc.state_mean[i], desc.state_std[i] = float(sm[i]), float(ss[i])
        return desc

    def upload(self, ctx):
        """Pack the current weights into ``ctx`` (``nlc_set_node_model``); returns the key they were taken at."""
        if any(p.dtype != torch.float64 for p in self.parameters()):
            raise NotImplementedError(
                "the HIP path computes in float64 only: call model.double() first (reference: mppi_with_model.py:101)"
            )
        key = self._weights_key()
        sd = self.state_dict()
        blob = torch.cat([sd[k].detach().to("cpu", torch.float64).reshape(-1) for k in _BLOB_KEYS]).contiguous()
        desc = self.model_desc()
        n = ctx.lib.nlc_node_blob_size(C.byref(desc))
        if n != blob.numel():
            raise ValueError(f"weight blob has {blob.numel()} doubles, library expects {n}")
        ctx.check(ctx.lib.nlc_set_node_model(ctx.h, C.byref(desc), _lib.ptr(blob), blob.numel()))
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
        """Grad-mode forward for training: the reference's op sequence (``train_utils.py:696-724``) with the restated
        fixed-grid Euler ``odeint`` in torch ops on PyTorch-ROCm; the HIP kernels serve inference / planning."""
        dev = next(self.parameters()).device
        if dev.type != "cuda":  # no CPU path in this package, training included
            raise RuntimeError("training forward: move the model to the GPU first (model.to('cuda'))")
        obs, act = in_batch_obs.to(dev), in_batch_action.to(dev)
        desc = self.model_desc()
        d = self.state_dim
        sm = torch.tensor(list(desc.state_mean)[:d], dtype=obs.dtype, device=dev)
        ss = torch.tensor(list(desc.state_std)[:d], dtype=obs.dtype, device=dev)
        x = (obs - sm) / ss
        if self.augment_dim > 0:
            x = torch.cat([x, torch.zeros(obs.shape[0], self.augment_dim, dtype=obs.dtype, device=dev)], 1)
        if act.dim() == 2:
            act = act.unsqueeze(1)
        self.x_ode_func_in_x_and_u.update_u(act[:, -1, :])
        t_end = float(torch.as_tensor(ts_pred).reshape(-1)[0]) / desc.time_div
        import math

        niters = int(math.ceil(t_end / self.step_size + 1))
        grid = [k * self.step_size for k in range(niters)]
        grid[-1] = t_end
        for k in range(niters - 1):
            x = x + (grid[k + 1] - grid[k]) * self.x_ode_func_in_x_and_u(None, x)
        return x[:, : x.shape[-1] - self.augment_dim].to(in_batch_obs.device)

    def forward(self, in_batch_obs, in_batch_action, ts_pred):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._forward_train(in_batch_obs, in_batch_action, ts_pred)
        if torch.is_grad_enabled():
            raise NotImplementedError(
                "neurallaplacecontrol_amd.NODE is inference-only on the HIP path: wrap the call in torch.no_grad() "
                "(as the reference harness does, mppi_with_model.py:319)"
            )
        out_device = in_batch_obs.device
        dev = compute_device(in_batch_obs, in_batch_action, next(self.parameters()))
        ctx = self.hip_ctx(dev)
        obs = in_batch_obs.detach().to(dev, torch.float64).contiguous()
        act = in_batch_action.detach().to(dev, torch.float64)
        if act.dim() == 2:  # train_utils.py:712-713
            act = act.unsqueeze(1)
        act = act[:, -1, :].contiguous()  # the newest action of the window, raw (:715)
        N, d = obs.shape
        t0 = float(torch.as_tensor(ts_pred).detach().reshape(-1)[0])  # every row integrates to ts_pred[0] (:719)
        out = torch.empty((N, d), dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            ctx.use_torch_stream()
            ctx.check(ctx.lib.nlc_node_forward(ctx.h, _lib.ptr(obs), _lib.ptr(act), t0, N, _lib.ptr(out)))
        return out.to(out_device)
