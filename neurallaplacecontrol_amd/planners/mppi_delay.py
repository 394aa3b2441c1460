"""MI355X drop-in for the reference planner ``planners/mppi_delay.py`` (class ``MPPIDelay`` :54-381).

Same constructor signature, ``command(state, action_buffer)``, ``reset()``, ``get_rollouts()`` and public
attributes (``U noise perturbed_action cost_total cost_total_non_zero omega states actions K T nx nu
lambda_ u_scale``).  Everything numeric runs in HIP kernels behind ``libnlc_hip.so``:

* fused path -- ``dynamics`` is an :class:`~neurallaplacecontrol_amd.envs.NLDynamics` or
  :class:`~neurallaplacecontrol_amd.envs.OracleDynamics` and ``running_cost`` an
  :class:`~neurallaplacecontrol_amd.envs.EnvCost`: one ``command()`` = shift/perturb kernel, hoisted GRU
  encode over all K*T windows, one persistent T-step rollout kernel, softmax-weight reduction, U update.
* fused dynamics + cost callables -- same rollout kernel, but ``running_cost`` (and ``terminal_state_cost``) are
  arbitrary callables, e.g. the harness's ``state_constraint`` / ``change_goal`` closures
  (``mppi_with_model.py:145-171``): the states do not depend on the cost, so the callables run once per horizon step
  on the stored (K, T, nx) device states after the kernel (``nlc_mppi_desc.cost_external``).
* generic path -- arbitrary dynamics callables (the reference's contract): sampling, bounding, weighting and the
  U update are the same HIP kernels; the T-step loop calls the user's callables on device tensors.

With a ``process_group`` every rank must make the SAME sequence of calls: the constructor, ``reset()`` and every assignment
to ``.U`` are collective (U is replicated from rank 0 by a broadcast, so that the ranks' copies cannot drift), ``command()``
contains the one all-gather of the shard partials.  Calling any of them on a subset of ranks deadlocks.

Extra keyword-only arguments (not in the reference): ``noise_rng`` ("torch": draw with
``MultivariateNormal`` on ``device`` exactly like the reference, so a CPU-seeded run reproduces the
reference's noise bit for bit; "philox": draw inside the perturb kernel), ``seed``, ``process_group``
(shard the K samples over the ranks of a torch.distributed group; RCCL all-gather of 2+T*nu doubles per
command), ``compute_device``, ``store_rollouts``.
"""

import ctypes as C

import torch
from torch.distributions.multivariate_normal import MultivariateNormal

from .. import _lib, _recognise
from ..envs import EnvCost, NLDynamics, OracleDynamics
from ..sharding import (all_ranks_agree, check_same_on_all_ranks, gather_partials, merge_partials_torch, replicate_from_rank0, shard_range,
                        share_bytes_from_rank0, slice_noise)


def _backend_is_rccl(group):
    import torch.distributed as dist

    return str(dist.get_backend(group)).lower() == "nccl"


def _per_dim(v, nu, name):
    t = torch.as_tensor(v, dtype=torch.float64).detach().cpu().reshape(-1)
    if t.numel() == 1:
        t = t.expand(nu)
    if t.numel() != nu:
        raise ValueError(f"{name} must be a scalar or have nu={nu} entries")
    return [float(x) for x in t]


class MPPIDelay:
    def __init__(
        self,
        dynamics,
        running_cost,
        nx,
        noise_sigma,
        num_samples=100,
        horizon=15,
        device="cpu",
        terminal_state_cost=None,
        lambda_=1.0,
        noise_mu=None,
        u_min=None,
        u_max=None,
        u_init=None,
        U_init=None,
        u_scale=1,
        u_per_command=1,
        step_dependent_dynamics=False,
        rollout_samples=1,
        rollout_var_cost=0,
        rollout_var_discount=0.95,
        dt=0.05,
        sample_null_action=False,
        noise_abs_cost=False,
        encode_obs_time=False,
        *,
        noise_rng="torch",
        seed=0,
        process_group=None,
        compute_device=None,
        store_rollouts=True,
        planner_options=None,
    ):
        self.d = torch.device(device)
        self.dtype = noise_sigma.dtype
        # what the HIP planner kernels are not built for (they compute in float64 and their descriptor carries NLC_MAX_NU action
        # dims) is not an error of a drop-in constructor -- the reference class takes any nu and dtype (planners/mppi_delay.py:
        # 115-135): such a planner runs the reference's op sequence as PyTorch-ROCm tensor ops on the GPU (_torch_command)
        self.torch_path = None  # the reason, once one is found
        if self.dtype != torch.float64:
            self.torch_path = f"noise_sigma is {self.dtype}: the HIP planner kernels compute in float64"
        self.K = int(num_samples)
        self.T = int(horizon)
        self.E = int(getattr(self, "E", 1))  # episodes planned side by side (BatchedMPPIDelay sets it first)
        self.encode_obs_time = encode_obs_time
        self.dt = dt
        self.nx = nx
        self.nu = 1 if len(noise_sigma.shape) == 0 else noise_sigma.shape[0]
        if self.nu > _lib.NLC_MAX_NU:
            self.torch_path = f"nu={self.nu}: the HIP planner kernels carry up to {_lib.NLC_MAX_NU} action dims"
        if self.torch_path and self.E != 1:
            raise NotImplementedError(f"BatchedMPPIDelay: {self.torch_path}")
        self.lambda_ = lambda_
        if noise_mu is None:
            noise_mu = torch.zeros(self.nu, dtype=self.dtype)
        if u_init is None:
            u_init = torch.zeros_like(noise_mu)
        if self.nu == 1:
            noise_mu = noise_mu.view(-1)
            noise_sigma = noise_sigma.view(-1, 1)
        # bounds: if only one of them is given the other is its negation (reference :143-153)
        self.u_min, self.u_max = u_min, u_max
        self.u_scale = u_scale
        self.u_per_command = u_per_command
        if self.u_max is not None and self.u_min is None:
            self.u_max = torch.as_tensor(self.u_max)
            self.u_min = -self.u_max
        if self.u_min is not None and self.u_max is None:
            self.u_min = torch.as_tensor(self.u_min)
            self.u_max = -self.u_min
        if self.u_min is not None:
            self.u_min = torch.as_tensor(self.u_min).to(device=self.d)
            self.u_max = torch.as_tensor(self.u_max).to(device=self.d)
        self.noise_mu = noise_mu.to(self.d)
        self.noise_sigma = noise_sigma.to(self.d)
        self.noise_sigma_inv = torch.inverse(self.noise_sigma)
        self.noise_dist = MultivariateNormal(self.noise_mu, covariance_matrix=self.noise_sigma)
        self.u_init = u_init.to(self.d)

        if rollout_samples < 1:
            raise ValueError("rollout_samples must be >= 1")
        self.M = rollout_samples
        self.rollout_var_cost = rollout_var_cost
        self.rollout_var_discount = rollout_var_discount
        self.step_dependency = step_dependent_dynamics
        self.terminal_state_cost = terminal_state_cost
        self.sample_null_action = sample_null_action
        self.noise_abs_cost = noise_abs_cost
        self._state_in = None

        if noise_rng not in ("torch", "philox"):
            raise ValueError("noise_rng must be 'torch' or 'philox'")
        self.noise_rng = noise_rng
        self.seed = int(seed)
        self._commands = 0
        self._store_rollouts_arg = store_rollouts

        # ---- K-sharding over a process group
        self.pg = process_group
        if process_group is not None:
            import torch.distributed as dist

            self.G, self.rank = dist.get_world_size(process_group), dist.get_rank(process_group)
        else:
            self.G, self.rank = 1, 0
        self.k_offset, self.K_local = shard_range(self.K, self.G, self.rank)

        self._F_callable = dynamics  # what get_rollouts() replays through (reference :358-381)
        self._decide_mode(dynamics, running_cost)
        # The reference harness's LITERAL closures (mppi_with_model.py:103-122, 129-143, 145-171): look inside them for a
        # model + constant ts_pred / an oracle partial / an env, and -- once verified at the first command() by a short
        # probe run both ways (_recognise.probe_equivalence) -- plan on the fused path instead of the generic one.
        self._candidate = None
        opts_in = dict(planner_options or {})
        self.recognised = False  # True once literal closures have been verified and replaced by their fused twins
        self.unsupported_shape = None  # the library's reason when the model's shape sent the planner to the callables path
        if self.torch_path:
            self.fused_dynamics = self.fused = self.cost_external = False  # the callables are called as they are
            self.store_rollouts = True
        if (bool(float(opts_in.pop("recognise_closures", 1))) and not self.fused_dynamics and not step_dependent_dynamics
                and self.E == 1 and type(self) is MPPIDelay and not self.torch_path):
            cd = _recognise.candidate_dynamics(dynamics)
            if cd is not None:
                self._candidate = (cd, _recognise.candidate_cost(running_cost) or running_cost)
        planner_options = opts_in

        if compute_device is None:
            model_holder = dynamics if isinstance(dynamics, NLDynamics) else (
                self._candidate[0] if self._candidate is not None and isinstance(self._candidate[0], NLDynamics) else None)
            if self.d.type == "cuda":
                compute_device = self.d
            elif model_holder is not None and next(model_holder.model.parameters()).is_cuda:
                compute_device = next(model_holder.model.parameters()).device
        if not torch.cuda.is_available():
            raise RuntimeError("neurallaplacecontrol_amd.MPPIDelay needs an AMD MI355X; there is no CPU path")
        self.cd = torch.device(compute_device) if compute_device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.cd.index is None:
            self.cd = torch.device("cuda", torch.cuda.current_device())
        if self.torch_path:
            import warnings

            warnings.warn(f"neurallaplacecontrol_amd.MPPIDelay: {self.torch_path} -- planning with PyTorch-ROCm tensor ops on "
                          f"{self.cd} (sampling, bounding, the callables' T-step loop, weights and the U update; no HIP planner kernel)",
                          stacklevel=2)
            self.ctx, self.native_collective = None, False
            self._B = self._buf = self._pending_U = self._model_key = None
            self._noise = self._perturbed = self._states = self._actions = None
            self._cost_total = self._cost_nz = self._omega = None
            self._U_t = None
            if self.pg is not None and self.G > 1:
                if self.M > 1 and self.rollout_var_cost != 0:
                    # (the variance of the running cost is a statistic of the WHOLE population: the HIP path all-reduces it,
                    # this path would silently use each rank's own samples)
                    raise NotImplementedError("rollout_samples > 1 with rollout_var_cost on a K-sharded planner that plans with "
                                              f"tensor ops ({self.torch_path})")
                check_same_on_all_ranks((self.K, self.T, self.nu), self.pg, "num_samples / horizon / nu", self.cd)
            self.U = U_init if U_init is not None else self.noise_dist.sample((self.T,))
            return
        # every planner owns its ctx: U and the folded layer-1 bias live there, so two planners over one model
        # must not share one (the model's own ctx serves model.forward only)
        self.ctx = _lib.Ctx(self.cd.index)
        # tuning knobs of include/nlc.h (nlc_set_option) come from `planner_options` only: no environment variable is read,
        # so the ranks of a sharded planner cannot silently differ
        opts = dict(planner_options or {})
        # "native_collective": the per-command all-gather runs inside nlc_mppi_finish on the library's own RCCL
        # communicator (include/nlc.h, nlc_comm_init) instead of torch.distributed between the two phases.  Default
        # (None): on whenever the group's backend is RCCL ("nccl") -- no host hop between the rollout and the action.
        nat = opts.pop("native_collective", None)
        if nat is None:
            nat = self.pg is not None and _backend_is_rccl(self.pg)
        self.native_collective = bool(float(nat)) and self.pg is not None
        for name, value in opts.items():
            if value is not None:
                self.ctx.set_option(name, float(value))
        if self.native_collective:
            uid = share_bytes_from_rank0(self.ctx.comm_unique_id() if self.rank == 0 else None, _lib.COMM_ID_BYTES, self.pg,
                                         self.cd)
            ok = 1
            try:
                with torch.cuda.device(self.cd):
                    self.ctx.use_torch_stream()
                    self.ctx.comm_init(self.rank, self.G, uid)
                    self.ctx.comm_self_test()  # one all-gather of the rank numbers, checked on the host
            except _lib.NlcError as err:
                ok, self._native_error = 0, str(err)
            # the ranks agree FIRST (a rank that raised on its own would leave its peers waiting in this all-reduce), then an
            # explicit request that could not be met fails on every rank together
            agreed = all_ranks_agree(ok, self.pg, self.cd)
            if not agreed and planner_options and planner_options.get("native_collective"):
                with torch.cuda.device(self.cd):
                    self.ctx.comm_destroy()
                raise RuntimeError("native_collective was requested, but the library-owned RCCL communicator could not be "
                                   f"brought up on every rank ({getattr(self, '_native_error', 'failed on another rank')})")
            if not agreed:
                # the default is a preference: if any rank could not bring the communicator up, every rank uses the
                # collective of the group the caller gave us
                import warnings

                warnings.warn("neurallaplacecontrol_amd: library-owned RCCL communicator unavailable "
                              f"({getattr(self, '_native_error', 'failed on another rank')}); using torch.distributed's all-gather")
                self.native_collective = False
                with torch.cuda.device(self.cd):
                    self.ctx.comm_destroy()
        self._model_key = None
        self._B = None
        self._buf = None
        self._pending_U = None

        # sampled results from the last command (device tensors; exposed through the properties below)
        self._noise = self._perturbed = self._states = self._actions = None
        self._cost_total = self._cost_nz = self._omega = None

        if self.pg is not None and self.G > 1:
            check_same_on_all_ranks((self.K, self.T, self.nu, self.seed, int(self.noise_rng == "philox")), self.pg,
                                    "num_samples / horizon / nu / seed / noise_rng", self.cd)
        # T x nu control sequence; defaults to a noise draw (consumes the RNG like the reference :161-164)
        self.U = U_init if U_init is not None else self.noise_dist.sample(self._lead(self.T))

    def _decide_mode(self, dynamics, running_cost):
        """Which planner path the (dynamics, running_cost) pair runs on: fused / fused dynamics + cost callables / generic."""
        self.F = dynamics
        self.running_cost = running_cost
        self.store_rollouts = self._store_rollouts_arg
        # fused dynamics: the whole T-step rollout is one HIP kernel (Fourier models), or the staged all-HIP path
        # (rep-func kernel -> ILT kernel -> state kernel per horizon step) for de Hoog, fixed_tablot and stehfest models
        self.fused_dynamics = (
            isinstance(dynamics, (NLDynamics, OracleDynamics))
            and not self.step_dependency
            # the collector's encode_obs_time variant only appends a time-stamp channel to the window; oracle
            # dynamics ignore it (oracle.py:23 takes [:, -(delay+1), :nu]), an NL model consumes it -> generic path
            and not (self.encode_obs_time and not isinstance(dynamics, OracleDynamics))
        )
        # fused: the running cost is evaluated inside the rollout kernel as well (EnvCost, no terminal cost)
        self.fused = self.fused_dynamics and isinstance(running_cost, EnvCost) and self.terminal_state_cost is None
        # otherwise, with fused dynamics, the cost callables (the harness's state_constraint / change_goal closures,
        # a terminal cost, ...) run on the stored device states after the rollout: the states do not depend on them
        self.cost_external = self.fused_dynamics and not self.fused
        if self.cost_external:
            self.store_rollouts = True  # the cost callables read the stored states
        if self.M > 1:
            # rollout_samples M > 1 (reference :291-292, 310).  The reference never replicates the state M times: its M
            # cost rows are copies and ``c.var(dim=0)`` is the variance of the running cost OVER THE K SAMPLES, one number
            # per horizon step -- every sample's cost gets the same rollout_var_cost * sum_t var_t * discount^t.  Softmax
            # weights, U and the action do not see a constant shift; it is added to .cost_total after the command, from
            # the stored rollout.
            # K-sharded: the variance is a statistic of the WHOLE population -- two tiny all-reduces per command (the
            # per-step sums, then the per-step squared deviations); BatchedMPPIDelay: one variance per episode.
            self.store_rollouts = True
        if isinstance(dynamics, OracleDynamics) and not self.fused_dynamics:
            raise NotImplementedError("OracleDynamics needs the default rollout options (no step-dependent dynamics)")

    def _verify_candidate(self, state, action_buffer):
        """First command(): run the literal closures and the recognised fused objects on one short probe command (this
        command's state and action buffer, a private noise draw) and switch to the fused path iff they agree."""
        cand, self._candidate = self._candidate, None
        literal = (self.F, self.running_cost)

        def make(dyn, cost, horizon):
            return MPPIDelay(
                dyn, cost, self.nx, self.noise_sigma if self.nu > 1 else self.noise_sigma.view(1, 1), self.K, horizon,
                device=self.d, terminal_state_cost=self.terminal_state_cost, lambda_=self.lambda_, noise_mu=self.noise_mu,
                u_min=self.u_min, u_max=self.u_max, u_init=self.u_init, U_init=torch.zeros(horizon, self.nu, dtype=torch.float64),
                u_scale=self.u_scale, sample_null_action=self.sample_null_action, noise_abs_cost=self.noise_abs_cost,
                encode_obs_time=self.encode_obs_time, dt=self.dt, compute_device=self.cd, store_rollouts=True,
                planner_options={"recognise_closures": 0},
            )

        ok = _recognise.probe_equivalence(make, literal, cand, state, action_buffer, self.K, self.nu)
        if self.pg is not None and self.G > 1:
            ok = all_ranks_agree(ok, self.pg, self.cd)
        self.recognised = bool(ok)
        if ok:
            self._decide_mode(*cand)

    def _lead(self, *shape):
        """Shape with the leading episode dimension of a batched planner."""
        return tuple(shape) if self.E == 1 else (self.E,) + tuple(shape)

    # ------------------------------------------------------------------ configuration
    def _configure(self, B):
        d = _lib.MppiDesc()
        d.K, d.K_global, d.k_offset = self.K_local, self.K, self.k_offset
        d.E = self.E
        d.T, d.nu, d.d, d.B = self.T, self.nu, self.nx, B
        d.lambda_, d.u_scale = float(self.lambda_), float(self.u_scale)
        d.has_bounds = int(self.u_max is not None)
        if self.u_max is not None:
            lo, hi = _per_dim(self.u_min, self.nu, "u_min"), _per_dim(self.u_max, self.nu, "u_max")
            for i in range(self.nu):
                d.u_min[i], d.u_max[i] = lo[i], hi[i]
        ui, mu = _per_dim(self.u_init, self.nu, "u_init"), _per_dim(self.noise_mu, self.nu, "noise_mu")
        sig = self.noise_sigma.detach().cpu().to(torch.float64)
        inv = self.noise_sigma_inv.detach().cpu().to(torch.float64)
        chol = torch.linalg.cholesky(sig)
        for i in range(self.nu):
            d.u_init[i], d.noise_mu[i] = ui[i], mu[i]
            for j in range(self.nu):
                d.noise_sigma[i * self.nu + j] = float(sig[i, j])
                d.noise_sigma_inv[i * self.nu + j] = float(inv[i, j])
                d.noise_chol[i * self.nu + j] = float(chol[i, j])
        d.sample_null_action = int(bool(self.sample_null_action))
        d.noise_abs_cost = int(bool(self.noise_abs_cost))
        d.u_per_command = int(self.u_per_command)
        if self.fused_dynamics:
            d.cost_external = int(self.cost_external)
            if isinstance(self.F, NLDynamics):
                d.dynamics, d.ts_pred = self.F.model._dyn_id, self.F.ts_pred  # DYN_NL or DYN_DTRNN
                d.env = _lib.ENV_IDS[self.running_cost.env_name] if self.fused else -1
            else:
                if self.fused and self.F.env_name != self.running_cost.env_name:
                    raise ValueError("OracleDynamics and EnvCost name different envs")
                d.env = _lib.ENV_IDS[self.F.env_name]
                d.dynamics, d.ts_pred = _lib.DYN_ORACLE, self.F.ts
                d.delay, d.friction = self.F.delay, int(self.F.friction)
        else:
            d.dynamics = _lib.DYN_EXTERNAL
        old_U = None
        if self._B is not None:
            old_U = self.U
        self.ctx.check(self.ctx.lib.nlc_mppi_configure(self.ctx.h, C.byref(d)))
        self._B = B
        K, T, nu, nx, dev = self.K_local, self.T, self.nu, self.nx, self.cd
        mk = lambda *s: torch.empty(self._lead(*s), dtype=torch.float64, device=dev)  # noqa: E731
        self._noise, self._perturbed = mk(K, T, nu), mk(K, T, nu)
        self._states = mk(K, T, nx) if self.store_rollouts else None
        self._actions = mk(K, T, nu) if self.store_rollouts else None
        self._cost_total, self._cost_nz, self._omega = mk(K), mk(K), mk(K)
        self._partials = mk(2 + T * nu)
        self._action = mk(self.u_per_command * nu)
        self._gathered = (
            torch.empty((self.G,) + self._lead(2 + T * nu), dtype=torch.float64, device=dev) if self.pg is not None else None
        )
        self._ws = torch.empty(self.ctx.lib.nlc_mppi_workspace_bytes(self.ctx.h) // 8, dtype=torch.float64, device=dev)
        b = _lib.MppiBuffers()
        b.noise, b.perturbed = self._noise.data_ptr(), self._perturbed.data_ptr()
        b.states = self._states.data_ptr() if self._states is not None else None
        b.actions = self._actions.data_ptr() if self._actions is not None else None
        b.cost_total, b.cost_nz, b.omega = self._cost_total.data_ptr(), self._cost_nz.data_ptr(), self._omega.data_ptr()
        b.partials, b.workspace = self._partials.data_ptr(), self._ws.data_ptr()
        b.action = self._action.data_ptr()
        self._buf = b
        if old_U is not None:
            self._pending_U = old_U
        if self._pending_U is not None:
            self._upload_U(self._pending_U)
            self._pending_U = None

    def _ensure_configured(self, B):
        """(Re)configure when the action-buffer length or the model's weights changed since the last command."""
        stale = self._buf is None or B != self._B
        if self.fused_dynamics and isinstance(self.F, NLDynamics):
            model = self.F.model
            _recognise.refresh_twin(model)  # a twin of a reference model instance follows that instance's weight updates
            if model._weights_key() != self._model_key:
                if self._buf is not None:  # nlc_set_model drops the planner configuration: carry U over
                    self._pending_U, self._buf, self._B = self.U, None, None
                try:
                    self._model_key = model.upload(self.ctx)
                except _lib.NlcError as err:
                    if err.code != _lib.NLC_ERR_UNSUPPORTED:
                        raise
                    # a model shape the rollout kernels are not instantiated for (hidden_units other than 64 / 128 / 256,
                    # state_dim > 6, ...): the reference's constructors accept it (w_nl.py:67-83), so plan on the callables
                    # path -- sampling, bounding, weighting and the U update stay HIP kernels, the T-step loop calls
                    # NLDynamics.__call__ -> model.forward (PyTorch-ROCm GRU / MLP + HIP ILT) and the cost callable
                    import warnings

                    warnings.warn(f"neurallaplacecontrol_amd.MPPIDelay: {err} -- planning on the callables path "
                                  "(NLC_DYN_EXTERNAL) with the model's PyTorch-ROCm forward", stacklevel=3)
                    self.fused_dynamics = self.fused = self.cost_external = False
                    self.unsupported_shape = str(err)
                    return self._ensure_configured(B)
                stale = True  # the constant sphere inputs are folded into the layer-1 bias at configure time
        if stale:
            self._configure(B)

    def _upload_U(self, U):
        Uh = torch.as_tensor(U).detach().to("cpu", torch.float64).reshape(self._lead(self.T, self.nu)).contiguous()
        self.ctx.check(self.ctx.lib.nlc_mppi_set_U(self.ctx.h, _lib.ptr(Uh)))

    # ------------------------------------------------------------------ public state
    @property
    def U(self):
        if self.torch_path:
            return self._U_t.to(self.d)
        if self._buf is None:
            return self._pending_U
        Uh = torch.empty(self._lead(self.T, self.nu), dtype=torch.float64)
        self.ctx.check(self.ctx.lib.nlc_mppi_get_U(self.ctx.h, _lib.ptr(Uh)))
        return Uh.to(self.d)

    @U.setter
    def U(self, value):
        if self.torch_path:
            value = torch.as_tensor(value).detach().to(dtype=self.dtype).reshape(self.T, self.nu).clone()
            if self.pg is not None and self.G > 1:
                value = replicate_from_rank0(value, self.pg, self.cd).to(self.dtype)
            self._U_t = value.to(self.cd)
            return
        value = torch.as_tensor(value).detach().to(dtype=torch.float64).reshape(self._lead(self.T, self.nu)).clone()
        if self.pg is not None and self.G > 1:
            # every rank applies the same update to its OWN copy of U: start them from rank 0's value
            value = replicate_from_rank0(value, self.pg, self.cd)
        if self._buf is None:
            self._pending_U = value
        else:
            self._upload_U(value)

    def _out(self, t):
        return None if t is None else (t if self.d == t.device else t.to(self.d))

    @property
    def state(self):
        """The state handed to the last command, on ``device`` (reference :196-198)."""
        s = self._state_in
        return None if s is None else s.to(dtype=self.dtype, device=self.d)

    @state.setter
    def state(self, value):
        self._state_in = value

    # ---- what the library did (include/nlc.h, nlc_get_stat); not in the reference
    _BODIES = {0: None, 1: "wave-per-tile", 2: "latency-split", 3: "fused", 4: "staged", 5: "dehoog-chain", 6: "oracle",
               7: "dtrnn", 8: "node", 9: "callables"}

    @property
    def rollout_body(self):
        """Which hand-written body phase 1 of the LAST command ran on ("fused": the one-launch body, which assumes the
        device to itself; "latency-split" / "wave-per-tile": GRU launch + rollout launch; ...); None before the first."""
        if self.torch_path:
            return "callables-torch" if self._cost_total is not None else None
        return self._BODIES.get(int(self.ctx.get_stat("rollout_body")))

    @property
    def fused_timeouts(self):
        """Fused launches of this planner whose bounded waits expired (each one: a command re-run on the two-launch body, or
        lost).  Non-zero means the GPU is shared with work the fused body cannot see: the planner has left that body."""
        return 0 if self.torch_path else int(self.ctx.get_stat("fused_timeouts"))

    @property
    def fused_fallbacks(self):
        """Commands re-run on the two-launch body inside nlc_mppi_finish (a give-up here or on another rank)."""
        return 0 if self.torch_path else int(self.ctx.get_stat("fused_fallbacks"))

    noise = property(lambda self: self._out(self._noise))
    perturbed_action = property(lambda self: self._out(self._perturbed))
    states = property(lambda self: self._out(self._states))
    actions = property(lambda self: self._out(self._actions))
    cost_total = property(lambda self: self._out(self._cost_total))
    cost_total_non_zero = property(lambda self: self._out(self._cost_nz))
    omega = property(lambda self: self._out(self._omega))

    # ------------------------------------------------------------------ command
    def command(self, state, action_buffer):
        """
        :param state: (nx) or (K x nx) current state, or samples of states
        :param action_buffer: (B x nu) most recent actions, oldest first (harness get_action contract)
        :returns action: (nu) best action, or (u_per_command x nu)
        """
        if not torch.is_tensor(state):
            state = torch.tensor(state)
        self._state_in = state  # .state (reference attribute) converts lazily: no device round trip per command
        if self.torch_path:
            return self._torch_command(state, action_buffer)
        st = state.detach().to("cpu", torch.float64).contiguous()
        per_sample = tuple(st.shape) == (self.K, self.nx)
        if per_sample:
            st = st[self.k_offset : self.k_offset + self.K_local].contiguous()
        elif st.numel() != self.nx:
            raise ValueError(f"state must have nx={self.nx} entries or be (K, nx)")
        ab = torch.as_tensor(action_buffer).detach().to("cpu", torch.float64).contiguous()
        if ab.dim() != 2:
            raise ValueError("action_buffer must be (B, nu)")
        lib, ctx = self.ctx.lib, self.ctx
        rng = 1 if self.noise_rng == "philox" else 0
        if self._candidate is not None:
            self._verify_candidate(state, action_buffer)
        with torch.cuda.device(self.cd):
            ctx.use_torch_stream()  # before any (re)configuration: its uploads are ordered on this stream too
            self._ensure_configured(ab.shape[0])
            if not rng:
                # K x T x nu draw on `device`, same generator consumption as the reference (:319)
                raw = self.noise_dist.sample((self.K, self.T))
                self._noise.copy_(slice_noise(raw, self.k_offset, self.K_local).reshape(self.K_local, self.T, self.nu))
            if self.fused_dynamics:
                if self.encode_obs_time and ab.shape[1] == self.nu + 1:
                    ab = ab[:, : self.nu].contiguous()  # drop the time-stamp column (mppi_delay.py:262-264)
                if ab.shape[1] != self.nu:
                    raise ValueError("action_buffer must have nu columns")
                ctx.check(
                    lib.nlc_mppi_rollout(
                        ctx.h, _lib.ptr(st), int(per_sample), _lib.ptr(ab), C.byref(self._buf), rng, self.seed, self._commands
                    )
                )
                if self.cost_external:
                    self._external_cost()
                    ctx.check(lib.nlc_mppi_weights(ctx.h, C.byref(self._buf)))
            else:
                ctx.check(lib.nlc_mppi_rollout(ctx.h, None, 0, None, C.byref(self._buf), rng, self.seed, self._commands))
                self._external_rollout(st, per_sample, ab)
                ctx.check(lib.nlc_mppi_weights(ctx.h, C.byref(self._buf)))
            self._commands += 1
            if self.native_collective:
                gathered = None  # nlc_mppi_finish gathers on the command's stream (its own communicator)
            elif self.pg is not None:  # also for a 1-rank group: the collective path is the same code
                gathered = gather_partials(self._partials, self._gathered, self.pg)
            else:
                gathered = self._partials
            if self.d.type == "cuda":
                # the action stays on the device (like the reference on a CUDA device): no copy-back, no host
                # synchronisation here -- the caller's .cpu() / .item() is the sync point
                ctx.check(lib.nlc_mppi_finish(ctx.h, _lib.ptr(gathered), self.G, self.rank, C.byref(self._buf), None))
                act = self._action.clone()
            else:
                act = torch.empty(self.u_per_command * self.nu, dtype=torch.float64)
                rc = lib.nlc_mppi_finish(ctx.h, _lib.ptr(gathered), self.G, self.rank, C.byref(self._buf), _lib.ptr(act))
                if rc == _lib.NLC_AGAIN:
                    # a rank's fused launch gave up: EVERY rank saw the marked partial row after the all-gather and has
                    # re-run the command on the two-launch body (include/nlc.h); the collective is ours, so gather again
                    gathered = gather_partials(self._partials, self._gathered, self.pg)
                    rc = lib.nlc_mppi_finish(ctx.h, _lib.ptr(gathered), self.G, self.rank, C.byref(self._buf), _lib.ptr(act))
                ctx.check(rc)
            if self.M > 1 and self.rollout_var_cost != 0 and self.fused_dynamics:
                self._add_rollout_var_cost()
        action = act.view(self.u_per_command, self.nu)
        if self.u_per_command == 1:
            action = action[0]
        return action if action.device == self.d or self.d.index is None and action.is_cuda == (self.d.type == "cuda") else action.to(self.d)

    # ------------------------------------------------------------------ planners the HIP kernels are not built for
    def _torch_command(self, state, action_buffer):
        """command() as PyTorch-ROCm tensor ops on the compute device, in the planner's own dtype and for any nu: the
        reference's op sequence (planners/mppi_delay.py:199-224, 319-344) -- shift U, draw on `device` (same generator
        consumption), bound, the callables' T-step loop (_external_rollout), softmax weights, U update.  K-sharded: the
        same (beta_r, eta_r, S_r) partials and the same one all-gather as the HIP path (sharding.merge_partials_torch)."""
        dev, dt, K, T, nu = self.cd, self.dtype, self.K_local, self.T, self.nu
        ab = torch.as_tensor(action_buffer).detach().to(dev, dt)
        if ab.dim() != 2:
            raise ValueError("action_buffer must be (B, nu)")
        st = state.detach().to(dev, dt)
        per_sample = tuple(st.shape) == (self.K, self.nx)
        if per_sample:
            st = st[self.k_offset : self.k_offset + K]
        elif st.numel() != self.nx:
            raise ValueError(f"state must have nx={self.nx} entries or be (K, nx)")
        U = torch.roll(self._U_t, -1, dims=0)
        U[-1] = self.u_init.to(dev, dt)
        self._U_t = U
        raw = self.noise_dist.sample((self.K, T))
        V = U + slice_noise(raw, self.k_offset, K).to(dev).reshape(K, T, nu)
        if self.sample_null_action and self.k_offset + K == self.K:
            V[K - 1] = 0  # global sample K - 1 lives on the last rank
        V = V * self.u_scale
        if self.u_max is not None:
            V = torch.max(torch.min(V, self.u_max.to(dev)), self.u_min.to(dev))
        V = V / self.u_scale
        self._perturbed, self._noise = V, V - U
        self._cost_total = torch.empty(K, dtype=dt, device=dev)
        self._states = torch.empty(K, T, self.nx, dtype=dt, device=dev)
        self._external_rollout(st, per_sample, ab)
        cost, lam = self._cost_total, self.lambda_
        if self.pg is None:
            beta = cost.min()
            self._cost_nz = torch.exp(-(cost - beta) / lam)
            self._omega = (1.0 / self._cost_nz.sum()) * self._cost_nz
            dU = (self._omega.view(-1, 1, 1) * self._noise).sum(dim=0)
        else:
            beta_r = cost.min()
            w = torch.exp(-(cost - beta_r) / lam)
            part = torch.cat((beta_r.view(1), w.sum().view(1), (w.view(-1, 1, 1) * self._noise).sum(dim=0).reshape(-1)))
            gathered = torch.empty(self.G, 2 + T * nu, dtype=dt, device=dev)
            beta, eta, S = merge_partials_torch(gather_partials(part, gathered, self.pg), lam)
            self._cost_nz = torch.exp(-(cost - beta) / lam)
            self._omega = self._cost_nz / eta
            dU = (S / eta).view(T, nu)
        self._U_t = U + dU
        self._commands += 1
        action = self._U_t[: self.u_per_command] * self.u_scale
        if self.u_per_command == 1:
            action = action[0]
        return action.to(self.d)

    # ------------------------------------------------------------------ generic callables (reference :232-313)
    def _dynamics(self, state, u, t):
        F = self.F if not isinstance(self.F, OracleDynamics) else self._F_callable
        return F(state, u, t) if self.step_dependency else F(state, u)

    def _running_cost(self, state, u):
        return self.running_cost(state, u)

    def _add_rollout_var_cost(self):
        """cost_total += rollout_var_cost * sum_t Var_k[c_t] * discount^t (reference :291-292, 310; see _decide_mode).
        The sample axis is the second-to-last of the leading dims ((K) or (E, K)); with a process group the variance is
        taken over all ranks' samples (torch.var's two passes: the mean, then the squared deviations, each all-reduced)."""
        A = self.u_scale * self._perturbed
        kdim = 0 if self.E == 1 else 1
        c = torch.stack([self._running_cost(self._states.select(kdim + 1, t), A.select(kdim + 1, t)) for t in range(self.T)], dim=-1)
        disc = torch.tensor([self.rollout_var_discount**t for t in range(self.T)], dtype=torch.float64, device=c.device)
        if self.pg is None or self.G == 1:
            var = c.var(dim=kdim)  # (T) or (E, T)
        else:
            import torch.distributed as dist

            def allsum(t):
                if dist.get_backend(self.pg) == "gloo" and t.is_cuda:
                    h = t.cpu()
                    dist.all_reduce(h, group=self.pg)
                    return h.to(t.device)
                dist.all_reduce(t, group=self.pg)
                return t

            mean = allsum(c.sum(dim=kdim)) / self.K
            var = allsum(((c - mean.unsqueeze(kdim)) ** 2).sum(dim=kdim)) / (self.K - 1)
        shift = (var * disc).sum(dim=-1) * self.rollout_var_cost  # () or (E)
        self._cost_total += shift if self.E == 1 else shift.unsqueeze(1)

    def _external_cost(self):
        """cost_external: the fused rollout left the states (K, T, nx) and the perturbation cost (:343-344) on the
        device; add the caller's running cost step by step (reference :288-290) and its terminal cost (:306-308)."""
        states, A = self._states, self.u_scale * self._perturbed
        cost = self._cost_total
        for t in range(self.T):
            cost += self._running_cost(states[:, t], A[:, t])
        if self.terminal_state_cost:
            cost += self.terminal_state_cost(states, A)

    def _external_rollout(self, st, per_sample, action_buffer):
        dev, K, T, nu = self.cd, self.K_local, self.T, self.nu
        ab = action_buffer.to(dev)
        x = st.to(dev) if per_sample else st.to(dev).view(1, -1).repeat(K, 1)
        A = self.u_scale * self._perturbed
        hist = torch.cat((ab[1:, :nu].reshape(1, -1, nu).repeat(K, 1, 1), A), dim=1)
        window = ab.shape[0]
        time_buffer = ab[:, nu:].clone() if self.encode_obs_time else None
        cost = torch.zeros(K, dtype=self.dtype, device=dev)
        cost_var = torch.zeros(K, dtype=self.dtype, device=dev)
        states, actions = [], []
        for t in range(T):
            win = hist[:, t : t + window, :]
            if self.encode_obs_time:
                # rolling time-stamp channel of the dataset collector's variant (reference :279-287)
                time_buffer = time_buffer + self.dt
                time_buffer = time_buffer.roll(-1, dims=0)
                time_buffer[-1] = 0
                win = torch.cat((win, time_buffer.view(1, -1, 1).repeat(K, 1, 1)), dim=2)
            x = self._dynamics(x, win, t)
            u = hist[:, t + window - 1, :]
            c = self._running_cost(x, u)
            cost = cost + c
            if self.M > 1:
                cost_var = cost_var + c.var(dim=0) * (self.rollout_var_discount**t)
            states.append(x)
            actions.append(u)
        actions = torch.stack(actions, dim=-2)
        states = torch.stack(states, dim=-2)
        if self.terminal_state_cost:
            cost = cost + self.terminal_state_cost(states, actions)
        cost = cost + cost_var * self.rollout_var_cost  # reference :310
        # action perturbation cost (reference :329-344)
        Ud = self.U.to(dev)
        sig_inv = self.noise_sigma_inv.to(dev)
        eps = torch.abs(self._noise) if self.noise_abs_cost else self._noise
        cost = cost + torch.sum(Ud * (self.lambda_ * eps @ sig_inv), dim=(1, 2))
        self._cost_total.copy_(cost)
        if self._states is not None:
            self._states.copy_(states.reshape(K, T, self.nx))
        if self.torch_path:
            self._actions = actions / self.u_scale  # reference :340

    def refresh_model(self):
        """Not in the reference.  Re-read the dynamics model's weights NOW: after a write the automatic look cannot see at
        once -- ``p.data.copy_()`` / ``p.data.mul_()`` on the model a recognised harness closure closes over (its
        ``(data_ptr, _version)`` key does not move; the content check runs every ``_recognise.TWIN_CONTENT_CHECK_EVERY``
        commands), or on one of this package's own models (``mark_weights_dirty``).  The next command() uploads them."""
        holder = self.F if isinstance(self.F, NLDynamics) else (
            self._candidate[0] if self._candidate is not None and isinstance(self._candidate[0], NLDynamics) else None)
        if holder is None:
            return
        _recognise.refresh_twin(holder.model, force=True)
        if hasattr(holder.model, "mark_weights_dirty"):
            holder.model.mark_weights_dirty()

    # ------------------------------------------------------------------ misc reference API
    def reset(self):
        """Clear controller state after finishing a trial (re-draws U, reference :226-230)."""
        self.U = self.noise_dist.sample(self._lead(self.T))

    def get_rollouts(self, state, num_rollouts=1):
        """Open-loop replay of U through the dynamics callable (reference :358-381)."""
        U = self.U.to(self.cd)
        state = torch.as_tensor(state).to(self.cd, self.dtype).view(-1, self.nx)
        if state.size(0) == 1:
            state = state.repeat(num_rollouts, 1)
        states = torch.zeros((num_rollouts, self.T + 1, self.nx), dtype=self.dtype, device=self.cd)
        states[:, 0] = state
        for t in range(self.T):
            states[:, t + 1] = self._dynamics(
                states[:, t].view(num_rollouts, -1), self.u_scale * U[t].view(num_rollouts, -1), t
            )
        return states[:, 1:].to(self.d)
