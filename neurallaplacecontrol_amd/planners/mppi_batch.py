"""Many independent MPPI problems planned side by side on one MI355X.

The reference's expert-data collector runs 5 000 episodes x 200 ``MPPIDelay.command()`` calls per (env, delay)
with K = 1000 samples (``mppi_dataset_collector.py:224-321,402-424``; ``config.py:17,21-23``), one episode per
worker process.  A K = 1000 command is far too small to fill 256 CUs, so :class:`BatchedMPPIDelay` plans E episodes
per ``command()``: every episode keeps its own state, action buffer, control sequence ``U`` and K samples, exactly
as E separate :class:`~neurallaplacecontrol_amd.planners.mppi_delay.MPPIDelay` objects would, but each kernel
launch covers all E*K samples (``nlc_mppi_desc.E``; per-episode softmax weights through ``blockIdx.y``).

Episode e of a batched command is bit-identical to a single ``MPPIDelay.command()`` fed the same noise
(``tests/test_gpu_batched_sharded.py::test_batched_planner_*``).
"""

import ctypes as C

import torch

from .. import _lib
from .mppi_delay import MPPIDelay


class BatchedMPPIDelay(MPPIDelay):
    """``MPPIDelay`` over ``num_envs`` episodes.  Same constructor arguments plus ``num_envs``; only the fused
    dynamics (``NLDynamics`` / ``OracleDynamics`` with an ``EnvCost``) are supported.

    * ``command(states (E, nx), action_buffers (E, B, nu)) -> actions (E, nu)``; the inputs may be host or device
      tensors (device tensors are not copied through the host)
    * ``U`` is ``(E, T, nu)``; ``noise``, ``states`` ... gain a leading E; ``reset(env_ids=None)`` re-draws ``U`` of
      the listed episodes (all by default), the collector's per-episode ``mppi_gym.reset()`` (:236)
    * ``noise_rng="torch"`` draws ``noise_dist.sample((E, K, T))`` on ``device`` (episode-major), ``"philox"`` draws on
      the GPU; episode e uses sample indices ``e*K + k`` of the same (seed, command counter) stream
    """

    def __init__(self, dynamics, running_cost, nx, noise_sigma, num_envs, *args, **kwargs):
        self.E = int(num_envs)
        if self.E < 1:
            raise ValueError("num_envs must be >= 1")
        if kwargs.get("process_group") is not None:
            raise NotImplementedError("shard the episodes over ranks (one BatchedMPPIDelay per GPU), not the samples")
        super().__init__(dynamics, running_cost, nx, noise_sigma, *args, **kwargs)
        if not self.fused:
            raise NotImplementedError("BatchedMPPIDelay needs NLDynamics / OracleDynamics and an EnvCost")
        if self.E == 1:
            raise ValueError("num_envs == 1: use MPPIDelay")

    def command(self, state, action_buffer):
        E, K, T, nu = self.E, self.K, self.T, self.nu
        st = torch.as_tensor(state)
        ab = torch.as_tensor(action_buffer)
        if tuple(st.shape) == (E, K, self.nx):
            per_sample = True
        elif tuple(st.shape) == (E, self.nx):
            per_sample = False
        else:
            raise ValueError(f"state must be (E={E}, nx={self.nx}) or (E, K, nx)")
        if ab.dim() != 3 or ab.shape[0] != E:
            raise ValueError("action_buffer must be (E, B, nu)")
        self._state_in = st

        def stage(t):  # device tensors stay where they are; host tensors are read by the library directly
            t = t.detach().to(dtype=torch.float64)
            if t.is_cuda and t.device != self.cd:
                t = t.to(self.cd)
            return t.contiguous()

        with torch.cuda.device(self.cd):
            self.ctx.use_torch_stream()  # (re)configuration uploads are ordered on the command's stream
            self._ensure_configured(ab.shape[1])
        if self.encode_obs_time and ab.shape[2] == nu + 1:
            ab = ab[:, :, :nu]  # drop the time-stamp column (mppi_delay.py:262-264)
        if ab.shape[2] != nu:
            raise ValueError("action_buffer must have nu columns")
        st, ab = stage(st), stage(ab)
        lib, ctx = self.ctx.lib, self.ctx
        rng = 1 if self.noise_rng == "philox" else 0
        with torch.cuda.device(self.cd):
            ctx.use_torch_stream()
            if not rng:
                self._noise.copy_(self.noise_dist.sample((E, K, T)).reshape(E, K, T, nu))
            ctx.check(
                lib.nlc_mppi_rollout(
                    ctx.h, _lib.ptr(st), int(per_sample), _lib.ptr(ab), C.byref(self._buf), rng, self.seed, self._commands
                )
            )
            self._commands += 1
            if self.d.type == "cuda":
                ctx.check(lib.nlc_mppi_finish(ctx.h, _lib.ptr(self._partials), 1, 0, C.byref(self._buf), None))
                act = self._action.clone()
            else:
                act = torch.empty(E, self.u_per_command * nu, dtype=torch.float64)
                ctx.check(lib.nlc_mppi_finish(ctx.h, _lib.ptr(self._partials), 1, 0, C.byref(self._buf), _lib.ptr(act)))
            if self.M > 1 and self.rollout_var_cost != 0:
                self._add_rollout_var_cost()  # one variance per episode (reference :291-292, 310)
        action = act.view(E, self.u_per_command, nu)
        if self.u_per_command == 1:
            action = action[:, 0]
        return action if self.d.type == action.device.type else action.to(self.d)

    def reset(self, env_ids=None):
        """Re-draw the control sequence of the listed episodes (all when None), reference :226-230."""
        if env_ids is None:
            self.U = self.noise_dist.sample((self.E, self.T))
            return
        U = self.U
        ids = torch.as_tensor(env_ids, dtype=torch.long).reshape(-1)
        U[ids] = self.noise_dist.sample((ids.numel(), self.T)).to(U.dtype)
        self.U = U

    def get_rollouts(self, state, num_rollouts=1):
        raise NotImplementedError("get_rollouts is per-episode: use MPPIDelay")
