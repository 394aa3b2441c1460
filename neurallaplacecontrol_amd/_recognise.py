"""Recognising the reference harness's LITERAL closures.

The reference hands ``MPPIDelay`` two Python callables built in ``mppi_with_model.py``:

* ``dynamics`` -- a local function closing over ``model`` and ``ts_pred`` (``state + model(state, window, ts_pred)``,
  :103-122), or ``functools.partial(<env>_dynamics_dt_delay, ts=ts_pred, delay=..., friction=...)`` (:129-143);
* ``running_cost`` -- a local function closing over ``env`` (``-(env.diff_obs_reward_(state, exp_reward=False) +
  env.diff_ac_reward_(action))`` on its default branch, :145-171).

A closure is opaque to a HIP kernel, but its free variables are not: this module looks at ``__closure__`` /
``functools.partial`` and proposes the fused objects of :mod:`.envs` (``NLDynamics`` / ``OracleDynamics`` / ``EnvCost``)
that WOULD compute the same thing.  A proposal is only a candidate: ``MPPIDelay`` verifies it at its first ``command()``
by running one short command both ways on the command's own state and action buffer and switches to the fused path only
if rollout states and costs agree (``probe_equivalence``); otherwise it stays on the generic path, silently correct.
A probe alone cannot establish equivalence (one state, a short horizon: a clamp, a termination branch or any
state-dependent logic outside the probed region would go unseen -- ADVICE r3), so a closure is only ever a candidate if it
also has the harness closures' STRUCTURE: its code object may reference nothing but the names those closures reference
(``_DYN_NAMES`` / ``_COST_NAMES``: the torch calls that append the time channel; the env's two reward methods), close over
nothing but their free variables, and carry no constants of its own beyond None / booleans / small integers / the strings
they use -- a closure with a ``clamp``, a ``where``, a threshold constant or a helper call fails this and stays on the
generic path.  Names and constants say nothing about OPERATORS (``state - model(...)``, ``out[out[:, 0] > 2] = 2``
use none), so the bytecode is held to the harness closures' operations as well (``_ops_ok``): every instruction must come
from a small set of opcodes (loads, calls, the branch on the closure's flags, store to a local, return) plus exactly the
arithmetic those closures do -- ONE ``+`` and no other binary operator in the dynamics closure, one ``+`` per reward
branch and one unary minus in the cost closure, one ``== "nl"`` comparison at most, subscripts only by a small constant
index (``perturbed_action.shape[0]``), no subscript or attribute store, no ``in`` / ``is`` / ``not``, no loops.
``planner_options={"recognise_closures": 0}`` switches recognition off altogether.

A twin built from a reference model instance (``NeuralLaplaceModel.from_reference``) is a weight SNAPSHOT; the dynamics
object keeps the source module and re-copies its weights whenever their ``(data_ptr, _version)`` key has moved since the last
command (``refresh_twin``), so a ``load_state_dict`` / optimizer step on the closed-over model is seen by the fused path as
it is by the literal closure.  A write through ``.data`` moves no version counter: it is caught by a per-tensor content
check every ``TWIN_CONTENT_CHECK_EVERY`` commands (so the fused path may lag such a write by up to that many commands), or
at once by ``MPPIDelay.refresh_model()``.
Nothing here executes or imports reference code: only attribute and type inspection of objects the caller passed in.
"""

import dis
import functools
import inspect

import torch

from .envs import EnvCost, NLDynamics, OracleDynamics

_ORACLE_FUNCS = {
    "cartpole_dynamics_dt_delay": "oderl-cartpole",
    "pendulum_dynamics_dt_delay": "oderl-pendulum",
    "acrobot_dynamics_dt_delay": "oderl-acrobot",
}
_ENV_CLASS_HINTS = (("cartpole", "oderl-cartpole"), ("pendulum", "oderl-pendulum"), ("acrobot", "oderl-acrobot"))


# What the harness's closures reference (mppi_with_model.py:103-122 and :145-171): global / attribute names, free variables,
# argument names.  A structural fingerprint -- names only, no code.
_DYN_NAMES = frozenset({"torch", "cat", "flip", "arange", "view", "repeat", "shape"})
_DYN_FREE = frozenset({"model", "ts_pred", "device"})
_DYN_ARGS = ("state", "perturbed_action")
_DYN_STRS = frozenset({"nl", "device", "dim"})  # the model name it compares with, keyword names of its torch calls
_COST_NAMES = frozenset({"diff_obs_reward_", "diff_ac_reward_", "change_goal_flipped"})
_COST_FREE = frozenset({"env", "state_constraint", "change_goal"})
_COST_STRS = frozenset({"exp_reward", "state_constraint", "change_goal", "change_goal_flipped"})


def _consts_ok(consts, strings, doc):
    for c in consts:
        if c is None or isinstance(c, bool):
            continue
        if isinstance(c, int) and -4 <= c <= 4:
            continue
        if isinstance(c, str) and (c in strings or c == doc):
            continue  # a keyword / model name the harness closure uses, or the function's own docstring
        if isinstance(c, tuple) and all((isinstance(e, str) and e in strings) or (type(e) is int and -4 <= e <= 4) for e in c):
            continue  # keyword-name tuple of a call / a small index tuple
        return False
    return True


# Opcodes the harness closures compile to, CPython 3.8 .. 3.13 spellings (an opcode outside this set -- a subscript store, a
# comparison chain, an in-place operator, a loop, an import, ... -- is something those closures do not do).  Arithmetic,
# comparisons and subscripts are in the set but COUNTED and checked separately below.
_PLAIN_OPS = frozenset({
    "RESUME", "CACHE", "NOP", "EXTENDED_ARG", "COPY_FREE_VARS", "MAKE_CELL", "PUSH_NULL", "PRECALL", "KW_NAMES", "POP_TOP",
    "LOAD_FAST", "LOAD_FAST_CHECK", "LOAD_FAST_LOAD_FAST", "LOAD_DEREF", "LOAD_CLOSURE", "LOAD_GLOBAL", "LOAD_CONST", "LOAD_ATTR",
    "LOAD_METHOD", "STORE_FAST", "STORE_FAST_LOAD_FAST", "BUILD_TUPLE", "CALL", "CALL_KW", "CALL_FUNCTION", "CALL_FUNCTION_KW",
    "CALL_METHOD", "RETURN_VALUE", "RETURN_CONST", "JUMP_FORWARD", "JUMP_ABSOLUTE", "POP_JUMP_IF_FALSE", "POP_JUMP_IF_TRUE",
    "POP_JUMP_FORWARD_IF_FALSE", "POP_JUMP_FORWARD_IF_TRUE", "JUMP_IF_FALSE_OR_POP", "JUMP_IF_TRUE_OR_POP", "TO_BOOL", "COPY",
})


def _ops_ok(code, kind):
    """The operation fingerprint (module docstring): every instruction is a plain load / call / branch / local store, and the
    arithmetic is exactly the harness closure's."""
    adds = negs = 0
    compares, subscripts = [], []
    prev = None
    for ins in dis.get_instructions(code):
        op = ins.opname
        if op == "BINARY_ADD" or (op == "BINARY_OP" and ins.argrepr == "+"):
            adds += 1
        elif op == "UNARY_NEGATIVE":
            negs += 1
        elif op == "COMPARE_OP":
            # (argrepr "==" up to 3.12, "bool(==)" from 3.13) against the constant loaded just before it
            compares.append((ins.argrepr.replace("bool(", "").replace(")", ""), prev.argval if prev is not None and prev.opname == "LOAD_CONST" else None))
        elif op == "BINARY_SUBSCR":
            subscripts.append(prev.argval if prev is not None and prev.opname == "LOAD_CONST" else None)
        elif op not in _PLAIN_OPS:
            return False  # BINARY_SUBTRACT / BINARY_OP other than +, STORE_SUBSCR, CONTAINS_OP, IS_OP, UNARY_NOT, FOR_ITER, ...
        if op not in ("CACHE", "EXTENDED_ARG"):
            prev = ins
    small = lambda v: type(v) is int and 0 <= v <= 3  # noqa: E731
    if kind == "dynamics":
        return (adds == 1 and negs == 0 and all(c == ("==", "nl") for c in compares) and len(compares) <= 1
                and all(small(i) for i in subscripts) and len(subscripts) <= 1)
    # (one `+` per reward branch; the compiler may copy the function's tail -- the negation -- into every branch)
    return 1 <= adds <= 3 and 1 <= negs <= adds and not compares and not subscripts


def has_harness_structure(fn, kind):
    """True iff the function `fn` references nothing the harness's `kind` closure ("dynamics" / "cost") does not -- names, free
    variables, leading argument names, constants -- and performs no operation it does not (see the module docstring)."""
    if not inspect.isfunction(fn):
        return False
    code = fn.__code__
    names, free, strs = (_DYN_NAMES, _DYN_FREE, _DYN_STRS) if kind == "dynamics" else (_COST_NAMES, _COST_FREE, _COST_STRS)
    if not set(code.co_names) <= names or not set(code.co_freevars) <= free:
        return False
    if kind == "dynamics" and tuple(code.co_varnames[:2]) != _DYN_ARGS:
        return False
    if kind == "cost" and code.co_argcount != 2:
        return False
    if any(inspect.iscode(c) for c in code.co_consts):  # nested functions / comprehensions: not the harness's closure
        return False
    return _consts_ok(code.co_consts, strs, fn.__doc__) and _ops_ok(code, kind)


def _module_key(mod):
    return tuple((t.data_ptr(), t._version) for t in list(mod.parameters()) + list(mod.buffers()))


TWIN_CONTENT_CHECK_EVERY = 32  # commands between two content checks of a twin's source (writes through .data)


def _module_checksum(mod):
    """One number per tensor (its float64 sum), gathered in ONE host transfer: catches a write that bypassed the version
    counter (``p.data.mul_()``, ``p.data.copy_()`` -- EMA / target-network code, some hand-written optimisers)."""
    ts = [t.detach() for t in list(mod.parameters()) + list(mod.buffers())]
    if not ts:
        return ()
    return tuple(torch.stack([t.to(torch.float64).sum().to("cpu") for t in ts]).tolist())


def refresh_twin(twin, force=False):
    """Re-copy the weights of a twin built by ``_model_twin`` from a foreign (reference) module if that module's tensors
    have been written or replaced since the last look.  No-op for the package's own models.
    The look is the tensors' ``(data_ptr, _version)`` key at every command; a write through ``.data`` does not move that
    key, so every ``TWIN_CONTENT_CHECK_EVERY``-th look also compares per-tensor sums, and ``force=True`` --
    ``MPPIDelay.refresh_model()`` -- copies unconditionally."""
    src = twin.__dict__.get("_twin_source")
    if src is None:
        return
    key = _module_key(src)
    looks = twin.__dict__["_twin_looks"] = twin.__dict__.get("_twin_looks", 0) + 1
    moved = force or key != twin.__dict__["_twin_source_key"]
    if not moved and looks % TWIN_CONTENT_CHECK_EVERY == 0:
        moved = _module_checksum(src) != twin.__dict__.get("_twin_source_sum")
    if moved:
        with torch.no_grad():
            for name in ("state_mean", "state_std", "action_mean", "action_std", "dt"):
                if hasattr(src, name):
                    getattr(twin, name).copy_(getattr(src, name).detach().to(getattr(twin, name).device))
            twin.load_state_dict({k: v.detach() for k, v in src.state_dict().items()})
        twin.__dict__["_twin_source_key"] = key
        twin.__dict__["_twin_source_sum"] = _module_checksum(src)
        if hasattr(twin, "mark_weights_dirty"):
            twin.mark_weights_dirty()  # (load_state_dict moved the twin's own key already; a forced copy of equal values did not)


def _free_variables(fn):
    """name -> value of a function's closure cells (empty cells skipped)."""
    out = {}
    if not inspect.isfunction(fn) or fn.__closure__ is None:
        return out
    for name, cell in zip(fn.__code__.co_freevars, fn.__closure__):
        try:
            out[name] = cell.cell_contents
        except ValueError:  # empty cell
            pass
    return out


def _constant_scalar(v):
    """float(v) if `v` is a number or a tensor / array whose entries are all equal, else None."""
    if isinstance(v, bool):
        return None
    if isinstance(v, (int, float)):
        return float(v)
    try:
        t = torch.as_tensor(v).detach().reshape(-1)
    except Exception:
        return None
    if t.numel() == 0 or not (t.dtype.is_floating_point or t.dtype in (torch.int32, torch.int64)):
        return None
    t = t.to("cpu", torch.float64)
    return float(t[0]) if bool((t == t[0]).all()) else None


def _model_twin(obj):
    """The package's model mirror for `obj`: the object itself if it is one of ours, a converted twin if it looks like
    the reference's ``w_nl.NeuralLaplaceModel`` (same sub-modules / attributes), else None."""
    from .nl_model import NeuralLaplaceModel
    from .node_model import NODE
    from .rnn_model import RNN, DeltaTRNN

    if isinstance(obj, (NeuralLaplaceModel, DeltaTRNN, RNN, NODE)):
        return obj
    if isinstance(obj, torch.nn.Module) and type(obj).__name__ == "NeuralLaplaceModel" and hasattr(obj, "laplace_rep_func"):
        try:
            twin = NeuralLaplaceModel.from_reference(obj)
        except Exception:
            return None
        # a snapshot: remember where it came from, so that later weight updates of `obj` reach the fused path (refresh_twin)
        # (through __dict__: a plain attribute assignment would register `obj` as a sub-module of the twin)
        twin.__dict__["_twin_source"], twin.__dict__["_twin_source_key"] = obj, _module_key(obj)
        twin.__dict__["_twin_source_sum"] = _module_checksum(obj)
        return twin
    return None


def candidate_dynamics(fn):
    """An ``NLDynamics`` / ``OracleDynamics`` that `fn` appears to compute, or None."""
    if isinstance(fn, (NLDynamics, OracleDynamics)):
        return fn
    if isinstance(fn, functools.partial):
        env = _ORACLE_FUNCS.get(getattr(fn.func, "__name__", ""))
        if env is None or fn.args:
            return None
        kw = dict(fn.keywords or {})
        ts = _constant_scalar(kw.pop("ts", None))
        delay = kw.pop("delay", None)
        friction = bool(kw.pop("friction", False))
        if ts is None or not isinstance(delay, int) or kw:
            return None
        return OracleDynamics(env, ts=ts, delay=delay, friction=friction)
    if not has_harness_structure(fn, "dynamics"):
        return None
    free = _free_variables(fn)
    if not free:
        return None
    models = [(n, m) for n, m in ((n, _model_twin(v)) for n, v in free.items()) if m is not None]
    if len(models) != 1:
        return None
    # the prediction time: the free variable the harness calls ts_pred, else the only constant-valued tensor / number
    if "ts_pred" in free:
        ts = _constant_scalar(free["ts_pred"])
    else:
        consts = [c for c in (_constant_scalar(v) for n, v in free.items() if n != models[0][0] and not isinstance(v, (str, torch.device))) if c is not None]
        ts = consts[0] if len(consts) == 1 else None
    if ts is None or not (ts > 0.0):
        return None
    try:
        return NLDynamics(models[0][1], ts)
    except Exception:
        return None


def candidate_cost(fn):
    """An ``EnvCost`` that `fn` appears to compute (the harness closure's default branch), or None."""
    if isinstance(fn, EnvCost):
        return fn
    if not has_harness_structure(fn, "cost"):
        return None
    free = _free_variables(fn)
    if not free:
        return None
    # non-default branches (state_constraint / change_goal) are NOT the plain env cost
    for flag in ("state_constraint", "change_goal"):
        if free.get(flag):
            return None
    envs = [v for v in free.values() if hasattr(v, "diff_obs_reward_") and hasattr(v, "diff_ac_reward_")]
    if len(envs) != 1:
        return None
    cls = type(envs[0]).__name__.lower()
    for hint, name in _ENV_CLASS_HINTS:
        if hint in cls:
            return EnvCost(name)
    return None


def probe_equivalence(make_planner, literal, candidate, state, action_buffer, K, nu, horizon=4, rtol=1e-9, atol=1e-10):
    """One short command on the command's own state / action buffer through the literal callables (generic path) and
    through the candidate objects (fused path), on the SAME noise draw (a private generator: the caller's RNG stream is
    not touched).  True iff rollout states and total costs agree.  Any exception on either side -> False."""
    try:
        gen = torch.Generator().manual_seed(0x5EED)
        raw = torch.randn(K, horizon, nu, dtype=torch.float64, generator=gen) * 0.5
        out = []
        for dyn, cost in (literal, candidate):
            p = make_planner(dyn, cost, horizon)
            p.noise_dist = type("Replay", (), {"sample": staticmethod(lambda shape, raw=raw: raw.to(p.d))})()
            with torch.no_grad():
                p.command(state, action_buffer)
            out.append((p.states.detach().cpu(), p.cost_total.detach().cpu()))
        (s0, c0), (s1, c1) = out
        if s0.shape != s1.shape or not bool(torch.isfinite(s0).all()) or not bool(torch.isfinite(c0).all()):
            return False
        return bool(torch.allclose(s0, s1, rtol=rtol, atol=atol)) and bool(torch.allclose(c0, c1, rtol=rtol, atol=atol))
    except Exception:
        return False
