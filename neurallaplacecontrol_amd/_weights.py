"""Change detection for the weights a model mirror has uploaded into an ``nlc_ctx``.

A planner asks ``weights_key(model)`` once per ``command()`` (``MPPIDelay._ensure_configured``) and re-uploads
(``nlc_set_model``) when the key differs from the one it uploaded at.  The key is ``(data_ptr, _version)`` of every
parameter and buffer:

* an in-place write (``p.mul_(2)`` under ``no_grad``, an optimizer step, ``load_state_dict``) bumps ``_version``;
* a replaced tensor (``lin.weight = nn.Parameter(...)``, ``model.state_std = t``, ``.to()/.double()/.cuda()``) is a new
  registration: torch's global parameter / buffer registration hooks bump a process-wide epoch here, and the cached
  tensor list is re-collected when the epoch moved (so the per-command cost stays a 20-tuple build, ~4 us);
* NOT detectable: a write through ``p.data`` (``p.data.mul_(2)``) -- ``.data`` is an alias with its own version counter.
  Call ``model.mark_weights_dirty()`` after such a write.
"""

import torch.nn.modules.module as _tm

_EPOCH = [0]


def _bump(*_args):
    _EPOCH[0] += 1
    return None


_tm.register_module_parameter_registration_hook(_bump)
_tm.register_module_buffer_registration_hook(_bump)
_tm.register_module_module_registration_hook(_bump)


class WeightsKeyMixin:
    """``_weights_key()`` / ``mark_weights_dirty()`` for the nn.Module mirrors (NL model, Delta-t RNN / RNN, NODE)."""

    _wk_epoch = -1
    _wk_tensors = None
    _wk_dirty = 0

    def _weights_key_extra(self):
        return ()

    def _apply(self, fn, *args, **kwargs):
        # .float().double() / .half().double() / .to(...) re-allocate through `param.data = fn(param.data)`: the Parameter
        # object and its version counter survive and the caching allocator may hand back the very block just freed, so
        # (data_ptr, _version) can stay equal although the values were rounded.  Conversions are rare: bump.
        self._wk_dirty += 1
        return super()._apply(fn, *args, **kwargs)

    def mark_weights_dirty(self):
        """Force the next planner command / forward to re-upload the weights (needed after a ``.data`` write)."""
        self._wk_dirty += 1

    def _weights_key(self):
        if self._wk_tensors is None or self._wk_epoch != _EPOCH[0]:
            self._wk_tensors = [p for p in self.parameters()] + [b for b in self.buffers()]
            self._wk_epoch = _EPOCH[0]
        return tuple([(t.data_ptr(), t._version) for t in self._wk_tensors]) + (self._wk_dirty,) + tuple(self._weights_key_extra())
