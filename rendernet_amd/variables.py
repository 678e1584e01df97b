"""A small variable store standing in for the TF1 graph's variable scopes.

The reference builds its nets with `tf.variable_scope` / `tf.get_variable` (tools/layer_util.py),
which is what gives weights their names (`encoder/res1_3/con1_3X3/weights`, ...).
This module reproduces just that naming machinery so that the mirrored layer builders
(rendernet_amd/tools/layer_util.py) keep the reference signatures, checkpoints keyed by the TF
names load unchanged, and the initialisers are the reference ones:
  xavier_initializer()            tf.contrib.layers.xavier_initializer (uniform)  RenderNet_Shader.py:38
  random_normal_initializer(0.02) tools/layer_util.py:149,188,229,271,313
  constant_initializer(v)         biases 0.001 (tools/layer_util.py:142), slim biases 0, PReLU alpha 0 (:39)
"""
import contextlib
import math
import threading

import numpy as np
import torch


def xavier_initializer():
    def init(shape, rng):
        shape = tuple(int(s) for s in shape)
        if len(shape) >= 2:
            rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
        else:
            fan_in = fan_out = shape[0]
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        return rng.uniform(-lim, lim, size=shape).astype(np.float32)
    return init


def random_normal_initializer(stddev=0.02):
    def init(shape, rng):
        return (rng.standard_normal(size=tuple(shape)) * stddev).astype(np.float32)
    return init


def constant_initializer(value):
    def init(shape, rng):
        return np.full(tuple(shape), value, np.float32)
    return init


class VariableStore:
    """name -> float32 device tensor (TF layout), plus a cache of packed conv filters."""

    def __init__(self, device="cuda", seed=1234):
        self.device = torch.device(device)
        self.rng = np.random.default_rng(seed)
        self.vars = {}
        self._packed = {}
        self._tls = threading.local()         # the scope stack is per thread: two streams / threads may build the same net from one store

    @property
    def _scope(self):
        sc = getattr(self._tls, "scope", None)
        if sc is None:
            sc = self._tls.scope = []
        return sc

    # -- scopes -----------------------------------------------------------------------------
    @contextlib.contextmanager
    def variable_scope(self, name):
        self._scope.append(name)
        try:
            yield
        finally:
            self._scope.pop()

    def full_name(self, name):
        return "/".join(self._scope + [name])

    # -- variables --------------------------------------------------------------------------
    def get_variable(self, name, shape=None, initializer=None):
        full = self.full_name(name)
        v = self.vars.get(full)
        if v is None:
            if isinstance(initializer, (np.ndarray, torch.Tensor)):
                arr = initializer
            else:
                if shape is None or initializer is None:
                    raise KeyError("variable %s does not exist and no shape/initializer given" % full)
                arr = initializer(shape, self.rng)
            v = torch.as_tensor(np.asarray(arr, np.float32) if not isinstance(arr, torch.Tensor) else arr,
                                dtype=torch.float32).to(self.device).contiguous()
            self.vars[full] = v
        elif shape is not None and tuple(v.shape) != tuple(int(s) for s in shape):
            raise ValueError("variable %s has shape %s, requested %s" % (full, tuple(v.shape), tuple(shape)))
        return v, full

    def packed(self, full_name, maker):
        """Cache of PackedWeight objects keyed by variable name (+ packing kind)."""
        p = self._packed.get(full_name)
        if p is None:
            p = maker()
            self._packed[full_name] = p
        return p

    # -- (de)serialisation by TF variable name ----------------------------------------------
    def state_dict(self):
        return {k: v.detach().cpu().numpy() for k, v in self.vars.items()}

    def load_state_dict(self, sd):
        """Keys starting with "__" are not variables (a training checkpoint also holds '__adam_m__', '__adam_v__',
        '__global_step__', '__epoch__' ... -- rendernet_amd/train.py): they are skipped, so the same .npz serves the
        inference entry points without uploading two parameter-sized moment buffers as if they were weights."""
        self._packed.clear()
        for k, v in sd.items():
            if str(k).startswith("__"):
                continue
            self.vars[k] = torch.as_tensor(np.asarray(v, np.float32)).to(self.device).contiguous()

    def num_parameters(self):
        return int(sum(v.numel() for v in self.vars.values()))

    # -- training: one flat parameter buffer ------------------------------------------------
    def flatten(self, names=None):
        """Move the variables (creation order, or `names`) into ONE contiguous float32 buffer, each
        variable 16-byte aligned, and make self.vars[name] views of it.  Returns
        (flat_buffer, {name: (offset, numel)}).  The optimiser (rn_adam_step), the gradient zeroing and
        the gradient all-reduce then work on flat buffers instead of ~230 small tensors.  The packed
        filter cache is dropped because it refers to the old storage."""
        names = list(self.vars.keys()) if names is None else list(names)
        layout, off = {}, 0
        for n in names:
            k = self.vars[n].numel()
            layout[n] = (off, k)
            off += (k + 3) // 4 * 4
        flat = torch.zeros(max(off, 4), dtype=torch.float32, device=self.device)
        for n in names:
            o, k = layout[n]
            view = flat[o:o + k].view(self.vars[n].shape)
            view.copy_(self.vars[n])
            self.vars[n] = view
        self._packed.clear()
        return flat, layout

    def repack_all(self):
        """Refresh every cached packed filter from its TF-layout master (after an optimiser step)."""
        for p in self._packed.values():
            p.repack()


_default = None


def get_default_store():
    global _default
    if _default is None:
        _default = VariableStore("cuda")
    return _default


def set_default_store(store):
    global _default
    _default = store
    return store


def variable_scope(name):
    return get_default_store().variable_scope(name)
