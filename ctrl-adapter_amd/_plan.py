"""Shared plumbing of the nn.Module mirrors: parameter trees built from the C-side inventory
(ctrl_*_param_spec is the single source of truth for names/shapes = the reference's state-dict keys), plan
lifetime, and diffusers-layout persistence (config.json + diffusion_pytorch_model.safetensors)."""
import ctypes as C
import json
import os

import torch
from torch import nn

from . import _lib as L


def c_spec(count_fn, spec_fn, cfg):
    n = count_fn(C.byref(cfg))
    if n < 0:
        raise ValueError("libctrlhip rejected the configuration: %s" % L.lib().ctrl_last_error().decode())
    out = []
    name = C.create_string_buffer(256)
    shape = (C.c_int64 * 6)()
    nd = C.c_int()
    for i in range(n):
        L.check(spec_fn(C.byref(cfg), i, name, 256, shape, C.byref(nd)))
        out.append((name.value.decode(), tuple(shape[k] for k in range(nd.value))))
    return out


class ParamTreeModule(nn.Module):
    """nn.Module whose parameters are registered under dotted names (nested plain containers), so that
    state_dict() keys equal the reference module tree's keys."""

    def __init__(self):
        super().__init__()
        self._plan = None
        self._plan_device = None

    def _register_spec(self, spec, dtype=torch.float16):
        for name, shape in spec:
            parts = name.split(".")
            m = self
            for p in parts[:-1]:
                if p not in m._modules:
                    m.add_module(p, nn.Module())
                m = m._modules[p]
            m.register_parameter(parts[-1], nn.Parameter(torch.zeros(shape, dtype=dtype), requires_grad=False))
        self._spec = spec

    # -- plan lifetime: any parameter movement / reload invalidates the packed copy --
    def _apply(self, fn, *a, **k):
        self._drop_plan()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._drop_plan()
        return super().load_state_dict(*a, **k)

    def _drop_plan(self):
        if getattr(self, "_plan", None) is not None:
            self._destroy(self._plan)
        self._plan = None

    def __del__(self):
        try:
            self._drop_plan()
        except Exception:
            pass

    def _tensor_refs(self):
        sd = {k: v for k, v in self.named_parameters()}
        refs = (L.TensorRef * len(sd))()
        keep = []
        for i, (k, v) in enumerate(sd.items()):
            if not v.is_cuda:
                raise RuntimeError("parameters must live on the GPU before the first forward (call .to('cuda')); "
                                   "there is no CPU path")
            t = v.detach().contiguous()
            keep.append(t)
            kb = k.encode()
            keep.append(kb)
            refs[i].name = kb
            refs[i].data = t.data_ptr()
            refs[i].dtype = L.dtype_code(t.dtype)
            refs[i].ndim = t.dim()
            for d in range(t.dim()):
                refs[i].shape[d] = t.shape[d]
        return refs, len(sd), keep

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    # -- diffusers-layout persistence --
    def save_pretrained(self, path):
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as fh:
            json.dump(dict(self.config_dict(), _class_name=type(self).__name__), fh, indent=2)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()},
                  os.path.join(path, "diffusion_pytorch_model.safetensors"))

    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=None, **unused):
        from safetensors.torch import load_file
        if subfolder:
            path = os.path.join(path, subfolder)
        with open(os.path.join(path, "config.json")) as fh:
            cfg = {k: v for k, v in json.load(fh).items() if not k.startswith("_")}
        m = cls(**cfg)
        m.load_state_dict(load_file(os.path.join(path, "diffusion_pytorch_model.safetensors")))
        if torch_dtype is not None:
            m = m.to(torch_dtype)
        return m


class Config(dict):
    """attribute-style access like diffusers' FrozenDict (pipelines read e.g. controlnet.config.global_pool_conditions)"""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def timesteps_to_device_f32(timestep, n, device):
    """python number / 0-d / 1-d (len 1 or n) / [b, F] tensor -> fp32 device tensor of 1 or n elements.
    Mirrors controlnet/controlnet.py:735-749 and model/adapter_spatial_temporal.py:190-197, except that the value
    stays fp32 (the reference rounds it to the hidden dtype, SURVEY.md note N3)."""
    if not torch.is_tensor(timestep):
        t = torch.tensor([float(timestep)], dtype=torch.float32, device=device)
    else:
        t = timestep.detach().to(device=device, dtype=torch.float32).reshape(-1)
    if t.numel() not in (1, n):
        raise ValueError("timestep must have 1 or %d elements, got %d" % (n, t.numel()))
    return t.contiguous()
