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


class PretrainedMixin:
    """diffusers-layout persistence shared by every mirror (ControlNetModel, ControlNetAdapter, ControlNetRouter):
    `config.json` + `diffusion_pytorch_model.safetensors`, as the reference's call sites expect
    (inference.py:217-254,324-345: `Cls.from_pretrained(repo_or_dir, subfolder=..., torch_dtype=...)`)."""

    _WEIGHT_STEMS = ("diffusion_pytorch_model", "model", "pytorch_model")

    def save_pretrained(self, path, safe_serialization=True, variant=None, **unused):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as fh:
            json.dump(dict(self.config_dict(), _class_name=type(self).__name__), fh, indent=2)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        stem = "diffusion_pytorch_model" + (".%s" % variant if variant else "")
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(path, stem + ".safetensors"))
        else:
            torch.save(sd, os.path.join(path, stem + ".bin"))

    @staticmethod
    def _resolve_dir(path, subfolder, **hub_kw):
        """local directory, or a hub id resolved through huggingface_hub (snapshot of the sub-folder only)"""
        if os.path.isdir(path):
            return os.path.join(path, subfolder) if subfolder else path
        try:
            from huggingface_hub import snapshot_download
        except ImportError as e:       # pragma: no cover
            raise FileNotFoundError("%r is not a directory and huggingface_hub is not installed" % path) from e
        kw = {k: hub_kw[k] for k in ("cache_dir", "revision", "token", "local_files_only", "proxies") if hub_kw.get(k) is not None}
        pats = [(subfolder + "/*") if subfolder else "*"]
        root = snapshot_download(repo_id=path, allow_patterns=pats, **kw)
        return os.path.join(root, subfolder) if subfolder else root

    @classmethod
    def _find_weights(cls, d, variant):
        tried = []
        for stem in cls._WEIGHT_STEMS:
            for var in ([variant] if variant else []) + [None]:
                for ext in (".safetensors", ".bin"):
                    f = os.path.join(d, stem + (".%s" % var if var else "") + ext)
                    tried.append(os.path.basename(f))
                    if os.path.exists(f):
                        return f
        raise FileNotFoundError("no weights file in %s (looked for %s)" % (d, ", ".join(tried)))

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, torch_dtype=None, variant=None, **kwargs):
        import inspect
        d = cls._resolve_dir(str(pretrained_model_name_or_path), subfolder, **kwargs)
        with open(os.path.join(d, "config.json")) as fh:
            raw = json.load(fh)
        accepted = set(inspect.signature(cls.__init__).parameters) - {"self"}
        cfg = {k: v for k, v in raw.items() if not k.startswith("_") and k in accepted}
        wfile = cls._find_weights(d, variant)
        if wfile.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd = load_file(wfile)
        else:
            sd = torch.load(wfile, map_location="cpu", weights_only=True)
        m = cls(**cfg)
        # build the module in the CHECKPOINT's dtype before loading: the packers convert to the MFMA operand format
        # themselves (with an fp16 range check), and the fp32-side parameters (norms, biases, mix factors, router
        # weights) must not take a detour through fp16
        fl = [v.dtype for v in sd.values() if v.is_floating_point()]
        if fl and any(p.dtype != fl[0] for p in m.parameters()):
            m = m.to(fl[0])
        m.load_state_dict(sd)
        if torch_dtype is not None:
            m = m.to(torch_dtype)
        return m


class ParamTreeModule(PretrainedMixin, nn.Module):
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
        self._text_ref, self._text_version, self._text_key, self._text_mode = None, -1, None, 0     # the cache lives in the plan

    # -- step-invariant text K/V cache, opt-in: `module.cache_text = True` --
    cache_text = False

    def _text_cache_mode(self, ehs, set_mode):
        """Chooses keep (1) / reuse (2) for this forward from the identity + version counter of the encoder_hidden_states
        tensor (the same, unmodified tensor object comes back on every denoise step of a request) and tells the plan."""
        mode = 0
        if self.cache_text and ehs is not None:
            key = (tuple(ehs.shape), ehs.dtype, ehs.data_ptr())
            same = (getattr(self, "_text_ref", None) is ehs and self._text_version == ehs._version and self._text_key == key)
            mode = 2 if same else 1
            self._text_ref, self._text_version, self._text_key = ehs, ehs._version, key
        if mode != getattr(self, "_text_mode", 0) or mode:
            L.check(set_mode(self._ensure_plan(), mode))
            self._text_mode = mode
        return mode

    def __del__(self):
        try:
            self._drop_plan()
        except Exception:
            pass

    selection = None      # what plan creation selected for this checkpoint (ctrl_*_selection), set with the plan

    def _note_selection(self, query):
        """reads the plan's selection line; a checkpoint whose normalisation scales have outlier channels takes the conservative (slower)
        selection -- said once, instead of silently running at another speed than the benchmarks"""
        import ctypes
        buf = ctypes.create_string_buffer(256)
        L.check(query(self._plan, buf, 256))
        self.selection = buf.value.decode()
        if "conservative" in self.selection or ("token_stream_fp32_blocks=" in self.selection and "token_stream_fp32_blocks=0" not in self.selection):
            import warnings
            warnings.warn("%s: this checkpoint's normalisation scales have outlier channels: the plan keeps the conservative precision "
                          "selection (%s)" % (type(self).__name__, self.selection))

    WEIGHT_GAIN_WARN = 1.5

    def _warn_on_hot_weights(self, sd):
        """The path rounds GEMM operands to fp16 (conv operands of the ControlNet to a [hi | lo] pair): its error against fp32 grows with
        the GAIN of the weight matrices (std * sqrt(fan_in): every residual branch then outweighs its skip path) -- measured 0.75e-3 rel-inf
        at gain 0.5 and 1.0, 2.0e-3 at gain 2.0 on the SDXL chain (tests/test_gpu_e2e.py::test_weight_distribution_sweep_sdxl_chain; no
        selection of the path changes that, it is the fp16 operand format).  Says so once at plan creation instead of failing silently."""
        gains = [v.detach().float().pow(2).mean().sqrt() * float(v[0].numel()) ** 0.5 for k, v in sd.items() if v.dim() >= 2 and v.numel() >= 4096]
        if not gains:
            return
        med = torch.stack(gains).median().item()
        self.weight_gain = med
        if med > self.WEIGHT_GAIN_WARN:
            import warnings
            warnings.warn("%s: median weight gain %.2f (std * sqrt(fan_in)) is above %.1f -- the fp16-operand path is validated to 1e-3 "
                          "rel-inf for gains <= 1; expect ~2e-3 at gain 2" % (type(self).__name__, med, self.WEIGHT_GAIN_WARN))

    def _tensor_refs(self):
        sd = {k: v for k, v in self.named_parameters()}
        self._warn_on_hot_weights(sd)
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
