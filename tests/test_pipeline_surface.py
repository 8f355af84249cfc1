"""Everything the reference's pipelines touch on the three modules besides calling them (attribute reads and isinstance
checks, collected from sdxl/pipelines/sdxl_controlnet_adapter_pipeline.py and
i2vgen_xl/pipelines/i2vgen_xl_controlnet_adapter_pipeline.py) exists on the drop-in classes with the reference's meaning.
The pipelines themselves cannot run here (diffusers is not installable offline); this is the static part of the boundary
(SURVEY.md 8b).  CPU only: constructing the modules needs no GPU."""
import inspect

import pytest
import torch

from helpers import ROOT  # noqa: F401
import cases
import ctrl_adapter_amd as P


def test_controlnet_attributes_the_pipelines_read():
    cn = P.ControlNetModel(cross_attention_dim=768)
    # sdxl pipeline :1154 / i2vgen pipeline :762,:780 -- prepare_image(..., dtype=controlnet.dtype)
    assert cn.dtype == torch.float16
    # :1037-1042 / :707-711 -- guess_mode = guess_mode or controlnet.config.global_pool_conditions
    assert cn.config.global_pool_conditions is False
    assert P.ControlNetModel(cross_attention_dim=768, global_pool_conditions=True).config.global_pool_conditions is True
    # :1029,:1035 / :675,:705 -- len(controlnet.nets) and controlnet.nets[0].config... for the multi form; isinstance checks
    multi = P.MultiControlNetModel([cn, P.ControlNetModel(cross_attention_dim=768)])
    assert len(multi.nets) == 2 and multi.nets[0].config.global_pool_conditions is False
    assert isinstance(multi, P.MultiControlNetModel) and not isinstance(multi, P.ControlNetModel)
    # :1021 / :667 -- `self.controlnet._orig_mod if is_compiled_module(self.controlnet)`: never a compiled module
    assert not hasattr(cn, "_orig_mod")
    # the call keywords the pipelines pass (sdxl :1323-1335 / i2vgen :957-970 and MultiControlNetModel.forward, multicontrolnet.py:45-64)
    sig = inspect.signature(P.ControlNetModel.forward).parameters
    for kw in ("sample", "timestep", "encoder_hidden_states", "controlnet_cond", "conditioning_scale", "guess_mode",
               "added_cond_kwargs", "return_dict", "skip_conv_in", "skip_time_emb"):
        assert kw in sig, kw
    msig = inspect.signature(P.MultiControlNetModel.forward).parameters
    for kw in ("sample", "timestep", "encoder_hidden_states", "controlnet_cond", "conditioning_scale", "guess_mode", "return_dict",
               "skip_conv_in", "skip_time_emb"):
        assert kw in msig, kw


def test_adapter_and_router_attributes_the_pipelines_read():
    ad = P.ControlNetAdapter(**cases.ADAPTER_SDXL)
    # sdxl :1339, i2vgen :1037,:1043 -- residuals are cast `.to(self.adapter.dtype)` before the call
    assert ad.dtype == torch.float16
    asig = inspect.signature(P.ControlNetAdapter.forward).parameters
    for kw in ("down_block_res_samples", "mid_block_res_sample", "sparsity_masking", "num_frames", "timestep", "encoder_hidden_states"):
        assert kw in asig, kw            # model/ctrl_adapter.py:170-171, called at sdxl :1338-1345 / i2vgen :1042-1049
    r = P.ControlNetRouter(num_experts=3, router_type="simple_weights", num_routers=12)
    # i2vgen :974-997 -- dispatch on router_type, `.to(self.router.dtype)`, num_routers / num_experts for the merge
    assert r.router_type == "simple_weights" and r.num_routers == 12 and r.num_experts == 3 and r.dtype == torch.float32
    rsig = inspect.signature(P.ControlNetRouter.forward).parameters
    assert list(rsig)[1:] == ["router_input", "sparse_mask", "fixed_weights"]        # model/ctrl_router.py:84
    # the router types the pipeline branches on but model/ctrl_router.py does not implement fail at construction, loudly
    for rt in ("timestep_weights", "embedding_weights", "timestep_embedding_weights"):
        with pytest.raises(ValueError):
            P.ControlNetRouter(num_experts=3, router_type=rt)
    eq = P.ControlNetRouter(num_experts=2, router_type="equal_weights")
    assert eq.router_type == "equal_weights" and len(list(eq.parameters())) == 0


def test_module_protocol_used_by_inference_py():
    """inference.py:218-247,337 loads with from_pretrained(..., torch_dtype=) and moves with .to(device): both exist on every module,
    and eval() / requires_grad_(False) (train-side freezing) are nn.Module's"""
    for cls in (P.ControlNetModel, P.ControlNetAdapter, P.ControlNetRouter):
        assert callable(getattr(cls, "from_pretrained")) and callable(getattr(cls, "save_pretrained"))
        assert issubclass(cls, torch.nn.Module)
        assert "torch_dtype" in inspect.signature(cls.from_pretrained).parameters or \
            any(p.kind == p.VAR_KEYWORD for p in inspect.signature(cls.from_pretrained).parameters.values())


def test_loading_calls_of_inference_py(tmp_path):
    """the literal call forms of inference.py:218-252 -- from_pretrained(<dir>, subfolder=..., low_cpu_mem_usage=False,
    device_map=None), then .to(data_type) / .eval() -- on checkpoints laid out like the training script writes them
    (<root>/adapter_<step>/, <root>/router_<step>/), incl. data_type = bfloat16 (the script's --mixed_precision bf16)"""
    from oracle.init import seeded_init
    cfg = dict(cases.ADAPTER_SDXL)
    cfg.update(add_adapter_location_B=False, add_adapter_location_C=False)          # one location: a small file
    ad = seeded_init(P.ControlNetAdapter(**cfg), seed=5)
    ad.save_pretrained(str(tmp_path / "adapter_70000"))
    r = seeded_init(P.ControlNetRouter(num_experts=3, router_type="simple_weights", num_routers=12), seed=6)
    r.save_pretrained(str(tmp_path / "router_70000"))
    for data_type in (torch.float32, torch.half, torch.bfloat16):
        a2 = P.ControlNetAdapter.from_pretrained(str(tmp_path), subfolder="adapter_70000", low_cpu_mem_usage=False, device_map=None)
        a2 = a2.to(data_type)
        a2.eval()
        assert a2.dtype == data_type and sorted(a2.state_dict()) == sorted(ad.state_dict())
        k = next(iter(ad.state_dict()))
        assert torch.equal(a2.state_dict()[k].float(), ad.state_dict()[k].to(data_type).float())
    r2 = P.ControlNetRouter.from_pretrained(str(tmp_path), subfolder="router_70000", low_cpu_mem_usage=False, device_map=None)
    assert r2.router_type == "simple_weights" and r2.num_experts == 3
    assert all(torch.equal(a, b) for a, b in zip(r2.state_dict().values(), r.state_dict().values()))
