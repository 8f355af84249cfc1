"""Definition of the golden cases (inputs are regenerated from seeds; only outputs are stored)."""
import torch

from oracle.init import seeded_tensor

SD15_PYRAMID = [(320, 1), (320, 1), (320, 1), (320, 2), (640, 2), (640, 2), (640, 4), (1280, 4), (1280, 4),
                (1280, 8), (1280, 8), (1280, 8)]   # (channels, down-sampling factor) of the 12 ControlNet outputs

CONTROLNET_CFG = dict()   # SD-1.5 defaults (controlnet/controlnet.py:181-217) with cross_attention_dim 768
CONTROLNET_KW = dict(cross_attention_dim=768)

ADAPTER_SDXL = dict(backbone_model_name="sdxl", num_blocks=1, num_frames=1, num_adapters_per_location=3,
                    cross_attention_dim=2048, add_spatial_resnet=True, add_temporal_resnet=False,
                    add_spatial_transformer=True, add_temporal_transformer=False,
                    add_adapter_location_A=True, add_adapter_location_B=True, add_adapter_location_C=True,
                    add_adapter_location_D=False, add_adapter_location_M=False)   # configs/sdxl_train_depth.yaml:36-56
ADAPTER_VIDEO = dict(backbone_model_name="i2vgen-xl", num_blocks=1, num_frames=4, num_adapters_per_location=3,
                     cross_attention_dim=1024, add_spatial_resnet=True, add_temporal_resnet=True,
                     add_spatial_transformer=True, add_temporal_transformer=True,
                     add_adapter_location_A=True, add_adapter_location_B=True, add_adapter_location_C=True,
                     add_adapter_location_D=True, add_adapter_location_M=True)    # configs/svd_train_depth.yaml:36-56


def controlnet_inputs(N=2, hs=8, seed=100):
    return dict(
        sample=seeded_tensor((N, 4, hs, hs), seed + 1),
        timestep=torch.tensor([999.0, 249.0][:N]) if N <= 2 else torch.full((N,), 499.0),
        encoder_hidden_states=seeded_tensor((N, 77, 768), seed + 2),
        controlnet_cond=seeded_tensor((N, 3, hs * 8, hs * 8), seed + 3, kind="uniform"),
    )


def controlnet_inputs_nonsquare(seed=150):
    """one image, 8 x 16 latents (64 x 128 condition image), 0-d timestep: the reference accepts any multiple of 8"""
    return dict(
        sample=seeded_tensor((1, 4, 8, 16), seed + 1),
        timestep=torch.tensor(499.0),
        encoder_hidden_states=seeded_tensor((1, 77, 768), seed + 2),
        controlnet_cond=seeded_tensor((1, 3, 64, 128), seed + 3, kind="uniform"),
    )


def pyramid_inputs(N, h0, seed, with_mid):
    downs = [seeded_tensor((N, c, max(h0 // f, 1), max(h0 // f, 1)), seed + i) for i, (c, f) in enumerate(SD15_PYRAMID)]
    mid = seeded_tensor((N, 1280, max(h0 // 8, 1), max(h0 // 8, 1)), seed + 50) if with_mid else None
    return downs, mid


# ---- configuration variants the reference supports beyond the shipped YAMLs (model/ctrl_adapter.py:17-116) ----
def _variant(base, **kw):
    d = dict(base)
    d.update(kw)
    return d


ADAPTER_VARIANTS = {
    # two adapters per location (selection_map[2] = first and last slot of a location), ResNet-only blocks: the
    # result leaves through the layout kernel instead of a GEMM epilogue
    "sdxl_resnet_only_AC_x2": (_variant(ADAPTER_SDXL, num_adapters_per_location=2, add_adapter_location_B=False,
                                        add_spatial_transformer=False), dict(N=2, frames=1, mid=False, ehs=(2, 77, 2048))),
    # two stacked (ResNet, transformer) layers per block: the inter-layer path (fp16 mirror feeding the next shortcut conv)
    "sdxl_two_blocks_C_x1": (_variant(ADAPTER_SDXL, num_blocks=2, num_adapters_per_location=1, add_adapter_location_A=False,
                                      add_adapter_location_B=False), dict(N=2, frames=1, mid=False, ehs=(2, 77, 2048))),
    # temporal modules only (no spatial ResNet / transformer), locations B and M, 2 clips x 3 frames
    "video_temporal_only_BM": (_variant(ADAPTER_VIDEO, add_spatial_resnet=False, add_spatial_transformer=False,
                                        add_adapter_location_A=False, add_adapter_location_C=False,
                                        add_adapter_location_D=False, num_frames=3), dict(N=6, frames=3, mid=True, ehs=(1, 1, 1024))),
    # transformer-only SDXL blocks: no ResNet to fold the x2 up-sampling into -> F.interpolate path (:235-237)
    "sdxl_transformer_only_BC_x1": (_variant(ADAPTER_SDXL, num_adapters_per_location=1, add_adapter_location_A=False,
                                             add_spatial_resnet=False), dict(N=2, frames=1, mid=False, ehs=(2, 77, 2048))),
    # 2-D encoder_hidden_states (adapter_spatial_temporal.py:240-241 -> one key per image: the query-independent
    # cross-attention with a PER-IMAGE vector) and per-sample timesteps
    "sdxl_ehs2d_per_sample_t": (dict(ADAPTER_SDXL), dict(N=2, frames=1, mid=False, ehs=(2, 2048), t=[749.0, 249.0])),
}


# ---- per-frame encoder states with MORE than one clip: the reference hands the temporal transformer a time context ordered
# (pixel, clip) while the block's rows are (clip, pixel) (model/adapter_spatial_temporal.py:246-249) -- goldens made by the
# reference's own file pin the oracle's restatement of that pairing (CPU); the HIP path is held against the oracle on the
# same inputs by tests/test_gpu_e2e.py::test_per_sample_context_with_several_clips ----
PER_CLIP_CONTEXT = {"video_per_clip_context_2x4": (2, 4), "video_per_clip_context_3x2": (3, 2)}


def per_clip_context_inputs(tag):
    clips, frames = PER_CLIP_CONTEXT[tag]
    N = clips * frames
    cfg = dict(ADAPTER_VIDEO)
    cfg.update(add_adapter_location_B=False, add_adapter_location_C=False, add_adapter_location_D=False, add_adapter_location_M=False)
    downs, _ = pyramid_inputs(N=N, h0=8, seed=810, with_mid=False)
    ehs = seeded_tensor((N, 1, 1024), 811)
    ehs = ehs * (1.0 + torch.arange(N).div(frames, rounding_mode="floor").view(N, 1, 1))      # clip b scaled by (1 + b)
    return cfg, clips, frames, downs, ehs, torch.full((N,), 500.0)


def variant_inputs(tag, seed=500):
    cfg, io = ADAPTER_VARIANTS[tag]
    downs, mid = pyramid_inputs(N=io["N"], h0=8, seed=seed, with_mid=io["mid"])
    ehs = seeded_tensor(io["ehs"], seed + 90)
    return cfg, io, downs, mid, ehs


def variant_timestep(io):
    return torch.tensor(io["t"]) if "t" in io else torch.tensor(333.0)
