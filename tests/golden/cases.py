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


def pyramid_inputs(N, h0, seed, with_mid):
    downs = [seeded_tensor((N, c, max(h0 // f, 1), max(h0 // f, 1)), seed + i) for i, (c, f) in enumerate(SD15_PYRAMID)]
    mid = seeded_tensor((N, 1280, max(h0 // 8, 1), max(h0 // 8, 1)), seed + 50) if with_mid else None
    return downs, mid
