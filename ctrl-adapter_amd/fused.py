"""One denoise step of the hot path as one call: ControlNet forward + Ctrl-Adapter forward, overlapped.

The pipelines call the two modules back to back (sdxl/pipelines/sdxl_controlnet_adapter_pipeline.py:1323 and :1338,
svd/pipelines/svd_controlnet_adapter_pipeline.py:684,709, i2vgen_xl/pipelines/...:957,1042) and only ever hand the
ControlNet's outputs to the adapter.  `controlled_step` takes the arguments of both calls, returns the results of both,
bit-identical to the separate calls, and lets every adapter block start as soon as ITS ControlNet output exists
(libctrlhip `ctrl_step_forward`: the ControlNet runs on a plan-owned HIP stream, per-output events, hipGraph-capturable).
"""
import ctypes as C

import torch

from . import _lib as L


@torch.no_grad()
def controlled_step(controlnet, adapter, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale=1.0,
                    guess_mode=False, skip_conv_in=False, skip_time_emb=False, *, adapter_encoder_hidden_states,
                    adapter_timestep=None, num_frames=None, use_mid=None, scatter_to=None, out_dtype=None, clip_batch=None,
                    discard_when_off=False):
    """-> ((down_block_res_samples, mid_block_res_sample), (adapted_down_block_res_samples, adapted_mid | None))

    Positional part = ControlNetModel.forward's arguments; keyword part = ControlNetAdapter.forward's
    (`adapter_timestep` defaults to `timestep`; `use_mid` defaults to the adapter having a mid block, as in the video
    pipelines that pass `mid_block_res_sample`).

    discard_when_off=True: on a step whose control is switched off (conditioning_scale == 0, i.e. controlnet_keep == 0
    outside [control_guidance_start, control_guidance_end]) the SDXL pipeline throws the adapter's down residuals away
    (sdxl/pipelines/sdxl_controlnet_adapter_pipeline.py:1348 `if cond_scale == 0: ...down_block_res_samples = None`) after
    having computed them.  With this flag the adapter is not run at all on such a step when it has no mid block to
    deliver (what the pipeline keeps in that case is None as well) and (None, None) is returned for it: 40 % of the steps
    at control_guidance_end = 0.6 (note N8).  Off by default: the plain call returns what the two
    modules return."""
    cn_dtype = controlnet._check_inputs(sample, encoder_hidden_states, controlnet_cond)
    if controlnet.config.get("global_pool_conditions", False):
        # pooled outputs are produced after the network (a pass over its outputs): nothing for the adapter to start early on
        down, mid = controlnet(sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale, guess_mode=guess_mode,
                               return_dict=False, skip_conv_in=skip_conv_in, skip_time_emb=skip_time_emb)
        use_m = adapter.add_adapter_location_M if use_mid is None else use_mid
        return (down, mid), adapter(down, mid if use_m else None, num_frames=num_frames,
                                    timestep=timestep if adapter_timestep is None else adapter_timestep,
                                    encoder_hidden_states=adapter_encoder_hidden_states, scatter_to=scatter_to,
                                    out_dtype=out_dtype, clip_batch=clip_batch)
    if isinstance(conditioning_scale, (int, float)) and conditioning_scale == 0:
        # control off for this step: the separate calls already skip the ControlNet (note N8); nothing to overlap
        down, mid = controlnet(sample, timestep, encoder_hidden_states, controlnet_cond, 0, guess_mode=guess_mode,
                               return_dict=False, skip_conv_in=skip_conv_in, skip_time_emb=skip_time_emb)
        use_m = adapter.add_adapter_location_M if use_mid is None else use_mid
        if discard_when_off and not use_m:
            return (down, mid), (None, None)
        return (down, mid), adapter(down, mid if use_m else None, num_frames=num_frames,
                                    timestep=timestep if adapter_timestep is None else adapter_timestep,
                                    encoder_hidden_states=adapter_encoder_hidden_states, scatter_to=scatter_to,
                                    out_dtype=out_dtype, clip_batch=clip_batch)
    cn_outs, cn_args, _k1 = controlnet._launch_args(sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale,
                                                    guess_mode, skip_conv_in, skip_time_emb, cn_dtype)
    down, mid = cn_outs[:12], cn_outs[12]
    use_m = bool(adapter.add_adapter_location_M if use_mid is None else use_mid)
    outs, mid_out, ad_args, tail, finish, _k2 = adapter._launch_args(
        down, mid if use_m else None, num_frames, timestep if adapter_timestep is None else adapter_timestep,
        adapter_encoder_hidden_states, scatter_to, out_dtype, clip_batch)
    N, H0, W0 = ad_args[0], ad_args[1], ad_args[2]
    frame_pos, n_out = (tail[0], tail[1]) if tail is not None else (None, N)
    import torch
    with torch.cuda.device(sample.device):
        # the plans' text K/V cache modes are sticky: set them for THIS call from the tensors of THIS call (a separate
        # controlnet(...) / adapter(...) call of an earlier request may have left REUSE behind -- ADVICE r2)
        controlnet._text_cache_mode(encoder_hidden_states, L.lib().ctrl_controlnet_text_cache)
        adapter._text_cache_mode(adapter_encoder_hidden_states, L.lib().ctrl_adapter_text_cache)
        L.check(L.lib().ctrl_step_forward(
            controlnet._ensure_plan(), adapter._ensure_plan(), *cn_args,
            ad_args[3], *ad_args[4:10], int(use_m and mid_out is not None), ad_args[10], ad_args[11], frame_pos, n_out, L.cur_stream()))
        L.raise_if_out_of_range("controlled_step")
    return (down, mid), finish(outs, mid_out)
