"""ControlNetModel -- drop-in mirror of the reference's controlnet/controlnet.py:ControlNetModel whose forward runs
entirely in libctrlhip (hand-written gfx950 kernels).  Same forward signature (:662-678), same return
(ControlNetOutput or the (down_block_res_samples, mid_block_res_sample) tuple, :876-881), same state-dict keys."""
import ctypes as C

import torch

from . import _lib as L
from ._plan import ParamTreeModule, Config, c_spec, timesteps_to_device_f32


class ControlNetOutput(tuple):
    """(down_block_res_samples, mid_block_res_sample) with attribute access (diffusers ControlNetOutput)."""

    def __new__(cls, down_block_res_samples, mid_block_res_sample):
        return super().__new__(cls, (down_block_res_samples, mid_block_res_sample))

    down_block_res_samples = property(lambda self: self[0])
    mid_block_res_sample = property(lambda self: self[1])


class ControlNetModel(ParamTreeModule):
    def __init__(self, in_channels=4, conditioning_channels=3, flip_sin_to_cos=True, freq_shift=0,
                 down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
                 mid_block_type="UNetMidBlock2DCrossAttn", only_cross_attention=False,
                 block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, downsample_padding=1,
                 mid_block_scale_factor=1, act_fn="silu", norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=1280,
                 transformer_layers_per_block=1, encoder_hid_dim=None, encoder_hid_dim_type=None, attention_head_dim=8,
                 num_attention_heads=None, use_linear_projection=False, class_embed_type=None, addition_embed_type=None,
                 addition_time_embed_dim=None, num_class_embeds=None, upcast_attention=False,
                 resnet_time_scale_shift="default", projection_class_embeddings_input_dim=None,
                 controlnet_conditioning_channel_order="rgb", conditioning_embedding_out_channels=(16, 32, 96, 256),
                 global_pool_conditions=False, addition_embed_type_num_heads=64):
        super().__init__()
        self.config = Config({k: v for k, v in locals().items() if k not in ("self", "__class__")})
        # opt-in step-invariant cache of the conditioning embedder: with the same `controlnet_cond`
        # tensor object, unmodified, on consecutive forwards only the embedder's last conv runs.  Bit-identical results.
        self.cache_condition = False
        self._cond_ref, self._cond_version, self._cond_key = None, -1, None
        # the hot path only reaches this configuration family (SD-1.5 ControlNets); anything else fails loudly
        unsupported = []
        if not flip_sin_to_cos or freq_shift != 0: unsupported.append("time_proj variant")
        if len(block_out_channels) != 4 or len(down_block_types) != 4: unsupported.append("block count != 4")
        if mid_block_type != "UNetMidBlock2DCrossAttn": unsupported.append("mid_block_type")
        if only_cross_attention or use_linear_projection or upcast_attention: unsupported.append("attention variant")
        if act_fn not in ("silu", "swish") or norm_num_groups != 32 or resnet_time_scale_shift != "default":
            unsupported.append("resnet variant")
        if transformer_layers_per_block != 1: unsupported.append("transformer_layers_per_block")
        if any(v is not None for v in (encoder_hid_dim, encoder_hid_dim_type, class_embed_type, addition_embed_type,
                                        num_class_embeds)):
            unsupported.append("class/addition embeddings")
        if controlnet_conditioning_channel_order != "rgb": unsupported.append("channel order")
        if downsample_padding != 1 or mid_block_scale_factor != 1: unsupported.append("padding/scale")
        if unsupported:
            raise ValueError("ControlNetModel (libctrlhip): unsupported configuration: " + ", ".join(unsupported))
        heads = num_attention_heads or attention_head_dim      # the reference's naming quirk (:221-227)
        if not isinstance(heads, int):
            raise ValueError("per-block head counts are not supported")
        cfg = L.ControlNetConfig()
        cfg.in_channels = in_channels
        cfg.conditioning_channels = conditioning_channels
        for i in range(4):
            cfg.block_out_channels[i] = block_out_channels[i]
            cfg.down_block_has_attn[i] = int(down_block_types[i] == "CrossAttnDownBlock2D")
            cfg.cond_embed_channels[i] = conditioning_embedding_out_channels[i]
        cfg.layers_per_block = layers_per_block
        cfg.num_attention_heads = heads
        cfg.cross_attention_dim = cross_attention_dim
        cfg.norm_eps = norm_eps
        self._cfg = cfg
        lib = L.lib()
        self._register_spec(c_spec(lib.ctrl_controlnet_param_count, lib.ctrl_controlnet_param_spec, cfg))
        co = list(block_out_channels)
        self._slot_channels = [co[0]] + sum(([co[i]] * (layers_per_block + (1 if i != 3 else 0)) for i in range(4)), [])
        self._slot_factor = [1] + sum(([2 ** i] * layers_per_block + ([2 ** (i + 1)] if i != 3 else []) for i in range(4)), [])

    def config_dict(self):
        return {k: (list(v) if isinstance(v, tuple) else v) for k, v in self.config.items()}

    def _destroy(self, plan):
        L.lib().ctrl_controlnet_destroy(plan)

    def trim(self):
        """frees the workspace / cache blocks the plan outgrew (they are kept alive for queued launches and captured graphs);
        synchronises the device -- call between requests, when no graph captured before the last growth will be replayed"""
        if getattr(self, "_plan", None) is not None:
            L.check(L.lib().ctrl_controlnet_trim(self._plan))

    def _ensure_plan(self):
        if self._plan is None:
            refs, n, keep = self._tensor_refs()
            h = C.c_void_p()
            L.check(L.lib().ctrl_controlnet_create(C.byref(self._cfg), refs, n, L.cur_stream(), C.byref(h)))
            self._plan = h
            self._note_selection(L.lib().ctrl_controlnet_selection)
        return self._plan

    def _drop_plan(self):
        self._cond_ref, self._cond_version, self._cond_key = None, -1, None      # the cache lives in the plan
        if getattr(self, "_plan_b", None) is not None:
            self._destroy(self._plan_b)
        self._plan_b = None
        super()._drop_plan()

    # -- batch lanes (round 6): the two halves of a batch on two stream lanes --
    # A ControlNet forward at the pipelines' batch sizes is a dependent chain of ~220 launches most of which fill a fraction of the chip
    # (M = 512 .. 32768 rows at b = 8: 0.10 of the MFMA peak, 7 of its 8 ms in launches below a quarter of either roof).  Images are
    # independent, so images [0, N/2) run on the caller's stream and [N/2, N) on a lane of the module's own through a CLONE of the plan
    # (same packed weights, its own workspace: ctrl_controlnet_clone), forked / joined with events -- the two chains interleave on the
    # chip.  Every image is computed exactly as in a forward of N/2 images (tile choices follow the batch the dispatcher sees, so the
    # last bits can differ from the one-batch forward, like any other batch size).  MEASURED AND NOT ADOPTED: SDXL b = 8 29.05 / 29.10 ms
    # without against 29.31 / 29.32 ms with it (same box, two rounds each, profiles/r06_cn_batch_lanes.txt) -- the half-batch kernels are not
    # half as long and the two chains contend; what does pay is lanes over INDEPENDENT nets (MultiControlNetModel: -7 %).  Opt-in:
    # CTRL_CN_BATCH_LANES=1; never while the per-launch profiler records, with the step-invariant caches or below 8 images.
    BATCH_LANES_MIN = 8

    def _batch_lanes_apply(self, N, timestep):
        from . import ops
        if N < self.BATCH_LANES_MIN or N % 2 or self.cache_condition or self.cache_text or ops.profiling() or getattr(self, "_no_aux_lane", False):
            return False          # (_no_aux_lane: this forward already runs on a lane of MultiControlNetModel -- no fork inside a fork)
        return ops.policy().get("CTRL_CN_BATCH_LANES", "0") == "1"

    def _forward_batch_lanes(self, args, outs, N, device):
        lib = L.lib()
        plan_a = self._ensure_plan()
        if getattr(self, "_plan_b", None) is None:
            h = C.c_void_p()
            L.check(lib.ctrl_controlnet_clone(plan_a, C.byref(h)))
            self._plan_b = h
            self._lane = torch.cuda.Stream(device=device)
        # args (see _launch_args): sample, dt, N, Hs, Ws, t, t_count, ehs, dt, Lk, cond, dt, scale, flags, out pointers, out dtype
        sample_p, sdt, _, Hs, Ws, t_p, t_n, ehs_p, edt, Lk, cond_p, cdt, scale, flags, ptrs, odt = args
        h = N // 2
        esz = {L.F32: 4, L.F16: 2, L.BF16: 2}

        def off(p, nbytes):
            return C.c_void_p((p.value or 0) + nbytes)

        cross = self.config.cross_attention_dim
        cc = int(self.config.get("conditioning_channels", 3))
        ptrs_b = (C.c_void_p * 13)()
        for i, o in enumerate(outs):
            ptrs_b[i] = o.data_ptr() + h * o.stride(0) * o.element_size()
        args_a = [sample_p, sdt, h, Hs, Ws, t_p, min(t_n, h) if t_n > 1 else 1, ehs_p, edt, Lk, cond_p, cdt, scale, flags | 32, ptrs, odt]
        cin = int(self.config.get("in_channels", 4))
        args_b = [off(sample_p, h * cin * Hs * Ws * esz[sdt]), sdt, h, Hs, Ws, off(t_p, h * 4) if t_n > 1 else t_p, h if t_n > 1 else 1,
                  off(ehs_p, h * Lk * cross * esz[edt]), edt, Lk, off(cond_p, h * cc * 64 * Hs * Ws * esz[cdt]), cdt, scale, flags | 32, ptrs_b, odt]
        cur = torch.cuda.current_stream(device)
        self._lane.wait_stream(cur)
        with torch.cuda.stream(self._lane):
            L.check(lib.ctrl_controlnet_forward(self._plan_b, *args_b, L.cur_stream()))
        L.check(lib.ctrl_controlnet_forward(plan_a, *args_a, L.cur_stream()))
        cur.wait_stream(self._lane)

    def _launch_args(self, sample, timestep, ehs, controlnet_cond, conditioning_scale, guess_mode, skip_conv_in,
                     skip_time_emb, out_dtype):
        """Output tensors + the C argument list shared by ctrl_controlnet_forward and ctrl_step_forward
        (everything between the plan handle and the stream); the third value keeps the marshalled inputs alive."""
        N, _, Hs, Ws = sample.shape
        t = timesteps_to_device_f32(timestep, N, sample.device)
        outs = [torch.empty(N, c, max(Hs // f, 1), max(Ws // f, 1), dtype=out_dtype, device=sample.device)
                for c, f in zip(self._slot_channels, self._slot_factor)]
        outs.append(torch.empty(N, self._slot_channels[-1], max(Hs // 8, 1), max(Ws // 8, 1), dtype=out_dtype, device=sample.device))
        sample_c, ehs_c, cond_c = sample.contiguous(), ehs.contiguous(), controlnet_cond.contiguous()
        ptrs = (C.c_void_p * 13)(*[o.data_ptr() for o in outs])
        flags = (1 if skip_conv_in else 0) | (2 if skip_time_emb else 0) | (4 if guess_mode else 0) | (32 if getattr(self, "_no_aux_lane", False) else 0)
        if self.cache_condition:
            # same tensor object, not written to since the forward that cached it (and same plan): the embedder's hidden
            # map is still valid.  The reference recomputes it on each of the ~50 steps.
            key = (tuple(controlnet_cond.shape), controlnet_cond.dtype, tuple(sample.shape))
            same = (self._cond_ref is controlnet_cond and self._cond_version == controlnet_cond._version and self._cond_key == key)
            flags |= 16 if same else 8
            self._cond_ref, self._cond_version, self._cond_key = controlnet_cond, controlnet_cond._version, key
        args = [L.ptr(sample_c), L.dtype_code(sample_c.dtype), N, Hs, Ws, L.ptr(t), t.numel(),
                L.ptr(ehs_c), L.dtype_code(ehs_c.dtype), ehs_c.shape[1], L.ptr(cond_c), L.dtype_code(cond_c.dtype),
                C.c_float(float(conditioning_scale)), flags, ptrs, L.dtype_code(out_dtype)]
        return outs, args, (t, sample_c, ehs_c, cond_c, ptrs)

    def _check_inputs(self, sample, encoder_hidden_states, controlnet_cond):
        if not sample.is_cuda:
            raise RuntimeError("ControlNetModel (libctrlhip) runs on the GPU only; there is no CPU fallback")
        N, _, Hs, Ws = sample.shape
        if controlnet_cond.shape[0] != N or controlnet_cond.shape[2] != 8 * Hs or controlnet_cond.shape[3] != 8 * Ws:
            raise ValueError("controlnet_cond must be [N, C, 8*H, 8*W] for a latent sample [N, 4, H, W]")
        ehs = encoder_hidden_states
        if ehs.dim() != 3 or ehs.shape[0] != N or ehs.shape[2] != self.config.cross_attention_dim:
            raise ValueError("encoder_hidden_states must be [N, L, %d]" % self.config.cross_attention_dim)
        return sample.dtype if sample.dtype in (torch.float16, torch.bfloat16, torch.float32) else self.dtype

    @torch.no_grad()
    def forward(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale=1.0,
                class_labels=None, timestep_cond=None, attention_mask=None, added_cond_kwargs=None,
                cross_attention_kwargs=None, guess_mode=False, return_dict=True, skip_conv_in=False, skip_time_emb=False):
        if class_labels is not None or timestep_cond is not None or attention_mask is not None:
            raise ValueError("class_labels / timestep_cond / attention_mask are not supported by the HIP hot path")
        out_dtype = self._check_inputs(sample, encoder_hidden_states, controlnet_cond)
        N, _, Hs, Ws = sample.shape
        ehs = encoder_hidden_states
        if isinstance(conditioning_scale, (int, float)) and conditioning_scale == 0:
            # control switched off for this step (controlnet_keep == 0, sdxl pipeline :1262-1266): every output of the
            # reference is `conv(x) * 0`; return the zeros without running the network (SURVEY.md note N8)
            # (global_pool_conditions: the reference's outputs are the [N, C, 1, 1] spatial means, controlnet.py:870-874)
            pooled = bool(self.config.get("global_pool_conditions", False))
            down = [torch.zeros(N, c, 1 if pooled else max(Hs // f, 1), 1 if pooled else max(Ws // f, 1), dtype=out_dtype, device=sample.device)
                    for c, f in zip(self._slot_channels, self._slot_factor)]
            mid = torch.zeros(N, self._slot_channels[-1], 1 if pooled else max(Hs // 8, 1), 1 if pooled else max(Ws // 8, 1),
                              dtype=out_dtype, device=sample.device)
            return ControlNetOutput(down, mid) if return_dict else (down, mid)
        pool = bool(self.config.get("global_pool_conditions", False))
        # guess-mode scaling is skipped for globally pooled conditions (controlnet/controlnet.py:861)
        outs, args, _keep = self._launch_args(sample, timestep, ehs, controlnet_cond, conditioning_scale, guess_mode and not pool,
                                              skip_conv_in, skip_time_emb, out_dtype)
        with torch.cuda.device(sample.device):      # plan, stream and launches follow the tensors' device, not the current one
            self._text_cache_mode(encoder_hidden_states, L.lib().ctrl_controlnet_text_cache)
            if self._batch_lanes_apply(N, timestep):
                self._forward_batch_lanes(args, outs, N, sample.device)
            else:
                L.check(L.lib().ctrl_controlnet_forward(self._ensure_plan(), *args, L.cur_stream()))
            L.raise_if_out_of_range("ControlNetModel.forward")
        if pool:       # controlnet/controlnet.py:870-874: torch.mean(sample, dim=(2, 3), keepdim=True) of every output
            from . import ops
            outs = [ops.avgpool_nchw(o, 1, 1) for o in outs]
        down, mid = outs[:12], outs[12]
        if not return_dict:
            return (down, mid)
        return ControlNetOutput(down, mid)


class MultiControlNetModel(torch.nn.Module):
    """controlnet/multicontrolnet.py:45-99: runs K nets on the same latents and returns per-net LISTS."""

    def __init__(self, controlnets):
        super().__init__()
        self.nets = torch.nn.ModuleList(controlnets)

    def forward(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale, class_labels=None,
                timestep_cond=None, attention_mask=None, added_cond_kwargs=None, cross_attention_kwargs=None,
                guess_mode=False, return_dict=True, skip_conv_in=False, skip_time_emb=False):
        jobs = list(zip(controlnet_cond, conditioning_scale, self.nets))                # positional zip (quirk N6)

        def run(image, scale, net, on_lane=False):
            # (a net on a lane of its own keeps to that one stream: its auxiliary lane would be a fork inside a fork, which crashes
            #  hipGraph's end-of-capture on ROCm 7.2 -- CTRL_NO_AUX_LANE; same kernels, same results)
            net._no_aux_lane = on_lane
            try:
                return net(sample, timestep, encoder_hidden_states, image, scale, guess_mode=guess_mode, return_dict=False,
                           skip_conv_in=skip_conv_in, skip_time_emb=skip_time_emb)
            finally:
                net._no_aux_lane = False

        from . import ops
        # (not while the per-launch profiler records: overlapping kernels would each be charged the others' time)
        lanes = len(jobs) > 1 and sample.is_cuda and (ops.policy().get("CTRL_MULTI_CN_LANES", "1") != "0") and not ops.profiling()
        if not lanes:
            outs = [run(*j) for j in jobs]
        else:
            # Round 6: the K nets are independent (same latents / prompt, their own condition image and weights), and a ControlNet forward
            # is a dependent chain of ~220 launches most of which fill a fraction of the chip -- so net k runs on its own stream lane,
            # forked from and joined to the caller's stream with events (hipGraph-capturable: bench.py captures it), and the chains overlap.
            # Same kernels, same order per net: results are bit-identical to the serial loop (CTRL_MULTI_CN_LANES=0).
            cur = torch.cuda.current_stream(sample.device)
            if getattr(self, "_lanes", None) is None or len(self._lanes) < len(jobs) - 1:
                self._lanes = [torch.cuda.Stream(device=sample.device) for _ in range(len(jobs) - 1)]
            outs = [None] * len(jobs)
            for k, j in enumerate(jobs[1:], start=1):
                s = self._lanes[k - 1]
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    outs[k] = run(*j, on_lane=True)
            outs[0] = run(*jobs[0], on_lane=True)         # net 0 on the caller's stream, beside the others
            for k in range(1, len(jobs)):
                cur.wait_stream(self._lanes[k - 1])
                for t_ in list(outs[k][0]) + [outs[k][1]]:
                    t_.record_stream(cur)                 # allocated on the lane, consumed on the caller's stream
        downs, mids = [o[0] for o in outs], [o[1] for o in outs]
        return downs, mids


def pool_latents(latents, size=(64, 64)):
    """F.adaptive_avg_pool2d(latents, (64, 64)) of the callers (sdxl pipeline :1306-1309) as a HIP kernel."""
    from . import ops
    if tuple(latents.shape[-2:]) == tuple(size):
        return latents
    return ops.avgpool_nchw(latents, size[0], size[1])
