"""ControlNetAdapter -- drop-in mirror of the reference's model/ctrl_adapter.py:ControlNetAdapter whose forward runs
entirely in libctrlhip.  Same constructor arguments (:17-44), forward signature and return (:171-172, :224), same
state-dict keys (down_blocks_adapter.{i}.*, mid_block_adapter.*)."""
import ctypes as C

import torch

from . import _lib as L
from ._plan import ParamTreeModule, Config, c_spec, timesteps_to_device_f32

_SLOT_C = [320, 320, 320, 320, 640, 640, 640, 1280, 1280, 1280, 1280, 1280]
_SLOT_F = [1, 1, 1, 2, 2, 2, 4, 4, 4, 8, 8, 8]


class ControlNetAdapter(ParamTreeModule):
    def __init__(self, backbone_model_name, num_blocks=2, num_frames=8, num_adapters_per_location=3,
                 cross_attention_dim=None, adapter_type="spatial_temporal_resnet_transformer",
                 add_spatial_resnet=True, add_temporal_resnet=False, add_spatial_transformer=True,
                 add_temporal_transformer=False, add_adapter_location_A=False, add_adapter_location_B=False,
                 add_adapter_location_C=False, add_adapter_location_D=False, add_adapter_location_M=False,
                 num_repeats=1, out_channels=None):
        super().__init__()
        self.config = Config({k: v for k, v in locals().items() if k not in ("self", "__class__")})
        if num_repeats != 1:
            raise ValueError("num_repeats > 1 (experimental zero-conv aggregation, ctrl_adapter.py:208-221) is not supported")
        if cross_attention_dim is None and (add_spatial_transformer or add_temporal_transformer):
            raise ValueError("cross_attention_dim is required when a transformer sub-module is enabled")
        cfg = L.AdapterConfig()
        cfg.backbone_sdxl = int(backbone_model_name in ["sdxl"])
        cfg.num_blocks = num_blocks
        cfg.num_adapters_per_location = num_adapters_per_location
        cfg.cross_attention_dim = cross_attention_dim or 0
        cfg.add_spatial_resnet, cfg.add_temporal_resnet = int(add_spatial_resnet), int(add_temporal_resnet)
        cfg.add_spatial_transformer, cfg.add_temporal_transformer = int(add_spatial_transformer), int(add_temporal_transformer)
        cfg.loc_A, cfg.loc_B, cfg.loc_C, cfg.loc_D, cfg.loc_M = (int(add_adapter_location_A), int(add_adapter_location_B),
                                                                 int(add_adapter_location_C), int(add_adapter_location_D),
                                                                 int(add_adapter_location_M))
        self._cfg = cfg
        self._up = 2 if cfg.backbone_sdxl else 1
        lib = L.lib()
        self._register_spec(c_spec(lib.ctrl_adapter_param_count, lib.ctrl_adapter_param_spec, cfg))
        self.num_adapters_per_location = num_adapters_per_location
        self.add_adapter_location_M = add_adapter_location_M

    def config_dict(self):
        return dict(self.config)

    def get_down_block_ids(self):
        sel = {"A": {3: [0, 1, 2], 2: [0, 2], 1: [2]}, "B": {3: [3, 4, 5], 2: [3, 5], 1: [5]},
               "C": {3: [6, 7, 8], 2: [6, 8], 1: [8]}, "D": {3: [9, 10, 11], 2: [9, 11], 1: [11]}}
        ids = []
        for k in "ABCD":
            if self.config["add_adapter_location_" + k]:
                ids += sel[k].get(self.num_adapters_per_location, [])
        return ids

    def _destroy(self, plan):
        L.lib().ctrl_adapter_destroy(plan)

    def trim(self):
        """frees the workspace / cache blocks the plan outgrew (see ControlNetModel.trim)"""
        if getattr(self, "_plan", None) is not None:
            L.check(L.lib().ctrl_adapter_trim(self._plan))

    def _ensure_plan(self):
        if self._plan is None:
            refs, n, keep = self._tensor_refs()
            h = C.c_void_p()
            L.check(L.lib().ctrl_adapter_create(C.byref(self._cfg), refs, n, L.cur_stream(), C.byref(h)))
            self._plan = h
            self._note_selection(L.lib().ctrl_adapter_selection)
        return self._plan

    @torch.no_grad()
    def forward(self, down_block_res_samples, mid_block_res_sample=None, sparsity_masking=None, num_frames=None,
                timestep=None, encoder_hidden_states=None, *, scatter_to=None, out_dtype=None, clip_batch=None, clip_comm=None):
        """Reference signature (model/ctrl_adapter.py:171) plus three keyword-only extensions that fold the pipelines'
        residual hand-over into the last epilogue of every adapter block:

        scatter_to=(positions, total_frames): input frame j is written at frame positions[j] of total_frames-frame
            outputs, all other frames are zero: the sparse -> dense loop of i2vgen_xl/pipelines/...:1052-1071 and
            svd/pipelines/...:719-741 without the zeros tensor and the per-frame python copies.
        out_dtype: dtype of the returned tensors (those loops build float32 tensors); default = the input dtype.
        clip_batch=bs: return `[bs, c, nf, h, w]` tensors, i.e. rearrange(x, "(bs nf) c h w -> bs c nf h w"), as
            zero-copy views; the UNets' inverse rearrange (i2vgen_xl/models/unets/unet_i2vgen_xl.py:683-684) is then a
            view of the same memory as well.
        clip_comm=transport (clip_parallel.ClipTransport): ONE clip's frames are sharded over the transport's ranks; the
            tensors of this call hold the rank's `num_frames` LOCAL frames of every clip (frames [rank*num_frames,
            (rank+1)*num_frames)), and the three frame-mixing ops exchange through the transport."""
        # sparsity_masking is accepted and ignored, exactly like the reference (SURVEY.md note N7)
        outs, mid_out, args, tail, finish, _keep = self._launch_args(
            down_block_res_samples, mid_block_res_sample, num_frames, timestep, encoder_hidden_states, scatter_to, out_dtype, clip_batch)
        in_ptrs, in_dt = _keep[0], _keep[1]
        with torch.cuda.device(down_block_res_samples[0].device):
            self._text_cache_mode(encoder_hidden_states, L.lib().ctrl_adapter_text_cache)
            if clip_comm is not None:
                pos, n_out = (tail[0], tail[1]) if tail is not None else (None, args[0])
                for attempt in range(2):
                    cs = clip_comm.c_struct()
                    clip_comm.error = None
                    rc = L.lib().ctrl_adapter_forward_clip_sharded(self._ensure_plan(), in_ptrs, in_dt, *args, pos, n_out,
                                                                   C.byref(cs), L.cur_stream())
                    if rc == 2 and attempt == 0:         # exchange workspace too small: grow it and retry
                        clip_comm.ensure(cs.ws_needed)
                        continue
                    if clip_comm.error is not None:
                        raise clip_comm.error
                    L.check(rc)
                    break
            elif tail is None:
                L.check(L.lib().ctrl_adapter_forward(self._ensure_plan(), in_ptrs, in_dt, *args, L.cur_stream()))
            else:
                L.check(L.lib().ctrl_adapter_forward_scatter(self._ensure_plan(), in_ptrs, in_dt, *args, *tail, L.cur_stream()))
            L.raise_if_out_of_range("ControlNetAdapter.forward")
        return finish(outs, mid_out)

    def _launch_args(self, down_block_res_samples, mid_block_res_sample, num_frames, timestep, encoder_hidden_states,
                     scatter_to, out_dtype, clip_batch):
        """Validation + output tensors + the C argument list shared by ctrl_adapter_forward[_scatter] and
        ctrl_step_forward: args = (N, H0, W0, num_frames, t, t_count, ehs, ehs_dtype, ehs_batch, Lk, out_ptrs, out_dtype),
        tail = (frame_pos, N_out) or None, finish = the clip_batch view step."""
        if len(down_block_res_samples) != 12:
            raise ValueError("expected the 12 ControlNet down_block_res_samples")
        x0 = down_block_res_samples[0]
        if not x0.is_cuda:
            raise RuntimeError("ControlNetAdapter (libctrlhip) runs on the GPU only; there is no CPU fallback")
        N, _, H0, W0 = x0.shape
        dt = x0.dtype
        for i, (t, c, f) in enumerate(zip(down_block_res_samples, _SLOT_C, _SLOT_F)):
            if tuple(t.shape) != (N, c, max(H0 // f, 1), max(W0 // f, 1)) or t.dtype != dt:
                raise ValueError("down_block_res_samples[%d] has shape %s, expected the SD-1.5 ControlNet pyramid" % (i, tuple(t.shape)))
        num_frames = int(num_frames) if num_frames is not None else 1
        t32 = timesteps_to_device_f32(timestep, N, x0.device)
        needs_ehs = self.config.add_spatial_transformer or self.config.add_temporal_transformer
        ehs = None
        eb, Lk = 1, 1
        if needs_ehs:
            ehs = encoder_hidden_states
            if ehs.dim() == 2:                      # adapter_spatial_temporal.py:240-241
                ehs = ehs.unsqueeze(1)
            ehs = ehs.contiguous()
            eb, Lk = ehs.shape[0], ehs.shape[1]
            if ehs.shape[2] != self.config.cross_attention_dim or eb not in (1, N):
                raise ValueError("encoder_hidden_states must be [1 or N, L, %d]" % self.config.cross_attention_dim)
        ids = self.get_down_block_ids()
        ins = [t.contiguous() for t in down_block_res_samples]
        odt = out_dtype if out_dtype is not None else dt
        N_out, pos = N, None
        if scatter_to is not None:
            positions, N_out = scatter_to
            positions = [int(p) for p in positions]
            N_out = int(N_out)
            if len(positions) != N or len(set(positions)) != N or min(positions) < 0 or max(positions) >= N_out:
                raise ValueError("scatter_to: need %d distinct frame positions in [0, %d)" % (N, N_out))
            pos = (C.c_int32 * N)(*positions)
        if clip_batch is not None and (int(clip_batch) < 1 or N_out % int(clip_batch)):
            raise ValueError("clip_batch must divide the number of output frames")
        outs = []
        for i, (c, f) in enumerate(zip(_SLOT_C, _SLOT_F)):
            h, w = max(H0 // f, 1), max(W0 // f, 1)
            s = self._up if i in ids else 1             # zeros_like keeps the input size (ctrl_adapter.py:193)
            outs.append(torch.empty(N_out, c, h * s, w * s, dtype=odt, device=x0.device))
        mid_in = mid_out = None
        if mid_block_res_sample is not None and self.add_adapter_location_M:
            mid_in = mid_block_res_sample.contiguous()
            mid_out = torch.empty(N_out, 1280, mid_in.shape[2] * self._up, mid_in.shape[3] * self._up, dtype=odt, device=x0.device)
        in_ptrs = (C.c_void_p * 13)(*([t.data_ptr() for t in ins] + [mid_in.data_ptr() if mid_in is not None else None]))
        out_ptrs = (C.c_void_p * 13)(*([t.data_ptr() for t in outs] + [mid_out.data_ptr() if mid_out is not None else None]))
        args = [N, H0, W0, num_frames, L.ptr(t32), t32.numel(),
                L.ptr(ehs), L.dtype_code(ehs.dtype) if ehs is not None else 0, eb, Lk, out_ptrs, L.dtype_code(odt)]
        tail = None if pos is None else [pos, N_out]

        def finish(outs, mid_out):
            if clip_batch is not None:
                bs = int(clip_batch)
                as_clips = lambda x: x.view(bs, N_out // bs, *x.shape[1:]).permute(0, 2, 1, 3, 4)
                outs = [as_clips(x) for x in outs]
                mid_out = as_clips(mid_out) if mid_out is not None else None
            return outs, mid_out
        return outs, mid_out, args, tail, finish, (in_ptrs, L.dtype_code(dt), ins, mid_in, t32, ehs, out_ptrs, pos)
