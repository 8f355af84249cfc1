"""ControlNetRouter -- mirror of model/ctrl_router.py:ControlNetRouter (:49-112) plus the caller-side expert merge
(i2vgen_xl/pipelines/i2vgen_xl_controlnet_adapter_pipeline.py:1000-1022; train.py:1262-1276), on libctrlhip."""
import torch
from torch import nn

from . import ops
from ._plan import Config, PretrainedMixin


class ControlNetRouter(PretrainedMixin, nn.Module):
    def __init__(self, num_experts=2, backbone_model_name=None, router_type="simple_weights", embedding_dim=None,
                 num_routers=12, add_mid_block_router=True, use_sparsemax=False):
        super().__init__()
        if router_type not in ("simple_weights", "equal_weights"):
            raise ValueError("router_type %r is referenced by the reference's callers but not implemented in "
                             "model/ctrl_router.py; only simple_weights / equal_weights exist" % router_type)
        self.config = Config({k: v for k, v in locals().items() if k not in ("self", "__class__")})
        self.num_experts, self.num_routers, self.router_type = num_experts, num_routers, router_type
        self.add_mid_block_router = add_mid_block_router
        if router_type == "simple_weights":
            self.down_blocks_router = nn.ModuleList([self._wg(num_experts) for _ in range(num_routers)])
            self.mid_block_router = self._wg(num_experts) if add_mid_block_router else None

    def config_dict(self):
        return dict(self.config)

    @staticmethod
    def _wg(e):
        m = nn.Module()
        m.wg = nn.Linear(1, e, bias=False)
        return m

    @property
    def dtype(self):
        p = next(self.parameters(), None)
        return p.dtype if p is not None else torch.float32

    @torch.no_grad()
    def forward(self, router_input=None, sparse_mask=None, fixed_weights=None):
        R = self.num_routers + (1 if self.add_mid_block_router else 0)
        if self.router_type == "simple_weights":
            rows = [r.wg.weight[:, 0] for r in self.down_blocks_router]
            if self.add_mid_block_router:
                rows.append(self.mid_block_router.wg.weight[:, 0])
            wg = torch.stack(rows).float().contiguous()
        else:
            # no parameters to take the device from (equal weights): the caller's current device, like every op here
            wg = torch.zeros(R, self.num_experts, device=torch.device("cuda", torch.cuda.current_device()))
        if not wg.is_cuda:
            raise RuntimeError("ControlNetRouter (libctrlhip) runs on the GPU only")
        mask = [int(m) for m in sparse_mask] if sparse_mask is not None else None
        w = ops.router_weights(wg, mask, equal_weights=(self.router_type == "equal_weights"))
        down = w[:self.num_routers]
        mid = w[self.num_routers] if self.add_mid_block_router else None
        return down, mid

    @staticmethod
    def merge(down_lists, mid_list, down_w, mid_w, expert_masks, num_frames=None, inference_quirk=True):
        """Weighted sum of the ACTIVE experts' ControlNet outputs.
        inference_quirk=True reproduces the reference inference pipeline, where the weight of expert e is
        `w[k].repeat_interleave(num_frames)[e]` == w[k][0] for e < num_frames (SURVEY.md note N6);
        inference_quirk=False applies train.py's formula w[k][idx] (idx = position among the active experts)."""
        E = len(expert_masks)
        act = [e for e in range(E) if expert_masks[e]]
        if inference_quirk and (num_frames is None or int(num_frames) < 1):
            raise ValueError("merge(inference_quirk=True) needs num_frames (the pipeline indexes "
                             "w.repeat_interleave(num_frames)); pass inference_quirk=False for train.py's formula")

        def widx(e, k):
            if inference_quirk:
                return e // num_frames          # index into the E weights after repeat_interleave(num_frames)
            return k
        idx = [widx(e, k) for k, e in enumerate(act)]
        merged = [ops.router_merge([down_lists[k][r] for k in range(len(act))], down_w[r].contiguous(), idx)
                  for r in range(down_w.shape[0])]
        mid = None
        if mid_w is not None and mid_list is not None:
            mid = ops.router_merge([mid_list[k] for k in range(len(act))], mid_w.contiguous(), idx)
        return merged, mid
