"""CPU restatement of the MoE router and the caller-side merge (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Router: model/ctrl_router.py:9-112 (EqualWeights :9-22, SimpleWeights :26-40, ControlNetRouter :49-112), without
the hard-coded .cuda() calls (:21,:38).  Merge: i2vgen_xl/pipelines/i2vgen_xl_controlnet_adapter_pipeline.py:1000-1022
(inference; note quirk N6: `w[k].repeat_interleave(num_frames)[e]` equals w[k][0] whenever e < num_frames) and
train.py:1262-1276 (training formula w[k][e]).
"""
import torch
import torch.nn.functional as F
from torch import nn


class _SimpleWeights(nn.Module):
    def __init__(self, num_experts):
        super().__init__()
        self.wg = nn.Linear(1, num_experts, bias=False)

    def forward(self):
        return self.wg(torch.ones(1, 1, dtype=self.wg.weight.dtype))


class _EqualWeights(nn.Module):
    def __init__(self, num_experts):
        super().__init__()
        self.num_experts = num_experts

    def forward(self):
        return torch.zeros(1, self.num_experts)


class RouterOracle(nn.Module):
    def __init__(self, num_experts=2, router_type="simple_weights", num_routers=12, add_mid_block_router=True):
        super().__init__()
        mk = _SimpleWeights if router_type == "simple_weights" else _EqualWeights
        self.num_experts, self.num_routers, self.router_type = num_experts, num_routers, router_type
        self.down_blocks_router = nn.ModuleList([mk(num_experts) for _ in range(num_routers)])
        self.mid_block_router = mk(num_experts) if add_mid_block_router else None

    def forward(self, router_input=None, sparse_mask=None, fixed_weights=None):
        down = [r() for r in self.down_blocks_router]
        mid = self.mid_block_router() if self.mid_block_router is not None else None
        if sparse_mask is not None:
            for i, m in enumerate(sparse_mask):
                if m == 0:
                    if mid is not None:
                        mid[0, i] -= 1e6
                    for d in down:
                        d[0, i] -= 1e6
        dw = F.softmax(torch.cat(down), dim=-1)
        mw = F.softmax(mid, dim=-1).squeeze(0) if mid is not None else None
        return dw, mw


def merge_inference(down_lists, mid_list, down_w, mid_w, masks, num_frames):
    """i2vgen_xl pipeline :1000-1022, verbatim semantics including quirk N6."""
    E = len(masks)
    mid = None
    if mid_w is not None:
        mid, k = 0, 0
        for e in range(E):
            if masks[e]:
                mid = mid + mid_list[k] * mid_w.repeat_interleave(num_frames, dim=0)[e]
                k += 1
    merged = []
    for r in range(down_w.shape[0]):
        acc, k = 0, 0
        for e in range(E):
            if masks[e]:
                acc = acc + down_lists[k][r] * down_w[r].repeat_interleave(num_frames, dim=0)[e]
                k += 1
        merged.append(acc)
    return merged, mid


def merge_training(down_lists, mid_list, down_w, mid_w, masks):
    """train.py:1262-1276: the k-th ACTIVE control type is weighted by w[slot][k] (idx of the active list)."""
    K = len(down_lists)
    mid = sum(mid_list[k] * mid_w[k] for k in range(K)) if mid_w is not None else None
    merged = [sum(down_lists[k][r] * down_w[r][k] for k in range(K)) for r in range(down_w.shape[0])]
    return merged, mid
