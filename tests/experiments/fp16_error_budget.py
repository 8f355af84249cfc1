"""CPU emulation of the HIP path's rounding points on the fp32 oracle (TEST TOOLING; imports oracle/).

Every GEMM / conv A-operand of the HIP path is an fp16 rounding of an fp32 value; attention rounds Q, K, V, P and O to
fp16; boundary tensors are fp16.  This script applies those roundings to the oracle through forward pre-hooks and
reports rel-inf against the exact oracle, per class of rounding, for the config-5 miniature chain of
tests/test_gpu_e2e.py::test_multi_condition_router_pipeline_vs_oracle -- so the error budget of the chain can be
studied without GPU time.   python tests/experiments/fp16_error_budget.py [what ...]
"""
import os
import sys

import torch
import torch.nn.functional as F
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]
import cases  # noqa: E402
from oracle import blocks  # noqa: E402
from oracle.init import seeded_init, seeded_tensor  # noqa: E402
from oracle.controlnet import ControlNetOracle, MultiControlNetOracle  # noqa: E402
from oracle.adapter import ControlNetAdapterOracle  # noqa: E402
from oracle.router import RouterOracle, merge_inference  # noqa: E402

r16 = lambda x: x.half().float()
FLAGS = set()


def rel_inf(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def hook_operands(model, tag, skip=()):
    hs = []
    for name, m in model.named_modules():
        if isinstance(m, (nn.Conv2d, nn.Conv3d, nn.Linear)):
            if any(s in name for s in skip):
                continue
            small = isinstance(m, nn.Linear) and ("time_emb" in name or "time_embedding" in name or "linear_" in name)
            if small:
                continue       # linear_small_kernel consumes fp32 activations
            def pre(mod, args, name=name):
                if tag in FLAGS:
                    if "controlnet_down_blocks" in name or "controlnet_mid_block" in name:
                        if "exact_zero_conv" in FLAGS:
                            return None
                    return (r16(args[0]),) + tuple(args[1:])
                return None
            hs.append(m.register_forward_pre_hook(pre))
    return hs


_sdpa = F.scaled_dot_product_attention


def sdpa_q(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False):
    if "attn" not in FLAGS:
        return _sdpa(q, k, v, attn_mask=attn_mask)
    q, k, v = r16(q), r16(k), r16(v)
    s = (q @ k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
    m = s.amax(-1, keepdim=True)
    p = torch.exp(s - m)
    l = p.sum(-1, keepdim=True)
    o = (r16(p) @ v) / l
    return r16(o)


blocks.F.scaled_dot_product_attention = sdpa_q


def chain(quant):
    FLAGS.clear()
    FLAGS.update(quant)
    torch.set_grad_enabled(False)
    F_, N, hs = 4, 8, 8
    inp = [cases.controlnet_inputs(N=N, hs=hs, seed=700 + 10 * k) for k in range(3)]
    sample, ehs_c = inp[0]["sample"], inp[0]["encoder_hidden_states"]
    conds = [i["controlnet_cond"] for i in inp]
    t = torch.tensor(961.0)
    masks, act = [1, 0, 1], [0, 2]
    od, om = MULTI(sample, t, ehs_c, [conds[k] for k in act], [1.0, 1.0], skip_conv_in=True)
    if "cn_out16" in FLAGS:
        od = [[r16(x) for x in d] for d in od]
        om = [r16(x) for x in om]
    dw, mw = ROUTER(sparse_mask=masks)
    md, mm = merge_inference(od, om, dw, mw, masks, F_)
    if "merge16" in FLAGS:
        md, mm = [r16(x) for x in md], r16(mm)
    e_img = seeded_tensor((1, 1, 1024), 391)
    ro, rmid = AD(md, mid_block_res_sample=mm, num_frames=F_, timestep=t, encoder_hidden_states=e_img)
    return [x.clone() for x in od[0]] + [om[0].clone()], list(ro) + [rmid]


if __name__ == "__main__":
    nets = [seeded_init(ControlNetOracle(**cases.CONTROLNET_KW).eval(), seed=50 + k) for k in range(3)]
    MULTI = MultiControlNetOracle([nets[0], nets[2]])
    ROUTER = seeded_init(RouterOracle(num_experts=3, router_type="simple_weights", num_routers=12).eval(), seed=44)
    AD = seeded_init(ControlNetAdapterOracle(**cases.ADAPTER_VIDEO).eval(), seed=33)
    for n in (nets[0], nets[2]):
        hook_operands(n, "cn_ops")
    hook_operands(AD, "ad_ops")
    cn_ref, ref = chain(set())
    trials = [("all", {"cn_ops", "ad_ops", "attn", "cn_out16", "merge16"}),
              ("cn only (ops+attn+out16)", {"cn_ops", "attn", "cn_out16"}),
              ("cn ops only", {"cn_ops"}),
              ("attn only", {"attn"}),
              ("cn_out16 only", {"cn_out16"}),
              ("merge16 only", {"merge16"}),
              ("adapter ops only", {"ad_ops"}),
              ("all but cn_out16/merge16", {"cn_ops", "ad_ops", "attn"}),
              ("all, exact zero convs", {"cn_ops", "ad_ops", "attn", "cn_out16", "merge16", "exact_zero_conv"})]
    for name, q in trials:
        cn, out = chain(q)
        e_cn = [rel_inf(a, b) for a, b in zip(cn, cn_ref)]
        e = [rel_inf(a, b) for a, b in zip(out, ref)]
        print("%-32s cn max %.2e (mid %.2e) | chain max %.2e mid %.2e | %s" % (name, max(e_cn), e_cn[-1], max(e), e[-1], " ".join("%.1e" % x for x in e)))
    # ---- which ControlNet operand roundings carry the error (subset exact, the rest rounded) ----
    print("--- ControlNet operand subsets made exact (everything else rounded as in 'all') ---")
    import itertools
    groups = {"resnet convs": ("resnets.",), "downsamplers": ("downsamplers",), "proj_in/out": ("proj_in", "proj_out"),
              "attn linears": ("attn1", "attn2"), "ff": (".ff.",), "cond embed": ("controlnet_cond_embedding",),
              "zero convs": ("controlnet_down_blocks", "controlnet_mid_block")}
    EXACT = []
    for n in (nets[0], nets[2]):
        for name, m in n.named_modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                def post_pre(mod, args, name=name):
                    return None
    # re-hook with an exclusion list that can change at run time
    for n in (nets[0], nets[2]):
        for m in n.modules():
            m._forward_pre_hooks.clear()
    def hook2(model):
        for name, m in model.named_modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)) and not ("time_emb" in name or "time_embedding" in name):
                def pre(mod, args, name=name):
                    if "cn_ops" in FLAGS and not any(s in name for s in EXACT):
                        return (r16(args[0]),) + tuple(args[1:])
                    return None
                m.register_forward_pre_hook(pre)
    for n in (nets[0], nets[2]):
        hook2(n)
    allq = {"cn_ops", "ad_ops", "attn", "cn_out16", "merge16"}
    for gname, pats in list(groups.items()) + [("resnet convs + ff", ("resnets.", ".ff.")), ("all convs", ("resnets.", "downsamplers", "proj_in", "proj_out", "controlnet_")),
                                               ("everything", ("",))]:
        EXACT[:] = list(pats)
        cn, out = chain(allq)
        e_cn = [rel_inf(a, b) for a, b in zip(cn, cn_ref)]
        e = [rel_inf(a, b) for a, b in zip(out, ref)]
        print("exact %-22s cn max %.2e (mid %.2e) | chain max %.2e mid %.2e" % (gname, max(e_cn), e_cn[-1], max(e), e[-1]))
