"""Which ControlNet convolutions need exact ([hi | lo] split) A operands -- per conv kind instead of per level (TEST TOOLING; imports oracle/).

Same machinery as fp16_error_budget.py (the HIP path's rounding points applied to the fp32 oracle on the CPU, config-5 miniature chain):
the baseline is what ships (CTRL_CN_SPLIT_LEVELS=3: every conv of down blocks 0-2 and every zero-conv exact, the 8x8 level and all
Linear operands plain fp16); each trial makes ONE more group of convolutions plain and reports the ControlNet-output and chain errors.
    python tests/experiments/split_per_conv.py
"""
import os
import sys

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fp16_error_budget as B  # noqa: E402
from fp16_error_budget import cases, seeded_init, ControlNetOracle, MultiControlNetOracle, RouterOracle, ControlNetAdapterOracle, r16, rel_inf  # noqa: E402

PLAIN = []          # name patterns of convolutions that are rounded although they lie in the exact set


def in_exact_set(name):
    lvl = [("down_blocks.%d." % i) in name for i in range(3)]
    return any(lvl) or "controlnet_down_blocks" in name or "controlnet_mid_block" in name


def hook(model):
    for name, m in model.named_modules():
        if isinstance(m, (nn.Conv2d, nn.Linear)) and not ("time_emb" in name or "time_embedding" in name):
            def pre(mod, args, name=name, conv=isinstance(m, nn.Conv2d)):
                if "cn_ops" not in B.FLAGS:
                    return None
                exact = conv and in_exact_set(name) and not any(p in name for p in PLAIN)
                return None if exact else (r16(args[0]),) + tuple(args[1:])
            m.register_forward_pre_hook(pre)


if __name__ == "__main__":
    nets = [seeded_init(ControlNetOracle(**cases.CONTROLNET_KW).eval(), seed=50 + k) for k in range(3)]
    B.MULTI = MultiControlNetOracle([nets[0], nets[2]])
    B.ROUTER = seeded_init(RouterOracle(num_experts=3, router_type="simple_weights", num_routers=12).eval(), seed=44)
    B.AD = seeded_init(ControlNetAdapterOracle(**cases.ADAPTER_VIDEO).eval(), seed=33)
    for n in (nets[0], nets[2]):
        hook(n)
    B.hook_operands(B.AD, "ad_ops")
    cn_ref, ref = B.chain(set())
    allq = {"cn_ops", "ad_ops", "attn", "cn_out16", "merge16"}
    names = sorted({n for n, m in nets[0].named_modules() if isinstance(m, nn.Conv2d) and in_exact_set(n)})
    print("%d convolutions in the exact set, e.g. %s" % (len(names), ", ".join(names[:6])))
    trials = [("baseline (ships)", []),
              ("resnet conv1 plain", [".conv1"]), ("resnet conv2 plain", [".conv2"]), ("resnet shortcuts plain", ["conv_shortcut"]),
              ("down-samplers plain", ["downsamplers"]), ("proj_in plain", ["proj_in"]), ("proj_out plain", ["proj_out"]),
              ("proj_in + proj_out plain", ["proj_in", "proj_out"]), ("zero-convs plain", ["controlnet_down_blocks", "controlnet_mid_block"]),
              ("level 0 (320 ch) resnets plain", ["down_blocks.0.resnets"]), ("level 1 (640 ch) resnets plain", ["down_blocks.1.resnets"]),
              ("level 2 (1280 ch) resnets plain", ["down_blocks.2.resnets"]),
              ("level 2 resnets + proj_in/out plain", ["down_blocks.2.resnets", "proj_in", "proj_out"]),
              ("shortcuts + down-samplers + proj_in/out plain", ["conv_shortcut", "downsamplers", "proj_in", "proj_out"]),
              ("level 1 + 2 resnet conv1 / conv2 plain", ["down_blocks.%d.resnets.%d.conv%d" % (l, r, c) for l in (1, 2) for r in (0, 1) for c in (1, 2)]),
              ("level 1 + 2 resnets plain (shortcuts too)", ["down_blocks.1.resnets", "down_blocks.2.resnets"]),
              ("level 1 + 2 resnets + level 0 conv2 plain", ["down_blocks.1.resnets", "down_blocks.2.resnets", "down_blocks.0.resnets.0.conv2",
                                                             "down_blocks.0.resnets.1.conv2"]),
              ("all resnet conv1 / conv2 plain (shortcuts exact)", [".conv1", ".conv2"]),
              ("all resnets plain", ["resnets."])]
    for tname, pats in trials:
        PLAIN[:] = pats
        cn, out = B.chain(allq)
        e_cn = [rel_inf(a, b) for a, b in zip(cn, cn_ref)]
        e = [rel_inf(a, b) for a, b in zip(out, ref)]
        print("%-48s ControlNet outputs max %.2e (mid %.2e) | chain max %.2e (mid %.2e)" % (tname, max(e_cn), e_cn[-1], max(e), e[-1]))
