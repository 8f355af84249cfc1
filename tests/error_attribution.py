"""Where the 1e-3 goes: error of the HIP chains against the fp32 oracle under ONE precision selection changed at a time.

Test infrastructure (it runs the oracle), not a collected test: `python tests/error_attribution.py [--chains svd16,sdxl1,mini5]
[--out gpurun_out/r06_error_attribution.md]` on a GPU box.  Three chains of tests/test_gpu_e2e.py are rebuilt per selection -- the
precision selections are read when a plan is created (csrc/policy.h, csrc/plan_controlnet.cpp:cn_split_levels, plan_common.h:
adapter_tok_f16) -- and compared with oracle outputs that are computed once per chain:

  svd16   BASELINE config 3 at the benched shapes (N = 32 frames, 64^2 latents, skip_conv_in, video adapter A-D + M)
  sdxl1   BASELINE config 2, one image at the benched shapes (1024^2 -> 64^2 ControlNet -> 128^2 SDXL adapter)
  mini5   config 5 in miniature (8 x 8 latents, two active of three nets -> router -> N6 merge -> video adapter): the chain whose
          1.001e-3 kept CTRL_CN_SPLIT_RESNET_LEVELS=1 from becoming the default in round 5

Per selection and chain: max rel-inf (max|a-b| / max|b|, the asserted metric) over the ControlNet outputs and over the chain outputs, the
tensor that sets it, the max rel-L2 (sqrt(sum (a-b)^2 / sum b^2): what the rel-inf maximum fluctuates around), and the eager GPU time of
the two forwards (HIP events, median of 5) as the price of the selection.  The `exact` row has every selectable rounding point at its
most precise setting: what is left there is the operand format of the token GEMMs and of attention, which no selection changes."""
import argparse
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "golden"))

import cases                                            # noqa: E402
from conftest import rel_inf                            # noqa: E402
from oracle.init import seeded_init, seeded_tensor      # noqa: E402

SELECTIONS = [
    ("default", {}),
    ("exact (split 5 levels, fp32 token stream)", {"CTRL_CN_SPLIT_LEVELS": "5", "CTRL_ADAPTER_TOK_F16": "0"}),
    ("ControlNet: split levels 5 (8^2 level + mid too)", {"CTRL_CN_SPLIT_LEVELS": "5"}),
    ("ControlNet: split levels 4", {"CTRL_CN_SPLIT_LEVELS": "4"}),
    ("ControlNet: split levels 2", {"CTRL_CN_SPLIT_LEVELS": "2"}),
    ("ControlNet: split levels 1", {"CTRL_CN_SPLIT_LEVELS": "1"}),
    ("ControlNet: split levels 0 (zero-convs only)", {"CTRL_CN_SPLIT_LEVELS": "0"}),
    ("ControlNet: ResNet 3x3 split on levels < 2", {"CTRL_CN_SPLIT_RESNET_LEVELS": "2"}),
    ("ControlNet: ResNet 3x3 split on level 0 only", {"CTRL_CN_SPLIT_RESNET_LEVELS": "1"}),
    ("ControlNet: ResNet 3x3 never split", {"CTRL_CN_SPLIT_RESNET_LEVELS": "0"}),
    ("ControlNet: no split operands at all", {"CTRL_CN_SPLIT": "0"}),
    ("adapter: fp32 spatial token stream", {"CTRL_ADAPTER_TOK_F16": "0"}),
    ("adapter: fp16 token stream forced (video too)", {"CTRL_ADAPTER_TOK_F16": "f"}),
    ("adapter: conv1 -> GroupNorm intermediate fp16", {"CTRL_ADAPTER_H1_F16": "1"}),
    ("adapter: two-launch feed-forward", {"CTRL_FF_FUSED": "0"}),
    ("both: fp16 residual streams", {"CTRL_STREAM_F32": "0"}),
]


def rel_l2(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).pow(2).sum().sqrt() / b.pow(2).sum().sqrt().clamp_min(1e-30)).item()


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return out, sorted(ms)[len(ms) // 2]


class Svd16:
    name = "svd16"

    def __init__(self):
        from oracle.controlnet import ControlNetOracle
        from oracle.adapter import ControlNetAdapterOracle
        self.F, N = 16, 32
        self.cfg = dict(cases.ADAPTER_VIDEO, backbone_model_name="svd", num_frames=self.F)
        self.lat = seeded_tensor((N, 4, 64, 64), 2001)
        self.ehs = seeded_tensor((N, 77, 768), 2002)
        self.cond = seeded_tensor((N, 3, 512, 512), 2003, kind="uniform")
        self.e_img = seeded_tensor((1, 1, 1024), 2004)
        self.t = torch.tensor(961.0)
        oc = seeded_init(ControlNetOracle(**cases.CONTROLNET_KW).eval(), seed=11)
        oa = seeded_init(ControlNetAdapterOracle(**self.cfg).eval(), seed=33)
        rd, rm = oc(self.lat, self.t, self.ehs, self.cond, skip_conv_in=True)
        ro, rom = oa(rd, mid_block_res_sample=rm, num_frames=self.F, timestep=self.t, encoder_hidden_states=self.e_img)
        self.ref_cn = list(rd) + [rm]
        self.ref_chain = list(ro) + [rom]

    def run(self, P, gpu):
        cn = seeded_init(P.ControlNetModel(**cases.CONTROLNET_KW), seed=11).to(gpu)
        ad = seeded_init(P.ControlNetAdapter(**self.cfg), seed=33).to(gpu)
        a = [x.half().to(gpu) for x in (self.lat, self.ehs, self.cond, self.e_img)]

        def fwd():
            d, m = cn(a[0], self.t, a[1], a[2], return_dict=False, skip_conv_in=True)
            o, om = ad(d, mid_block_res_sample=m, num_frames=self.F, timestep=self.t, encoder_hidden_states=a[3])
            return list(d) + [m], list(o) + [om]
        (g_cn, g_chain), ms = timed(fwd)
        return g_cn, g_chain, ms, "%s | %s" % (cn.selection, ad.selection)


class Sdxl1:
    name = "sdxl1"

    def __init__(self):
        from oracle.controlnet import ControlNetOracle
        from oracle.adapter import ControlNetAdapterOracle
        self.lat = seeded_tensor((1, 4, 128, 128), 1)
        self.ehs_c = seeded_tensor((1, 77, 768), 2)
        self.cond = seeded_tensor((1, 3, 512, 512), 3, kind="uniform")
        self.ehs_a = seeded_tensor((1, 77, 2048), 4)
        self.t = torch.tensor(499.0)
        oc = seeded_init(ControlNetOracle(**cases.CONTROLNET_KW).eval(), seed=11)
        oa = seeded_init(ControlNetAdapterOracle(**cases.ADAPTER_SDXL).eval(), seed=22)
        rd, rm = oc(torch.nn.functional.adaptive_avg_pool2d(self.lat, (64, 64)), self.t, self.ehs_c, self.cond)
        ro, _ = oa(rd, num_frames=1, timestep=self.t, encoder_hidden_states=self.ehs_a)
        self.ref_cn = list(rd) + [rm]
        self.ref_chain = list(ro[:9])

    def run(self, P, gpu):
        cn = seeded_init(P.ControlNetModel(**cases.CONTROLNET_KW), seed=11).to(gpu)
        ad = seeded_init(P.ControlNetAdapter(**cases.ADAPTER_SDXL), seed=22).to(gpu)
        a = [x.half().to(gpu) for x in (self.lat, self.ehs_c, self.cond, self.ehs_a)]

        def fwd():
            s = P.pool_latents(a[0], (64, 64))
            d, m = cn(s, self.t, a[1], a[2], return_dict=False)
            o, _ = ad(d, num_frames=1, timestep=self.t, encoder_hidden_states=a[3])
            return list(d) + [m], list(o[:9])
        (g_cn, g_chain), ms = timed(fwd)
        return g_cn, g_chain, ms, "%s | %s" % (cn.selection, ad.selection)


class Mini5:
    name = "mini5"

    def __init__(self):
        from oracle.controlnet import ControlNetOracle, MultiControlNetOracle
        from oracle.adapter import ControlNetAdapterOracle
        from oracle.router import RouterOracle, merge_inference
        self.F, N, hs = 4, 8, 8
        inp = [cases.controlnet_inputs(N=N, hs=hs, seed=700 + 10 * k) for k in range(3)]
        self.sample, self.ehs = inp[0]["sample"], inp[0]["encoder_hidden_states"]
        self.conds = [i["controlnet_cond"] for i in inp]
        self.t = torch.tensor(961.0)
        self.masks, self.act = [1, 0, 1], [0, 2]
        self.e_img = seeded_tensor((1, 1, 1024), 391)
        nets = [seeded_init(ControlNetOracle(**cases.CONTROLNET_KW).eval(), seed=50 + k) for k in range(3)]
        od, om = MultiControlNetOracle([nets[k] for k in self.act])(self.sample, self.t, self.ehs, [self.conds[k] for k in self.act], [1.0, 1.0],
                                                                    skip_conv_in=True)
        dw, mw = seeded_init(RouterOracle(num_experts=3, router_type="simple_weights", num_routers=12).eval(), seed=44)(sparse_mask=self.masks)
        md, mm = merge_inference(od, om, dw, mw, self.masks, self.F)
        ro, rmid = seeded_init(ControlNetAdapterOracle(**cases.ADAPTER_VIDEO).eval(), seed=33)(
            md, mid_block_res_sample=mm, num_frames=self.F, timestep=self.t, encoder_hidden_states=self.e_img)
        self.ref_cn = [x for dd in od for x in dd] + list(om)
        self.ref_chain = list(ro) + [rmid]

    def run(self, P, gpu):
        nets = [seeded_init(P.ControlNetModel(**cases.CONTROLNET_KW), seed=50 + k).to(gpu) for k in self.act]
        multi = P.MultiControlNetModel(nets)
        router = seeded_init(P.ControlNetRouter(num_experts=3, router_type="simple_weights", num_routers=12), seed=44).to(gpu)
        ad = seeded_init(P.ControlNetAdapter(**cases.ADAPTER_VIDEO), seed=33).to(gpu)
        s, e, ei = self.sample.half().to(gpu), self.ehs.half().to(gpu), self.e_img.half().to(gpu)
        cs = [self.conds[k].half().to(gpu) for k in self.act]

        def fwd():
            gd, gm = multi(s, self.t, e, cs, [1.0, 1.0], return_dict=False, skip_conv_in=True)
            gdw, gmw = router(sparse_mask=self.masks)
            md, mm = router.merge(gd, gm, gdw, gmw, self.masks, num_frames=self.F, inference_quirk=True)
            o, om = ad(md, mid_block_res_sample=mm, num_frames=self.F, timestep=self.t, encoder_hidden_states=ei)
            return [x for dd in gd for x in dd] + list(gm), list(o) + [om]
        (g_cn, g_chain), ms = timed(fwd)
        return g_cn, g_chain, ms, "%s | %s" % (nets[0].selection, ad.selection)


CHAINS = {"svd16": Svd16, "sdxl1": Sdxl1, "mini5": Mini5}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", default="svd16,sdxl1,mini5")
    ap.add_argument("--out", default="gpurun_out/r06_error_attribution.md")
    ap.add_argument("--only", default=None, help="substring filter on the selection names")
    a = ap.parse_args()
    torch.set_grad_enabled(False)
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    import ctrl_adapter_amd as P
    from ctrl_adapter_amd import ops
    gpu = torch.device("cuda:0")
    lines = ["# Error attribution: one precision selection changed at a time (tests/error_attribution.py)", "",
             "rel-inf = max|a-b| / max|b| per tensor, maximum over the tensors of the group; rel-L2 likewise; `ms` = eager GPU time of the chain's",
             "forwards on this box (HIP events, median of 5).  Oracle = fp32 CPU restatement, same seeded weights and inputs.", ""]
    for cname in a.chains.split(","):
        chain = CHAINS[cname]()
        lines += ["## %s" % cname, "", "| selection | ControlNet rel-inf | chain rel-inf | set by | ControlNet rel-L2 | chain rel-L2 | ms |", "|---|---|---|---|---|---|---|"]
        base = None
        for sname, pol in SELECTIONS:
            if a.only and a.only not in sname and sname != "default":
                continue
            prev = {k: ops.set_policy(k, v) for k, v in pol.items()}
            try:
                g_cn, g_chain, ms, sel = chain.run(P, gpu)
            finally:
                for k, v in prev.items():
                    ops.set_policy(k, v)
            e_cn = [rel_inf(x, y) for x, y in zip(g_cn, chain.ref_cn)]
            e_ch = [rel_inf(x, y) for x, y in zip(g_chain, chain.ref_chain)]
            l_cn = max(rel_l2(x, y) for x, y in zip(g_cn, chain.ref_cn))
            l_ch = max(rel_l2(x, y) for x, y in zip(g_chain, chain.ref_chain))
            worst = max(range(len(e_ch)), key=lambda i: e_ch[i])
            if base is None:
                base = (max(e_cn), max(e_ch), l_cn, l_ch, ms)
            row = "| %s | %.2e | %.2e (%+.1e) | out[%d] | %.2e | %.2e (%+.1e) | %.2f (%+.2f) |" % (
                sname, max(e_cn), max(e_ch), max(e_ch) - base[1], worst, l_cn, l_ch, l_ch - base[3], ms, ms - base[4])
            print(cname, row, flush=True)
            print("   selection:", sel, flush=True)
            lines.append(row)
            del g_cn, g_chain
            torch.cuda.empty_cache()
        lines.append("")
        del chain
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as fh:
        fh.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
