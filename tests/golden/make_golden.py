"""Generates tests/golden/*.pt by running the REFERENCE'S OWN hot-path files from /root/reference
(controlnet/controlnet.py, model/ctrl_adapter.py, model/adapter_spatial_temporal.py, model/resnet_block_2d.py,
model/ctrl_router.py) unmodified, on top of a REAL `diffusers` when one is importable, else on oracle/_shim (a minimal
`diffusers` whose blocks are oracle/blocks.py).  Every file records which (`provenance`: tests/golden/blocks_source.py,
tests/golden/README.md has the recipe that closes the pin on a box with `pip install diffusers==0.27.2`).

Run in the build container only (the GPU box has no /root/reference):
    python tests/golden/make_golden.py [--require-real-diffusers] [--out DIR] [per_clip_context]
Stored per output tensor: shape, float64 sum and abs-sum, and <=4096 evenly strided fp32 samples; for a selection of
outputs of every case (all 13 of the plain ControlNet run, the largest / deepest adapter slots, every mid block) the
WHOLE fp32 tensor as well, so those are pinned element by element.  Inputs and weights are regenerated from seeds (tests/golden/cases.py,
oracle/init.py), keyed by parameter NAME so a state-dict key mismatch would surface as a value mismatch.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path[:0] = [REF, ROOT, os.path.join(ROOT, "tests", "golden")]
import blocks_source  # noqa: E402
ARGS = [a for a in sys.argv[1:]]
BLOCKS = blocks_source.select(require_real="--require-real-diffusers" in ARGS)      # prepares sys.path for `import diffusers`

torch.Tensor.cuda = lambda self, *a, **k: self        # model/ctrl_router.py:21,38 hard-code .cuda()

from oracle.init import seeded_init  # noqa: E402
import cases  # noqa: E402


def digest(t, full=False):
    """digest of one output tensor; full=True additionally stores the WHOLE tensor (fp32: an fp16 copy would cost up to
    4.9e-4 of the 1e-3 rel-inf budget) so that at least one output of every case is pinned element by element"""
    t = t.detach().float().contiguous()
    flat = t.reshape(-1)
    step = max(1, flat.numel() // 4096)
    d = dict(shape=list(t.shape), sum=float(flat.double().sum()), abssum=float(flat.double().abs().sum()),
             step=step, samples=flat[::step][:4096].clone())
    if full:
        d["full"] = t.clone()
    return d


def per_clip_context(out_dir):
    """G3c: per-frame encoder states with several clips (tests/golden/cases.py:PER_CLIP_CONTEXT), location A of the video adapter"""
    from model.ctrl_adapter import ControlNetAdapter
    gq = {}
    for tag in cases.PER_CLIP_CONTEXT:
        cfg, clips, frames, downs, ehs, ts = cases.per_clip_context_inputs(tag)
        ad = seeded_init(ControlNetAdapter(**cfg).eval(), seed=36)
        out, mid = ad(downs, sparsity_masking=None, num_frames=frames, timestep=ts, encoder_hidden_states=ehs)
        assert mid is None
        gq[tag] = {"keys": sorted(ad.state_dict().keys()), "out": [digest(o, full=(i == 0)) for i, o in enumerate(out)]}
        del ad
    gq["__provenance__"] = blocks_source.provenance(BLOCKS)
    torch.save(gq, os.path.join(out_dir, "adapter_per_clip_context.pt"))
    print("per-clip context: %s" % ", ".join(k for k in gq if not k.startswith("__")))


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    if "--out" in ARGS:
        out_dir = os.path.abspath(ARGS[ARGS.index("--out") + 1])
        os.makedirs(out_dir, exist_ok=True)
    print("blocks: %s" % BLOCKS)
    prov = blocks_source.provenance(BLOCKS)
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    if "per_clip_context" in ARGS:      # only this file (the others are unchanged)
        return per_clip_context(out_dir)

    from controlnet.controlnet import ControlNetModel
    from model.ctrl_adapter import ControlNetAdapter
    from model.ctrl_router import ControlNetRouter

    # ---- G1: ControlNet (SD-1.5 architecture), tiny latent grid ----
    net = seeded_init(ControlNetModel(**cases.CONTROLNET_KW).eval(), seed=11)
    nparams = sum(p.numel() for p in net.parameters())
    keys = sorted(net.state_dict().keys())
    g1 = {"n_params": nparams, "keys": keys, "runs": {}, "provenance": prov}
    inp = cases.controlnet_inputs()
    for tag, kw in {"plain": {}, "scale0.5": dict(conditioning_scale=0.5), "skip_conv_in": dict(skip_conv_in=True),
                    "skip_time_emb": dict(skip_time_emb=True), "guess": dict(guess_mode=True)}.items():
        down, mid = net(inp["sample"], inp["timestep"], encoder_hidden_states=inp["encoder_hidden_states"],
                        controlnet_cond=inp["controlnet_cond"], return_dict=False, **kw)
        g1["runs"][tag] = [digest(d, full=(tag == "plain")) for d in down] + [digest(mid, full=True)]
    inp = cases.controlnet_inputs_nonsquare()
    down, mid = net(inp["sample"], inp["timestep"], encoder_hidden_states=inp["encoder_hidden_states"],
                    controlnet_cond=inp["controlnet_cond"], return_dict=False)
    g1["nonsquare_n1"] = [digest(d, full=(i in (0, 8))) for i, d in enumerate(down)] + [digest(mid, full=True)]
    torch.save(g1, os.path.join(out_dir, "controlnet_sd15.pt"))
    print("controlnet: %d params (%.1f M), %d keys" % (nparams, nparams / 1e6, len(keys)))
    del net

    # ---- G2: SDXL adapter (spatial, up-sampling x2) ----
    ad = seeded_init(ControlNetAdapter(**cases.ADAPTER_SDXL).eval(), seed=22)
    downs, _ = cases.pyramid_inputs(N=2, h0=8, seed=200, with_mid=False)
    ehs = cases.seeded_tensor((2, 77, 2048), 290)
    out, mid = ad(downs, sparsity_masking=None, num_frames=1, timestep=torch.tensor(749.0), encoder_hidden_states=ehs)
    assert mid is None
    torch.save({"n_params": sum(p.numel() for p in ad.parameters()), "keys": sorted(ad.state_dict().keys()), "provenance": prov,
                "out": [digest(o, full=(i in (0, 4, 8))) for i, o in enumerate(out)]}, os.path.join(out_dir, "adapter_sdxl.pt"))
    print("adapter sdxl: %.1f M params" % (sum(p.numel() for p in ad.parameters()) / 1e6))
    del ad

    # ---- G3: video adapter (all four sub-modules, A-D + M), 2 clips x 4 frames ----
    ad = seeded_init(ControlNetAdapter(**cases.ADAPTER_VIDEO).eval(), seed=33)
    downs, midin = cases.pyramid_inputs(N=8, h0=8, seed=300, with_mid=True)
    ehs = cases.seeded_tensor((1, 1, 1024), 390)
    out, mid = ad(downs, mid_block_res_sample=midin, sparsity_masking=None, num_frames=4,
                  timestep=torch.tensor(961.0), encoder_hidden_states=ehs)
    torch.save({"n_params": sum(p.numel() for p in ad.parameters()), "keys": sorted(ad.state_dict().keys()), "provenance": prov,
                "out": [digest(o, full=(i in (0, 5, 8, 11))) for i, o in enumerate(out)] + [digest(mid, full=True)]},
               os.path.join(out_dir, "adapter_video.pt"))
    print("adapter video: %.1f M params" % (sum(p.numel() for p in ad.parameters()) / 1e6))
    del ad

    # ---- G3b: configuration variants outside the shipped YAMLs (tests/golden/cases.py:ADAPTER_VARIANTS) ----
    gv = {}
    for tag in cases.ADAPTER_VARIANTS:
        cfg, io, downs, midin, ehs = cases.variant_inputs(tag)
        ad = seeded_init(ControlNetAdapter(**cfg).eval(), seed=77)
        out, mid = ad(downs, mid_block_res_sample=midin, sparsity_masking=None, num_frames=io["frames"],
                      timestep=cases.variant_timestep(io), encoder_hidden_states=ehs)
        gv[tag] = {"keys": sorted(ad.state_dict().keys()), "n_params": sum(p.numel() for p in ad.parameters()),
                   "out": [digest(o, full=(o.abs().max() > 0 and o.numel() <= 50000)) for o in out] + ([digest(mid, full=True)] if mid is not None else [])}
        del ad
    gv["__provenance__"] = prov
    torch.save(gv, os.path.join(out_dir, "adapter_variants.pt"))
    print("adapter variants: %s" % ", ".join(k for k in gv if not k.startswith("__")))

    # ---- G4: router ----
    r = seeded_init(ControlNetRouter(num_experts=3, router_type="simple_weights", num_routers=12).eval(), seed=44)
    g4 = {"keys": sorted(r.state_dict().keys()), "runs": {}, "provenance": prov}
    for tag, mask in {"all": [1, 1, 1], "m101": [1, 0, 1], "none": None}.items():
        dw, mw = r(sparse_mask=mask)
        g4["runs"][tag] = dict(down=dw.clone(), mid=mw.clone())
    r2 = ControlNetRouter(num_experts=2, router_type="equal_weights", num_routers=12).eval()
    dw, mw = r2(sparse_mask=[1, 1])
    g4["equal"] = dict(down=dw.clone(), mid=mw.clone())
    torch.save(g4, os.path.join(out_dir, "router.pt"))
    print("router ok")
    per_clip_context(out_dir)


if __name__ == "__main__":
    main()
