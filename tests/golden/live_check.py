"""Randomised live comparison of the oracle with the REFERENCE'S OWN files (build container only: needs /root/reference).

The goldens pin a fixed set of configurations; this script draws others -- adapter locations, adapters per location, number of
blocks, which of the four sub-modules exist, clips x frames, broadcast / per-sample / 2-D encoder states, scalar / per-sample
timesteps; ControlNet scale, guess mode, skip flags, global pooling -- and runs model/ctrl_adapter.py + model/adapter_spatial_temporal.py
+ model/resnet_block_2d.py + controlnet/controlnet.py (over oracle/_shim, like make_golden.py) against oracle/ on the same seeded
weights and inputs.  Prints one line per case and exits non-zero on the first mismatch > 2e-5 (fp32 vs fp32).

    python tests/golden/live_check.py [n_cases [first_seed]]
"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path[:0] = [REF, ROOT, os.path.join(ROOT, "tests", "golden")]
import blocks_source  # noqa: E402
BLOCKS = blocks_source.select(require_real="--require-real-diffusers" in sys.argv)      # a real diffusers when there is one, else the shim
sys.argv = [a for a in sys.argv if a != "--require-real-diffusers"]
torch.Tensor.cuda = lambda self, *a, **k: self

from oracle.init import seeded_init, seeded_tensor  # noqa: E402
import cases  # noqa: E402

TOL = 2e-5


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def adapter_case(rng, seed):
    from model.ctrl_adapter import ControlNetAdapter
    from oracle.adapter import ControlNetAdapterOracle
    video = rng.random() < 0.6
    cfg = dict(cases.ADAPTER_VIDEO if video else cases.ADAPTER_SDXL)
    locs = [l for l in "ABCD" if rng.random() < 0.5] or ["A"]
    for l in "ABCD":
        cfg["add_adapter_location_" + l] = l in locs
    cfg["add_adapter_location_M"] = video and rng.random() < 0.5
    cfg["num_adapters_per_location"] = rng.choice([1, 2, 3])
    cfg["num_blocks"] = rng.choice([1, 1, 2])
    if video:
        flags = [rng.random() < 0.7 for _ in range(4)]
        if not any(flags):
            flags[0] = True
        (cfg["add_spatial_resnet"], cfg["add_temporal_resnet"], cfg["add_spatial_transformer"], cfg["add_temporal_transformer"]) = flags
        clips, frames = rng.choice([(1, 3), (2, 2), (2, 3), (3, 2)])
    else:
        sr, st = rng.choice([(True, True), (True, False), (False, True)])
        cfg["add_spatial_resnet"], cfg["add_spatial_transformer"] = sr, st
        clips, frames = rng.choice([1, 2, 3]), 1
    N = clips * frames
    cross = cfg["cross_attention_dim"]
    kind = rng.choice(["broadcast", "per_sample"]) if video else rng.choice(["tokens", "tokens", "vector2d"])
    if kind == "broadcast":
        ehs = seeded_tensor((1, 1, cross), seed + 1)
    elif kind == "per_sample":
        ehs = seeded_tensor((N, 1, cross), seed + 1)
    elif kind == "tokens":
        ehs = seeded_tensor((N, rng.choice([5, 77]), cross), seed + 1)
    else:
        ehs = seeded_tensor((N, cross), seed + 1)
    ts = torch.tensor(float(rng.choice([1, 333, 961]))) if rng.random() < 0.5 else torch.tensor([float(rng.randrange(1, 999)) for _ in range(N)])
    downs, mid = cases.pyramid_inputs(N=N, h0=rng.choice([4, 8]), seed=seed + 2, with_mid=cfg["add_adapter_location_M"])
    ref = seeded_init(ControlNetAdapter(**cfg).eval(), seed=seed)
    ora = seeded_init(ControlNetAdapterOracle(**cfg).eval(), seed=seed)
    assert sorted(ref.state_dict()) == sorted(ora.state_dict()), "state-dict keys differ"
    ro, rm = ref(downs, mid_block_res_sample=mid, sparsity_masking=None, num_frames=frames, timestep=ts, encoder_hidden_states=ehs)
    oo, om = ora(downs, mid_block_res_sample=mid, num_frames=frames, timestep=ts, encoder_hidden_states=ehs)
    errs = [rel(a, b) for a, b in zip(oo, ro) if b.abs().max() > 0]
    assert all(torch.equal(a, b) for a, b in zip(oo, ro) if b.abs().max() == 0), "zero slots differ"
    if rm is not None:
        errs.append(rel(om, rm))
    else:
        assert om is None
    desc = "adapter %s loc %s%s x%d blocks %d sr/tr/st/tt %d%d%d%d clips %d frames %d ehs %s t %s" % (
        "video" if video else "sdxl", "".join(locs), "M" if cfg["add_adapter_location_M"] else "", cfg["num_adapters_per_location"],
        cfg["num_blocks"], cfg["add_spatial_resnet"], cfg.get("add_temporal_resnet", 0), cfg["add_spatial_transformer"],
        cfg.get("add_temporal_transformer", 0), clips, frames, kind, "scalar" if ts.dim() == 0 else "per-sample")
    return desc, max(errs)


_nets = {}


def controlnet_case(rng, seed):
    from controlnet.controlnet import ControlNetModel
    from oracle.controlnet import ControlNetOracle
    gp = rng.random() < 0.3
    key = gp
    if key not in _nets:          # 360 M parameters: build each flavour once
        kw = dict(cases.CONTROLNET_KW, global_pool_conditions=gp)
        _nets[key] = (seeded_init(ControlNetModel(**kw).eval(), seed=11), seeded_init(ControlNetOracle(**kw).eval(), seed=11))
    ref, ora = _nets[key]
    inp = cases.controlnet_inputs() if rng.random() < 0.5 else cases.controlnet_inputs_nonsquare()
    kw = dict(conditioning_scale=rng.choice([1.0, 0.5, 2.0]), guess_mode=rng.random() < 0.4, skip_conv_in=rng.random() < 0.4,
              skip_time_emb=rng.random() < 0.3)
    rd, rm = ref(inp["sample"], inp["timestep"], encoder_hidden_states=inp["encoder_hidden_states"], controlnet_cond=inp["controlnet_cond"],
                 return_dict=False, **kw)
    od, om = ora(inp["sample"], inp["timestep"], inp["encoder_hidden_states"], inp["controlnet_cond"], **kw)
    errs = [rel(a, b) for a, b in zip(list(od) + [om], list(rd) + [rm])]
    assert all(a.shape == b.shape for a, b in zip(list(od) + [om], list(rd) + [rm])), "shapes differ"
    return "controlnet N=%d pool %d %s" % (inp["sample"].shape[0], gp, " ".join("%s=%s" % kv for kv in sorted(kw.items()))), max(errs)


def multi_case(rng, seed):
    """controlnet/multicontrolnet.py:45-99 (per-net lists, no summation) against MultiControlNetOracle: two nets, two images, two scales"""
    from controlnet.multicontrolnet import MultiControlNetModel
    from oracle.controlnet import MultiControlNetOracle
    for gp in (False, True):
        if gp not in _nets:
            controlnet_case(random.Random(seed), seed)      # builds the missing flavour as a side effect (cached)
            if gp not in _nets:
                from controlnet.controlnet import ControlNetModel
                from oracle.controlnet import ControlNetOracle
                kw = dict(cases.CONTROLNET_KW, global_pool_conditions=gp)
                _nets[gp] = (seeded_init(ControlNetModel(**kw).eval(), seed=11), seeded_init(ControlNetOracle(**kw).eval(), seed=11))
    ref = MultiControlNetModel([_nets[False][0], _nets[True][0]])
    ora = MultiControlNetOracle([_nets[False][1], _nets[True][1]])
    inp = cases.controlnet_inputs()
    conds = [inp["controlnet_cond"], seeded_tensor(tuple(inp["controlnet_cond"].shape), seed + 5).abs()]
    scales = [rng.choice([1.0, 0.5]), rng.choice([1.0, 2.0])]
    skip = rng.random() < 0.5
    rd, rm = ref(inp["sample"], inp["timestep"], inp["encoder_hidden_states"], conds, scales, return_dict=False, skip_conv_in=skip)
    od, om = ora(inp["sample"], inp["timestep"], inp["encoder_hidden_states"], conds, scales, skip_conv_in=skip)
    assert len(rd) == len(od) == 2 and len(rm) == len(om) == 2
    errs = [rel(a, b) for k in range(2) for a, b in zip(list(od[k]) + [om[k]], list(rd[k]) + [rm[k]])]
    return "multi-controlnet 2 nets scales %s skip_conv_in %d" % (scales, skip), max(errs)


def router_case(rng, seed):
    """model/ctrl_router.py:49-112 against RouterOracle: expert count, router type, number of routers, random sparse masks"""
    from model.ctrl_router import ControlNetRouter
    from oracle.router import RouterOracle
    E, rt, nr = rng.choice([2, 3, 5]), rng.choice(["simple_weights", "equal_weights"]), rng.choice([9, 12])
    ref = seeded_init(ControlNetRouter(num_experts=E, router_type=rt, num_routers=nr).eval(), seed=seed)
    ora = seeded_init(RouterOracle(num_experts=E, router_type=rt, num_routers=nr).eval(), seed=seed)
    assert sorted(ref.state_dict()) == sorted(ora.state_dict()), "router state-dict keys differ"
    worst = 0.0
    for _ in range(4):
        mask = [rng.randrange(2) for _ in range(E)]
        if not any(mask):
            mask[rng.randrange(E)] = 1
        mask = rng.choice([mask, None])
        rd, rm = ref(sparse_mask=mask)
        od, om = ora(sparse_mask=mask)
        assert rd.shape == od.shape == (nr, E) and rm.shape == om.shape == (E,)
        worst = max(worst, (rd - od).abs().max().item(), (rm - om).abs().max().item())
    return "router E=%d %s routers %d, 4 masks" % (E, rt, nr), worst


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    torch.set_grad_enabled(False)
    worst = 0.0
    for i in range(n):
        seed = first + 17 * i
        rng = random.Random(seed)
        desc, err = (multi_case if i % 8 == 7 else controlnet_case if i % 4 == 3 else router_case if i % 4 == 1 else adapter_case)(rng, seed)
        worst = max(worst, err)
        print("%-120s rel_inf %.2e%s" % (desc, err, "" if err <= TOL else "   MISMATCH"), flush=True)
        if not err <= TOL:
            raise SystemExit(1)
    print("LIVE CHECK OK: %d cases, worst %.2e  (blocks: %s)" % (n, worst, BLOCKS))


if __name__ == "__main__":
    main()
