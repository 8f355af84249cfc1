"""The standalone oracle restatements (oracle/*.py) against the golden vectors produced by the reference's OWN
files (tests/golden/make_golden.py) -- over a real `diffusers` or over the shim oracle/_shim: every golden file says which
(`provenance`), test_golden_provenance prints it and `pytest --require-real-diffusers` fails on the shim
(tests/golden/README.md).  CPU only."""
import os

import pytest
import torch

from helpers import load_golden, check_digest
import cases
from oracle.init import seeded_init, seeded_tensor
from oracle.controlnet import ControlNetOracle
from oracle.adapter import ControlNetAdapterOracle
from oracle.router import RouterOracle, merge_inference, merge_training

TOL = 2e-5   # fp32 vs fp32, different op grouping only

GOLDEN_FILES = ("controlnet_sd15.pt", "adapter_sdxl.pt", "adapter_video.pt", "adapter_variants.pt", "adapter_per_clip_context.pt", "router.pt")


def test_golden_provenance(request):
    """every golden file records what made it: the reference's files over `diffusers==<version>` or over the shim whose blocks are
    oracle/blocks.py (then the diffusers arithmetic is pinned by restatement only: "parity partially pinned")"""
    kinds = set()
    for f in GOLDEN_FILES:
        g = load_golden(f)
        p = g.get("provenance") or g.get("__provenance__")
        assert p and p.get("blocks") and p.get("generator") == "tests/golden/make_golden.py", "%s carries no provenance" % f
        assert p["blocks"].startswith("diffusers==") or p["blocks"].startswith("oracle-shim"), p
        print("GOLDEN %-32s blocks: %s (torch %s)" % (f, p["blocks"], p.get("torch")))
        kinds.add(p["blocks"])
    assert len(kinds) == 1, "golden files of mixed provenance: %s" % sorted(kinds)
    if request.config.getoption("--require-real-diffusers"):
        assert next(iter(kinds)).startswith("diffusers=="), "goldens were made over the shim, not a real diffusers: %s" % sorted(kinds)


@pytest.fixture(scope="module")
def controlnet():
    torch.set_grad_enabled(False)
    return seeded_init(ControlNetOracle(**cases.CONTROLNET_KW).eval(), seed=11)


def test_controlnet_structure(controlnet):
    g = load_golden("controlnet_sd15.pt")
    assert sum(p.numel() for p in controlnet.parameters()) == g["n_params"] == 361279120
    assert sorted(controlnet.state_dict().keys()) == g["keys"]


@pytest.mark.parametrize("tag,kw", [("plain", {}), ("scale0.5", dict(conditioning_scale=0.5)),
                                    ("skip_conv_in", dict(skip_conv_in=True)), ("skip_time_emb", dict(skip_time_emb=True)),
                                    ("guess", dict(guess_mode=True))])
def test_controlnet_golden(controlnet, tag, kw):
    g = load_golden("controlnet_sd15.pt")["runs"][tag]
    inp = cases.controlnet_inputs()
    down, mid = controlnet(inp["sample"], inp["timestep"], inp["encoder_hidden_states"], inp["controlnet_cond"], **kw)
    for i, (t, d) in enumerate(zip(list(down) + [mid], g)):
        check_digest(t, d, TOL, "controlnet[%s] out %d" % (tag, i))


def test_adapter_sdxl_golden():
    torch.set_grad_enabled(False)
    g = load_golden("adapter_sdxl.pt")
    ad = seeded_init(ControlNetAdapterOracle(**cases.ADAPTER_SDXL).eval(), seed=22)
    assert sorted(ad.state_dict().keys()) == g["keys"]
    assert sum(p.numel() for p in ad.parameters()) == g["n_params"]
    downs, _ = cases.pyramid_inputs(N=2, h0=8, seed=200, with_mid=False)
    out, mid = ad(downs, num_frames=1, timestep=torch.tensor(749.0), encoder_hidden_states=seeded_tensor((2, 77, 2048), 290))
    assert mid is None
    for i, (t, d) in enumerate(zip(out, g["out"])):
        check_digest(t, d, TOL, "adapter_sdxl out %d" % i)
    for i in (9, 10, 11):
        assert out[i].abs().max().item() == 0.0      # zeros_like for slots without an adapter


def test_adapter_video_golden():
    torch.set_grad_enabled(False)
    g = load_golden("adapter_video.pt")
    ad = seeded_init(ControlNetAdapterOracle(**cases.ADAPTER_VIDEO).eval(), seed=33)
    assert sorted(ad.state_dict().keys()) == g["keys"]
    assert sum(p.numel() for p in ad.parameters()) == g["n_params"]
    downs, midin = cases.pyramid_inputs(N=8, h0=8, seed=300, with_mid=True)
    out, mid = ad(downs, mid_block_res_sample=midin, num_frames=4, timestep=torch.tensor(961.0),
                  encoder_hidden_states=seeded_tensor((1, 1, 1024), 390))
    for i, (t, d) in enumerate(zip(list(out) + [mid], g["out"])):
        check_digest(t, d, TOL, "adapter_video out %d" % i)


def test_router_golden():
    torch.set_grad_enabled(False)
    g = load_golden("router.pt")
    r = seeded_init(RouterOracle(num_experts=3, router_type="simple_weights", num_routers=12).eval(), seed=44)
    assert sorted(r.state_dict().keys()) == g["keys"]
    for tag, mask in {"all": [1, 1, 1], "m101": [1, 0, 1], "none": None}.items():
        dw, mw = r(sparse_mask=mask)
        assert torch.allclose(dw, g["runs"][tag]["down"], atol=1e-7)
        assert torch.allclose(mw, g["runs"][tag]["mid"], atol=1e-7)
    dw, mw = RouterOracle(num_experts=2, router_type="equal_weights")(sparse_mask=[1, 1])
    assert torch.allclose(dw, g["equal"]["down"]) and torch.allclose(mw, g["equal"]["mid"])


def test_merge_semantics():
    """quirk N6: the inference merge weights every active expert by w[k][0]."""
    E, F = 3, 4
    downs = [[seeded_tensor((2, 8, 2, 2), 10 * e + r) for r in range(12)] for e in range(E)]
    mids = [seeded_tensor((2, 8, 1, 1), 500 + e) for e in range(E)]
    dw = torch.softmax(seeded_tensor((12, E), 1, fp16_round=False), -1)
    mw = torch.softmax(seeded_tensor((E,), 2, fp16_round=False), -1)
    md, mm = merge_inference(downs, mids, dw, mw, [1, 1, 1], F)
    for r in range(12):
        assert torch.allclose(md[r], dw[r][0] * sum(downs[e][r] for e in range(E)), atol=1e-6)
    assert torch.allclose(mm, mw[0] * sum(mids), atol=1e-6)
    td, tm = merge_training(downs, mids, dw, mw, [1, 1, 1])
    assert torch.allclose(td[3], sum(dw[3][e] * downs[e][3] for e in range(E)), atol=1e-6)


@pytest.mark.parametrize("tag", sorted(cases.ADAPTER_VARIANTS))
def test_adapter_variant_golden(tag):
    """configurations outside the shipped YAMLs: 2 adapters per location / ResNet-only, num_blocks = 2, temporal-only"""
    torch.set_grad_enabled(False)
    g = load_golden("adapter_variants.pt")[tag]
    cfg, io, downs, mid, ehs = cases.variant_inputs(tag)
    ad = seeded_init(ControlNetAdapterOracle(**cfg).eval(), seed=77)
    assert sorted(ad.state_dict().keys()) == g["keys"] and sum(p.numel() for p in ad.parameters()) == g["n_params"]
    out, m = ad(downs, mid_block_res_sample=mid, num_frames=io["frames"], timestep=cases.variant_timestep(io), encoder_hidden_states=ehs)
    for i, (t, d) in enumerate(zip(list(out) + ([m] if m is not None else []), g["out"])):
        check_digest(t, d, TOL, "%s out %d" % (tag, i))


@pytest.mark.parametrize("tag", sorted(cases.PER_CLIP_CONTEXT))
def test_adapter_per_clip_context_golden(tag):
    """per-frame encoder states with several clips: the time context reaches the temporal transformer ordered (pixel, clip)
    while its rows are (clip, pixel) (model/adapter_spatial_temporal.py:246-249) -- the oracle's restatement against goldens
    made by the reference's own file; and the pairing is NOT the natural one (each clip run alone differs by > 1e-1)"""
    torch.set_grad_enabled(False)
    g = load_golden("adapter_per_clip_context.pt")[tag]
    cfg, clips, frames, downs, ehs, ts = cases.per_clip_context_inputs(tag)
    ad = seeded_init(ControlNetAdapterOracle(**cfg).eval(), seed=36)
    assert sorted(ad.state_dict().keys()) == g["keys"]
    out, m = ad(downs, num_frames=frames, timestep=ts, encoder_hidden_states=ehs)
    assert m is None
    for i, (t, d) in enumerate(zip(out, g["out"])):
        check_digest(t, d, TOL, "%s out %d" % (tag, i))
    alone = ad([d[:frames] for d in downs], num_frames=frames, timestep=ts[:frames], encoder_hidden_states=ehs[:frames])[0][0]
    full0 = g["out"][0]["full"][:frames]
    assert ((alone - full0).abs().max() / full0.abs().max()).item() > 1e-1 or clips == 1


def test_controlnet_nonsquare_single_image_golden(controlnet):
    g = load_golden("controlnet_sd15.pt")["nonsquare_n1"]
    inp = cases.controlnet_inputs_nonsquare()
    down, mid = controlnet(inp["sample"], inp["timestep"], inp["encoder_hidden_states"], inp["controlnet_cond"])
    for i, (t, d) in enumerate(zip(list(down) + [mid], g)):
        check_digest(t, d, TOL, "controlnet[nonsquare] out %d" % i)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference checkout (build container only)")
def test_oracle_against_the_reference_files_live_random_configurations():
    """beyond the fixed goldens: randomly drawn adapter / ControlNet configurations, the reference's own files (over the diffusers
    shim) against the oracle on the same seeded weights and inputs -- tests/golden/live_check.py, in its own process because it
    installs the shim as `diffusers`"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "live_check.py"), "8", "3000"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "LIVE CHECK OK: 8 cases" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
