"""Benchmark of the Ctrl-Adapter denoising hot path on MI355X (BASELINE.json metric).

One "step" = what the reference's pipelines do per denoising step around the UNet call
(sdxl/pipelines/sdxl_controlnet_adapter_pipeline.py:1306-1343; svd/pipelines/...:684-747; i2vgen_xl/pipelines/...:957-1082):
pool the latents to 64x64, ControlNetModel.forward (K of them + router + merge in the multi-condition config),
ControlNetAdapter.forward -- on synthetic latents / prompts / condition images already resident in HBM and seeded random
weights of the real architectures (361 M + 184 M / 587 M parameters).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload sdxl|svd16|i2vgen16|multi3]

`--gpus N` with N > 1 launches the N ranks itself (re-executes under `python -m torch.distributed.run`, one rank per
GPU over RCCL, 127.0.0.1 rendezvous); started under a launcher already (RANK / WORLD_SIZE in the environment) it joins it.

Rank 0 prints the headline as the LAST stdout line, ONE JSON object of about 3 KB (the driver keeps an 8 KB tail of stdout).  Besides the driver contract it carries:
  roofline      ONE kernel (symbol with template arguments + shape): algorithmic FLOPs per launch / HIP-event time per
                launch on the launch stream, vs the dense fp16 MFMA peak; `traffic` = PMC HBM bytes per launch of that
                kernel from the committed rocprofv3 passes (profiles/, tools/profile_round.sh)
  cpu_baseline  the pure-PyTorch fp32 oracle on the host cores: ONE pass over the whole step of this workload (the same
                inputs the HIP step was timed on); its outputs are kept and compared with the HIP step's:
  parity_at_bench_config   worst rel-inf over the 13 ControlNet + 12 (+ mid) adapter tensors, HIP step vs oracle
The per-kernel table (every template instantiation x shape of the step, sorted by time) and the per-class table go to a
FILE (`--per-kernel-out`, default bench_per_kernel.json next to this script), never to stdout.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]

MFMA_PEAK_TFLOPS = 2500.0      # dense fp16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md chip table
# what back-to-back v_mfma_f32_32x32x16_f16 sustain on pseudo-random fp16 operands (0.708 of the nominal rate: the clock the power
# management holds depends on the data; tools/ubench/attn_mix.hip, profiles/r03_mfma_data_dependence.txt) -- reported BESIDE
# `peak`, never instead of it
MFMA_SUSTAINED_TFLOPS = 1770.0
HBM_PEAK_GBS = 8000.0

SDXL_ADAPTER = dict(backbone_model_name="sdxl", num_blocks=1, num_frames=1, num_adapters_per_location=3,
                    cross_attention_dim=2048, add_spatial_resnet=True, add_temporal_resnet=False,
                    add_spatial_transformer=True, add_temporal_transformer=False,
                    add_adapter_location_A=True, add_adapter_location_B=True, add_adapter_location_C=True)
VIDEO_ADAPTER = dict(backbone_model_name="svd", num_blocks=1, num_frames=16, num_adapters_per_location=3,
                     cross_attention_dim=1024, add_spatial_resnet=True, add_temporal_resnet=True,
                     add_spatial_transformer=True, add_temporal_transformer=True,
                     add_adapter_location_A=True, add_adapter_location_B=True, add_adapter_location_C=True,
                     add_adapter_location_D=True, add_adapter_location_M=True)

# BASELINE.json configs -> workloads.  flops: SURVEY.md section 8d (ControlNet 0.2835 TFLOP / frame, SDXL adapter 2.258 /
# image, video adapter 0.659 / frame)
WORKLOADS = {
    "sdxl": dict(config=2, video=False, n_cn=1, skip_conv_in=False, adapter=SDXL_ADAPTER,
                 metric="denoise-steps/s (ControlNet+adapter fwd) SDXL 1024^2 b=8",
                 what="SDXL depth 1024^2 batch=%d per GPU (N=%d images enter ControlNet+adapter, no CFG doubling); "
                      "pool->ControlNet(SD1.5, 64x64 latents, 512^2 cond)->Ctrl-Adapter(A,B,C x3, up x2)"),
    "svd16": dict(config=3, video=True, n_cn=1, skip_conv_in=True, adapter=VIDEO_ADAPTER,
                  metric="denoise-steps/s (ControlNet+adapter fwd) SVD 16-frame clip (CFG pair, 32 frames)",
                  what="SVD depth 576x1024, 16 frames, CFG pair (N=32 frames enter ControlNet+adapter), skip_conv_in; "
                       "ControlNet(64x64 latents)->Ctrl-Adapter(A-D+M, spatial+temporal ResNets and transformers)"),
    "i2vgen16": dict(config=4, video=True, n_cn=1, skip_conv_in=False, adapter=dict(VIDEO_ADAPTER, backbone_model_name="i2vgen-xl"),
                     metric="denoise-steps/s (ControlNet+adapter fwd) I2VGen-XL 16-frame clip per GPU (CFG pair, 32 frames)",
                     what="I2VGen-XL depth, one 16-frame clip per GPU (CFG pair, N=32 frames; 8 clips on 8 GPUs = BASELINE config 4), "
                          "whole clips sharded across ranks: no data-path collective"),
    "multi3": dict(config=5, video=True, n_cn=3, skip_conv_in=False, adapter=dict(VIDEO_ADAPTER, backbone_model_name="i2vgen-xl"),
                   metric="denoise-steps/s (3 ControlNets + router + merge + adapter fwd) I2VGen-XL 16-frame clip per GPU",
                   what="I2VGen-XL multi-condition (depth+canny+softedge): K=3 ControlNets -> ctrl_router weights -> expert merge -> "
                        "Ctrl-Adapter, one 16-frame clip per GPU (CFG pair, N=32 frames)"),
}


def step_flops(w, n):
    if not w["video"]:
        return (0.2835 + 2.258) * 1e12 * n
    return (w["n_cn"] * 0.2835 + 0.659) * 1e12 * n


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one per GPU) and relay their output."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this host driver (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def build_models(dev, w):
    import ctrl_adapter_amd as P
    from ctrl_adapter_amd.synthetic import seeded_init      # seeded random weights (no checkpoints offline)
    cns = [seeded_init(P.ControlNetModel(cross_attention_dim=768), seed=11 + 100 * k).to(dev) for k in range(w["n_cn"])]
    ad = seeded_init(P.ControlNetAdapter(**w["adapter"]), seed=22).to(dev)
    router = None
    if w["n_cn"] > 1:
        router = seeded_init(P.ControlNetRouter(num_experts=w["n_cn"], router_type="simple_weights", num_routers=12), seed=44).to(dev)
    return P, cns, ad, router


def make_inputs(dev, w, n, seed):
    import torch
    g = torch.Generator().manual_seed(seed)
    if not w["video"]:
        lat = torch.randn(n, 4, 128, 128, generator=g)
        ehs_a = torch.randn(n, 77, 2048, generator=g)
    else:
        lat = torch.randn(n, 4, 64, 64, generator=g)
        ehs_a = torch.randn(1, 1, 1024, generator=g)
    d = dict(latents=lat, ehs_c=torch.randn(n, 77, 768, generator=g), ehs_a=ehs_a)
    for k in range(w["n_cn"]):
        d["cond%d" % k] = torch.rand(n, 3, 512, 512, generator=g)
    return {k: v.half().to(dev) for k, v in d.items()}


def quick_time(dev, wname, batch, steps, warmup, world):
    """one more BASELINE workload timed the same way as the headline (graph-captured step, the pipelines' two-call form, barrier +
    sync on both sides of exactly `steps` steps) -- no roofline / CPU leg: their parity at these shapes is held by the GPU test
    suite (tests/test_gpu_e2e.py: ..._at_benched_shapes_vs_oracle)"""
    import torch
    import ctrl_adapter_amd.dp as dp
    w = WORKLOADS[wname]
    n = batch if not w["video"] else 32
    nf = 1 if not w["video"] else 16
    P, cns, ad, router = build_models(dev, w)
    x = make_inputs(dev, w, n, seed=4321)
    t = torch.tensor([499.0], device=dev)
    masks = [1] * w["n_cn"]
    multi = P.MultiControlNetModel(cns) if w["n_cn"] > 1 else None

    def step():
        s = P.pool_latents(x["latents"], (64, 64))
        if w["n_cn"] == 1:
            down, mid = cns[0](s, t, x["ehs_c"], x["cond0"], conditioning_scale=1.0, return_dict=False, skip_conv_in=w["skip_conv_in"])
        else:
            # MultiControlNetModel.forward (controlnet/multicontrolnet.py:45-99): per-net lists
            downs, mids = multi(s, t, x["ehs_c"], [x["cond%d" % k] for k in range(len(cns))], [1.0] * len(cns), return_dict=False,
                                skip_conv_in=w["skip_conv_in"])
            dw, mw = router(sparse_mask=masks)
            down, mid = router.merge(downs, mids, dw, mw, masks, num_frames=nf, inference_quirk=True)
        return ad(down, mid_block_res_sample=mid, num_frames=nf, timestep=t, encoder_hidden_states=x["ehs_a"])

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        keep = step()      # noqa: F841
    for _ in range(warmup):
        graph.replay()
    el = dp.timed_region(graph.replay, steps, device=dev)
    ms = el / steps * 1e3
    fl = step_flops(w, n)
    res = {"value": round(dp.aggregate_throughput(1, steps, el, world), 3), "ms_per_step": round(ms, 3),      # (unit: the headline's)
           "mfma_frac": round(fl / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4), "baseline_config": w["config"], "batch_per_gpu": n}      # (20 steps each)
    del graph, keep, cns, ad, router, x
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return res


def pmc_traffic_for(kernel_row):
    """HBM bytes per launch of one (symbol, grid) from the committed rocprofv3 PMC passes (cannot run inside the bench)"""
    tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic*.json")))
    for tfile in reversed(tfiles):
        try:
            with open(tfile) as fh:
                doc = json.load(fh)
        except Exception:
            continue
        ent = (doc.get("kernels_by_shape") or {}).get((kernel_row["symbol"] + " " + kernel_row["shape"]).strip()) or \
            doc.get("kernels", {}).get("%s|%d" % (kernel_row["symbol"], kernel_row["grid"]))
        if ent:
            return ent.get("hbm_bytes_per_launch_corrected"), os.path.basename(tfile)
    return None, None


def pmc_total_for(workload):
    """whole-step HBM bytes (corrected FETCH_SIZE x 2 + WRITE_SIZE) from the newest committed PMC file of this workload (the sdxl
    headline: profiles/rNN_pmc_hbm_traffic_vK.json; counters cannot be collected inside the bench)"""
    if workload != "sdxl":
        return None, None
    for tfile in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic_v*.json")))):
        try:
            with open(tfile) as fh:
                tot = json.load(fh).get("total_hbm_bytes_per_step_corrected")
        except Exception:
            continue
        if tot:
            return round(tot / 1e9, 2), os.path.basename(tfile)
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="sdxl: images per GPU entering the hot path (BASELINE: 8)")
    ap.add_argument("--workload", default="sdxl", choices=sorted(WORKLOADS))
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of a captured hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the svd16 / i2vgen16 / multi3 / sdxl b=16 legs the default invocation times after the headline")
    ap.add_argument("--per-kernel-out", default=os.path.join(ROOT, "bench_per_kernel.json"),
                    help="where rank 0 writes the per-kernel / per-class tables (they are too long for the headline line)")
    ap.add_argument("--clip-split", action="store_true",
                    help="video workloads: ONE clip for the whole job, its 16 frames sharded over the --gpus ranks (frame-mixing "
                         "ops exchange over RCCL: all_to_all around the temporal transformers, Conv3d halo, GroupNorm "
                         "all-reduce; SURVEY.md 8e row 2); strong scaling, eager launches (the exchanges are host callbacks)")
    ap.add_argument("--clip-lanes", type=int, default=4, help="--clip-split, native transport: communicators = stream lanes of the adapter (1 = one stream)")
    ap.add_argument("--clip-transport", default="rccl", choices=["rccl", "torch"],
                    help="--clip-split: native RCCL enqueued from C++ (graph-capturable, default) or torch.distributed callbacks (eager)")
    ap.add_argument("--profile-scope", default="step", choices=["step", "controlnet", "adapter"],
                    help="what the per-kernel (HIP-event) leg runs: the whole step, the controlnet(...) call(s) alone, or the adapter(...) call alone")
    ap.add_argument("--fused", action="store_true",
                    help="time the fused controlled_step(controlnet, adapter, ...) instead of the pipelines' two calls "
                         "controlnet(...) ; adapter(...) (same arithmetic, bit-identical results)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("CTRL_BENCH_MARKER_DIR"):      # tests/test_dist.py: proof that this rank was started (written before anything can fail)
        with open(os.path.join(os.environ["CTRL_BENCH_MARKER_DIR"], "rank%d_of_%d" % (rank, world)), "w") as fh:
            fh.write(os.environ.get("MASTER_ADDR", "") + "\n")
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP hot path has no CPU fallback")
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))
    if local >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import ctrl_adapter_amd.dp as dp
    dp.init("nccl")
    torch.set_grad_enabled(False)

    w = WORKLOADS[args.workload]
    n = args.batch if not w["video"] else 32          # video: one CFG pair of a 16-frame clip per GPU
    nf = 1 if not w["video"] else 16
    P, cns, ad, router = build_models(dev, w)
    comm = None
    if args.clip_split:
        if not w["video"] or w["n_cn"] != 1 or nf % world:
            raise SystemExit("--clip-split: a single-ControlNet video workload whose 16 frames divide by --gpus")
        from ctrl_adapter_amd.clip_parallel import RcclTransport, TorchDistTransport, shard_frames
        if args.clip_transport == "torch":       # round 3's Python-callback transport (eager launches only)
            if world == 1 and not torch.distributed.is_initialized():      # a one-rank group still runs every exchange through RCCL
                s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port_ = s_.getsockname()[1]; s_.close()
                torch.distributed.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port_, rank=0, world_size=1)
            comm = TorchDistTransport()
            args.no_graph = True
        else:                                    # native RCCL on the forward's stream (csrc/clip_rccl.cpp): hipGraph-capturable
            comm = RcclTransport(rank=rank, world=world, lanes=args.clip_lanes) if world == 1 else RcclTransport(lanes=args.clip_lanes)
            if args.clip_lanes > 1:
                args.no_graph = True             # several communicators inside one capture: RCCL 2.26 refuses (hipErrorStreamCaptureUnsupported);
                                                 # four eager lanes (59.7 ms at world 1) beat one captured lane (62.4 ms) -- profiles/r04_clip_split_world1.txt
        x = make_inputs(dev, w, n, seed=1234)          # the SAME clip on every rank ...
        x = {k: (shard_frames(v, nf, rank, world) if v.shape[0] == n else v) for k, v in x.items()}      # ... its frames sharded
        n, nf = n // world, nf // world
    else:
        x = make_inputs(dev, w, n, seed=1234 + rank)   # every rank owns different images / clips (no collective)
    t = torch.tensor([499.0], device=dev)
    masks = [1] * w["n_cn"]
    multi = P.MultiControlNetModel(cns) if w["n_cn"] > 1 else None

    def controlnets(s):
        if w["n_cn"] == 1:
            return cns[0](s, t, x["ehs_c"], x["cond0"], conditioning_scale=1.0, return_dict=False, skip_conv_in=w["skip_conv_in"])
        # MultiControlNetModel.forward (controlnet/multicontrolnet.py:45-99): per-net lists (the mirror runs the nets on stream lanes)
        downs, mids = multi(s, t, x["ehs_c"], [x["cond%d" % k] for k in range(len(cns))], [1.0] * len(cns), return_dict=False,
                            skip_conv_in=w["skip_conv_in"])
        dw, mw = router(sparse_mask=masks)            # model/ctrl_router.py:85-112, then the pipeline's merge (:1000-1022)
        return router.merge(downs, mids, dw, mw, masks, num_frames=nf, inference_quirk=True)

    def step_separate():
        s = P.pool_latents(x["latents"], (64, 64))
        down, mid = controlnets(s)
        if comm is not None:
            return (down, mid), ad(down, mid_block_res_sample=mid, num_frames=nf, timestep=t, encoder_hidden_states=x["ehs_a"], clip_comm=comm)
        return (down, mid), ad(down, mid_block_res_sample=mid, num_frames=nf, timestep=t, encoder_hidden_states=x["ehs_a"])

    def step_fused():
        s = P.pool_latents(x["latents"], (64, 64))
        return P.controlled_step(cns[0], ad, s, t, x["ehs_c"], x["cond0"], 1.0, skip_conv_in=w["skip_conv_in"], num_frames=nf,
                                 adapter_encoder_hidden_states=x["ehs_a"])

    if args.fused and w["n_cn"] != 1:
        raise SystemExit("--fused covers one ControlNet")
    step = step_fused if args.fused else step_separate

    for _ in range(2):          # eager warm-up: builds the plans (weight packing) and sizes the workspaces
        step()
    torch.cuda.synchronize()

    run, mode = step, "eager"
    if not args.no_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = step()      # noqa: F841  (keeps the captured outputs alive)
            run, mode = graph.replay, "hipgraph"
        except Exception as e:   # capture is an optimisation only
            print("bench: hipGraph capture failed (%s); running eager" % str(e).split("\n")[0], file=sys.stderr)
            torch.cuda.synchronize()
            run = step

    for _ in range(args.warmup):
        run()

    elapsed = dp.timed_region(run, args.steps, device=dev)        # barrier + sync on both sides, MAX over ranks
    ms_per_step = elapsed / args.steps * 1e3
    value = dp.aggregate_throughput(1, args.steps, elapsed, world)   # whole-job denoise-steps/s (each rank: its own batch / clip)
    if comm is not None:
        value = args.steps / elapsed                                 # ONE clip for the whole job: strong scaling
    # the median over single steps ("median of >= 20"), taken AFTER the timed region with an event pair per step on
    # the launch stream; reported beside the contract's mean-of-K `ms_per_step`, never instead of it
    median_ms = None
    if rank == 0 and comm is None:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(args.steps, 20))]
        for e0, e1 in evs:
            e0.record()
            run()
            e1.record()
        torch.cuda.synchronize()
        ts = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
        median_ms = round(ts[len(ts) // 2], 3)

    # what the LAST graph replay wrote (the timed artifact itself: replay number warmup + steps + len(evs)), copied out before
    # anything else runs on these plans -- compared with the oracle in the cpu_baseline leg below
    replay_out, replays = None, 0
    if rank == 0 and mode == "hipgraph" and comm is None and not args.fused:
        replays = args.warmup + args.steps + max(args.steps, 20)
        (gd, gm), (ga, gam) = static_out
        torch.cuda.synchronize()
        replay_out = [v.float().cpu() for v in gd] + [gm.float().cpu()] + [v.float().cpu() for v in ga] + ([gam.float().cpu()] if gam is not None else [])

    # ---- the same step through the fused entry point (ControlNet on its own stream, adapter blocks start when their input
    #      exists): same arithmetic, bit-identical results; reported beside the headline, never as `value` ----
    fused = None
    if w["n_cn"] == 1 and not args.fused and not args.no_graph and comm is None:
        try:
            for _ in range(2):
                step_fused()
            torch.cuda.synchronize()
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2):
                static_out2 = step_fused()      # noqa: F841
            for _ in range(args.warmup):
                g2.replay()
            el2 = dp.timed_region(g2.replay, args.steps, device=dev)
            fused = {"ms_per_step": round(el2 / args.steps * 1e3, 3),
                     "value": round(dp.aggregate_throughput(1, args.steps, el2, world), 3)}
        except Exception as e:
            print("bench: fused leg skipped (%s)" % str(e).split("\n")[0], file=sys.stderr)
            torch.cuda.synchronize()

    # ---- roofline leg: eager steps with HIP events around every launch, on the launch stream (lanes off) ----
    kernels, per_kernel, roof = {}, [], None
    below_quarter_ms, below_quarter_launches = 0.0, 0.0
    if rank == 0:
        from ctrl_adapter_amd import ops
        reps = 3
        if args.profile_scope == "step":
            prof_fn = step
        else:
            s_ = P.pool_latents(x["latents"], (64, 64))
            down_, mid_ = controlnets(s_)
            torch.cuda.synchronize()
            if args.profile_scope == "controlnet":
                prof_fn = lambda: controlnets(s_)      # noqa: E731
            else:
                kw_ = dict(clip_comm=comm) if comm is not None else {}
                prof_fn = lambda: ad(down_, mid_block_res_sample=mid_, num_frames=nf, timestep=t, encoder_hidden_states=x["ehs_a"], **kw_)      # noqa: E731
        with ops.Profiler() as prof:
            for _ in range(reps):
                prof_fn()
        for k, v in prof.rows.items():
            ms = v[0] / reps
            kernels[k] = {"ms_per_step": round(ms, 4), "launches_per_step": v[1] // reps,
                          "tflops": round(v[2] / reps / (ms * 1e-3) / 1e12, 1) if v[2] else None,
                          "gbs": round(v[3] / reps / (ms * 1e-3) / 1e9, 1) if v[3] else None}
        rows = prof.per_kernel(reps)
        order = sorted(rows, key=lambda k: -rows[k]["ms_per_step"])
        for k in order:
            r = rows[k]
            # the roofline that BINDS the launch: algorithmic FLOPs / time vs the dense MFMA peak, algorithmic bytes / time vs the HBM
            # peak, whichever fraction is larger (a K = 144 convolution on the matrix cores is an HBM kernel; a streaming kernel has no FLOPs)
            fm = (r["tflops"] or 0.0) / MFMA_PEAK_TFLOPS
            fh = (r["gbs"] or 0.0) / HBM_PEAK_GBS
            if fm > 0.0 and fm >= fh:
                nearest, ach, peak = "mfma", r["tflops"], MFMA_PEAK_TFLOPS
            elif fh > 0.0:
                nearest, ach, peak = "hbm", r["gbs"], HBM_PEAK_GBS
            else:
                nearest, ach, peak = None, None, None
            # a launch below a quarter of BOTH roofs is bound by neither (latency / epilogue / occupancy): it is labelled "none" and
            # its time is summed into the headline's `ms_below_quarter_roof`
            bound = nearest if max(fm, fh) >= 0.25 else ("none" if nearest else None)
            if bound == "none":
                below_quarter_ms += r["ms_per_step"]
                below_quarter_launches += r["launches_per_step"]
            per_kernel.append({"kernel": k, "class": r["tag"], "launches_per_step": round(r["launches_per_step"], 2),
                               "ms_per_step": round(r["ms_per_step"], 4), "avg_launch_ms": round(r["avg_launch_ms"], 5),
                               "bound": bound, "nearest_roof": nearest, "achieved": round(ach, 1) if ach else None,
                               "frac": round(ach / peak, 4) if ach else None, "frac_mfma": round(fm, 4), "frac_hbm": round(fh, 4),
                               "grid": r["grid"]})
        dom = next((k for k in order if rows[k]["tflops"] or rows[k]["gbs"]), None)
        if dom is not None:
            r = rows[dom]
            traffic, tsrc = pmc_traffic_for(r)
            if r["tflops"]:
                roof = {"kernel": dom, "bound": "mfma", "achieved": round(r["tflops"], 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(r["tflops"] / MFMA_PEAK_TFLOPS, 4), "sustained_peak": MFMA_SUSTAINED_TFLOPS,
                        "frac_of_sustained": round(r["tflops"] / MFMA_SUSTAINED_TFLOPS, 4), "traffic": traffic, "traffic_source": tsrc,
                        "flops_per_launch": r["flops_per_launch"], "algorithmic_bytes_per_launch": r["bytes_per_launch"],
                        "avg_launch_ms": r["avg_launch_ms"], "launches_per_step": r["launches_per_step"],
                        "ms_per_step": round(r["ms_per_step"], 4)}
            else:
                roof = {"kernel": dom, "bound": "hbm", "achieved": round(r["gbs"], 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(r["gbs"] / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": tsrc,
                        "bytes_per_launch": r["bytes_per_launch"], "avg_launch_ms": r["avg_launch_ms"],
                        "launches_per_step": r["launches_per_step"], "ms_per_step": round(r["ms_per_step"], 4)}

    # ---- cpu_baseline leg: the fp32 oracle on the host cores, ONE pass over the whole step of this workload (rank 0, N=1
    #      only) on the very inputs the HIP step was timed on; its outputs are the parity check of the benched configuration ----
    cpu, parity = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and comm is None:
        import cases  # noqa: F401
        from oracle.init import seeded_init
        from oracle.controlnet import ControlNetOracle
        from oracle.adapter import ControlNetAdapterOracle
        from oracle.router import RouterOracle, merge_inference
        (hd, hm), (ha, ham) = step()                 # the HIP step, eager, same plans / inputs as the timed graph
        torch.cuda.synchronize()
        hip_out = [("controlnet.down%d" % i, v.float().cpu()) for i, v in enumerate(hd)] + [("controlnet.mid", hm.float().cpu())] + \
                  [("adapter.down%d" % i, v.float().cpu()) for i, v in enumerate(ha)] + \
                  ([("adapter.mid", ham.float().cpu())] if ham is not None else [])
        box_cores = os.cpu_count() or 1
        cores = min(box_cores, 32)                  # more threads than this slow the fp32 oracle down (NUMA / oversubscription)
        torch.set_num_threads(cores)
        ocs = [seeded_init(ControlNetOracle(cross_attention_dim=768).eval(), seed=11 + 100 * k) for k in range(w["n_cn"])]
        oa = seeded_init(ControlNetAdapterOracle(**w["adapter"]).eval(), seed=22)
        orr = seeded_init(RouterOracle(num_experts=w["n_cn"], router_type="simple_weights", num_routers=12).eval(), seed=44) if w["n_cn"] > 1 else None
        xc = {k: v.float().cpu() for k, v in x.items()}
        tc = torch.tensor(499.0)

        def cpu_step():
            s = torch.nn.functional.adaptive_avg_pool2d(xc["latents"], (64, 64))
            outs = [oc(s, tc, xc["ehs_c"], xc["cond%d" % k], skip_conv_in=w["skip_conv_in"]) for k, oc in enumerate(ocs)]
            if orr is None:
                d, m = outs[0]
            else:
                dw, mw = orr(sparse_mask=masks)
                d, m = merge_inference([o[0] for o in outs], [o[1] for o in outs], dw, mw, masks, nf)
            return (d, m), oa(d, mid_block_res_sample=m if w["video"] else None, num_frames=nf, timestep=tc, encoder_hidden_states=xc["ehs_a"])
        c0 = time.perf_counter()
        (od, om), (oad, oam) = cpu_step()
        per_step = time.perf_counter() - c0
        cpu = {"value": round(1.0 / per_step, 5), "unit": "denoise-steps/s", "cores": cores, "box_cores": box_cores, "kind": "port",
               "note": "oracle/ = fp32 restatement, live-checked bit-identical to the reference's files (tests/golden/live_check.py)",
               "sample": "one pass over the whole step of this workload (N=%d, the timed inputs), fp32 PyTorch oracle on %d "
                         "threads (the box has %d logical cores): %.2f s" % (n, cores, box_cores, per_step)}
        ref_out = list(od) + [om] + list(oad) + ([oam] if oam is not None else [])
        worst, worst_name, zeros_ok, ntens = 0.0, None, True, 0
        for (name, a), b in zip(hip_out, ref_out):
            ntens += 1
            den = b.abs().max().item()
            if den == 0.0:                           # non-selected adapter slots are zeros_like (model/ctrl_adapter.py:193)
                zeros_ok = zeros_ok and a.abs().max().item() == 0.0
                continue
            e = ((a - b).abs().max() / den).item()
            if not (e <= worst):                     # NaN-propagating maximum
                worst, worst_name = e, name
        eager = {"rel_inf_worst": float("%.3e" % worst), "tensor": worst_name, "zero_slots_exact": zeros_ok,
                 "what": "a separate EAGER step of the same plans / inputs vs the oracle"}

        def worst_vs_oracle(outs):
            wv, wn_, zok, nt = 0.0, None, True, 0
            for (name, _), a, b in zip(hip_out, outs, ref_out):
                nt += 1
                den = b.abs().max().item()
                if den == 0.0:                       # non-selected adapter slots are zeros_like (model/ctrl_adapter.py:193)
                    zok = zok and a.abs().max().item() == 0.0
                    continue
                e = ((a - b).abs().max() / den).item()
                if not (e <= wv):                    # NaN-propagating maximum
                    wv, wn_ = e, name
            return wv, wn_, zok, nt
        if replay_out is not None:
            # the headline's parity is that of the thing that was TIMED: the tensors the last hipGraph replay left behind
            wv, wn_, zok, nt = worst_vs_oracle(replay_out)
            same = all(torch.equal(a, b) for (_, a), b in zip(hip_out, replay_out))
            parity = {"rel_inf_worst": float("%.3e" % wv), "tensor": wn_, "tensors": nt, "zero_slots_exact": zok,
                      "bound": 1e-3, "ok": bool(wv <= 1e-3 and zok),
                      "what": "hipGraph replay #%d (the timed graph's own output tensors, %d distinct inputs) vs fp32 oracle -> oracle chain" % (replays, n),
                      "replay_bit_identical_to_eager": bool(same), "eager": eager}
        else:
            parity = dict(eager, tensors=ntens, bound=1e-3, ok=bool(worst <= 1e-3 and zeros_ok),
                          what="HIP step (eager; these plans, these %d distinct inputs) vs fp32 oracle -> oracle chain" % n)

    # ---- the other north-star workloads (BASELINE configs 3, 4, 5 and the CFG-doubled SDXL batch), after the headline's timed
    #      region, the default single-GPU invocation only (the N > 1 scaling runs stay short): `other_workloads` of the headline ----
    others = None
    if args.workload == "sdxl" and args.batch == 8 and world == 1 and comm is None and not args.no_graph and not args.fused and not args.no_other_workloads:
        others = {}
        for name, wn, b in (("svd16", "svd16", 0), ("i2vgen16", "i2vgen16", 0), ("multi3", "multi3", 0), ("sdxl_b16", "sdxl", 16),
                               ("sdxl_b2_cfg_pair", "sdxl", 2)):      # (the last one: BASELINE config 1's shape, one image + its CFG twin)
            try:
                others[name] = quick_time(dev, wn, b, 20, 3, world)
            except Exception as e:
                others[name] = {"error": str(e).split("\n")[0][:120]}
                torch.cuda.synchronize()
        # the headline once more under the CONSERVATIVE precision selection -- what plan creation picks by itself for a checkpoint whose
        # normalisation scales have outlier channels (module.selection): split operands on every ResNet convolution of the ControlNet's
        # levels 0-2, fp32 adapter token stream.  The price of that fallback, on record beside the number it protects.
        from ctrl_adapter_amd import ops as _ops
        prev = {k: _ops.set_policy(k, v) for k, v in (("CTRL_CN_SPLIT_RESNET_LEVELS", "3"), ("CTRL_ADAPTER_TOK_F16", "0"))}
        try:
            others["sdxl_b8_conservative_selection"] = quick_time(dev, "sdxl", 8, 20, 3, world)
        except Exception as e:
            others["sdxl_b8_conservative_selection"] = {"error": str(e).split("\n")[0][:120]}
            torch.cuda.synchronize()
        finally:
            for k, v in prev.items():
                _ops.set_policy(k, v)

    out = None
    if rank == 0:
        flops_step = step_flops(w, n)       # this rank's share (clip split: N / world frames)
        if kernels or per_kernel:
            try:
                with open(args.per_kernel_out, "w") as fh:
                    json.dump({"workload": args.workload, "batch_per_gpu": n, "ms_per_step": round(ms_per_step, 3), "scope": args.profile_scope,
                               "scope_ms": round(sum(v["ms_per_step"] for v in kernels.values()), 3),
                               "kernels": kernels, "per_kernel": per_kernel}, fh, indent=0)
            except OSError as e:
                print("bench: per-kernel table not written (%s)" % e, file=sys.stderr)
        top = [{"kernel": r["kernel"][:96], "ms_per_step": r["ms_per_step"], "frac": r["frac"]} for r in per_kernel[1:4]]
        launches = int(round(sum(v["launches_per_step"] for v in kernels.values()))) if kernels else None
        hbm_gb, hbm_src = pmc_total_for(args.workload) if args.batch == 8 and comm is None else (None, None)
        line = {
            "metric": w["metric"], "value": round(value, 3), "unit": "denoise-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "ms_per_step_median": median_ms, "higher_is_better": True,
            "scaling": "strong" if comm is not None else "weak",
            "vs_baseline": None,
            "dtype": "f16",    # MFMA operands fp16; fp32 accumulate / statistics / softmax / residual streams
            "data": "synthetic latents, prompts, condition images; seeded random weights",
            "config": {"workload": args.workload + ": " + ((w["what"] % (n, n)) if "%d" in w["what"] else w["what"])[:150],
                       "baseline_config": w["config"], "batch_per_gpu": n,
                       "parallelism": ("one clip, frames sharded over %d ranks: all_to_all / halo / all-reduce over RCCL, %.0f MB sent "
                                       "per rank and step" % (world, comm.bytes_sent / max(1, args.steps + args.warmup + 2 + 3) / 1e6))
                       if comm is not None else "dp%d, no collective" % world,
                       "launch": mode, "call_form": "controlnet(...) ; adapter(...)" if not args.fused else "controlled_step(...)"},
            "algorithmic_tflop_per_step": round(flops_step / 1e12, 2),
            "mfma_frac_whole_step": round(flops_step / (ms_per_step * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
            "launches_per_step": launches,        # kernel launches of one step (HIP-event leg: eager, lanes off)
            # time of the launches below a quarter of BOTH roofs (per_kernel rows with bound "none"), same HIP-event leg
            "ms_below_quarter_roof": round(below_quarter_ms, 3) if per_kernel else None,
            "launches_below_quarter_roof": int(round(below_quarter_launches)) if per_kernel else None,
            # NOT measured by this run: PMC counters of the newest committed profile of this workload (tools/profile_round.sh)
            "committed_profile": {"hbm_gb_per_step": hbm_gb, "source": hbm_src} if hbm_gb else None,
            "fused_step": fused, "roofline": roof, "cpu_baseline": cpu, "parity_at_bench_config": parity, "other_workloads": others,
            "policy": __import__("ctrl_adapter_amd").ops.policy() or None,      # every CTRL_* switch that is set (csrc/policy.h): None = all defaults
            "next_kernels": top, "per_kernel_file": os.path.basename(args.per_kernel_out) if (kernels or per_kernel) else None,
        }
        out = json.dumps(line)
        if len(out) > 2500:                          # the driver keeps an 8 KB tail of stdout: the headline must fit
            line.pop("next_kernels", None)
            out = json.dumps(line)
        if len(out) > 2500 and line.get("cpu_baseline"):
            line["cpu_baseline"].pop("note", None)
            out = json.dumps(line)
    # The headline must be the LAST thing on stdout.  Libraries write there through C stdio, which is block-buffered on a
    # pipe and flushed at process exit -- RCCL prints a version banner that way when its first communicator is created, and
    # it landed AFTER the JSON line of the first --clip-split run.  So: every rank pushes its C stdio out, all ranks meet, and
    # only then rank 0 prints.
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    dp.barrier(sync_cuda=False)
    if rank == 0:
        print(out, flush=True)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
