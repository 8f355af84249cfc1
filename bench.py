"""Benchmark of the Ctrl-Adapter denoising hot path on MI355X (BASELINE.json metric).

One "step" = what the reference's SDXL pipeline does per denoising step around the UNet call
(sdxl/pipelines/sdxl_controlnet_adapter_pipeline.py:1306-1343): pool the latents to 64x64, ControlNetModel.forward,
ControlNetAdapter.forward -- for a batch of 8 images at 1024^2 (BASELINE.json configs[1]), synthetic latents / prompts /
condition images already resident in HBM, seeded random weights of the real architecture (361 M + 184 M parameters).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU, RCCL)

Prints ONE JSON line (rank 0).  Besides the driver contract it carries:
  roofline      the dominant kernel class (flash attention), algorithmic FLOPs / HIP-event time, vs the dense fp16 MFMA peak
  cpu_baseline  the pure-PyTorch fp32 oracle on the host cores, on a bounded sample (1 image of the batch)
  kernels       per-kernel-class time of one profiled step (HIP events around every launch on the launch stream)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]

import torch  # noqa: E402

MFMA_PEAK_TFLOPS = 2500.0      # dense fp16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md chip table
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


def build_models(dev, workload):
    import ctrl_adapter_amd as P
    from oracle.init import seeded_init      # seeded random weights (shared with the parity tests)
    cn = seeded_init(P.ControlNetModel(cross_attention_dim=768), seed=11).to(dev)
    ad = seeded_init(P.ControlNetAdapter(**(SDXL_ADAPTER if workload == "sdxl" else VIDEO_ADAPTER)), seed=22).to(dev)
    return P, cn, ad


def make_inputs(dev, workload, n, seed):
    g = torch.Generator().manual_seed(seed)
    if workload == "sdxl":
        lat = torch.randn(n, 4, 128, 128, generator=g)
        ehs_a = torch.randn(n, 77, 2048, generator=g)
    else:
        lat = torch.randn(n, 4, 64, 64, generator=g)
        ehs_a = torch.randn(1, 1, 1024, generator=g)
    d = dict(latents=lat, ehs_c=torch.randn(n, 77, 768, generator=g), cond=torch.rand(n, 3, 512, 512, generator=g), ehs_a=ehs_a)
    return {k: v.half().to(dev) for k, v in d.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU entering the hot path (BASELINE: 8)")
    ap.add_argument("--workload", default="sdxl", choices=["sdxl", "svd16"])
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of a captured hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fused", action="store_true",
                    help="time the fused controlled_step(controlnet, adapter, ...) instead of the pipelines' two calls "
                         "controlnet(...) ; adapter(...) (same arithmetic, bit-identical results, ~1 % faster)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    assert args.gpus == world, "--gpus must equal the number of launched ranks"
    torch.set_grad_enabled(False)

    workload = "sdxl" if args.workload == "sdxl" else "video"
    n = args.batch if workload == "sdxl" else 32          # svd16: one CFG pair of 16-frame clips
    P, cn, ad = build_models(dev, workload)
    x = make_inputs(dev, workload, n, seed=1234 + rank)   # every rank owns different images (data parallel, no collective)
    t = torch.tensor([499.0], device=dev)
    nf = 1 if workload == "sdxl" else 16
    skip_conv_in = workload != "sdxl"                     # configs/svd_train_depth.yaml:59

    def step_separate():
        s = P.pool_latents(x["latents"], (64, 64))
        down, mid = cn(s, t, x["ehs_c"], x["cond"], conditioning_scale=1.0, return_dict=False, skip_conv_in=skip_conv_in)
        return ad(down, mid_block_res_sample=mid, num_frames=nf, timestep=t, encoder_hidden_states=x["ehs_a"])

    def step_fused():
        s = P.pool_latents(x["latents"], (64, 64))
        return P.controlled_step(cn, ad, s, t, x["ehs_c"], x["cond"], 1.0, skip_conv_in=skip_conv_in, num_frames=nf,
                                 adapter_encoder_hidden_states=x["ehs_a"])[1]

    step = step_fused if args.fused else step_separate

    # eager warm-up: builds the plans (weight packing) and sizes the workspaces
    for _ in range(2):
        step()
    torch.cuda.synchronize()

    run = step
    mode = "eager"
    if not args.no_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = step()
            run = graph.replay
            mode = "hipgraph"
        except Exception as e:   # capture is an optimisation only
            print("bench: hipGraph capture failed (%s); running eager" % str(e).split("\n")[0], file=sys.stderr)
            torch.cuda.synchronize()
            run = step

    for _ in range(args.warmup):
        run()

    import ctrl_adapter_amd.dp as dp
    elapsed = dp.timed_region(run, args.steps, device=dev)        # barrier + sync on both sides, MAX over ranks
    ms_per_step = elapsed / args.steps * 1e3
    value = dp.aggregate_throughput(1, args.steps, elapsed, world)   # whole-job denoise-steps/s (each rank: batch of 8)

    # ---- roofline leg: one eager step with HIP events around every launch (on the launch stream) ----
    from ctrl_adapter_amd import ops
    kernels = {}
    roof = None
    if rank == 0:
        reps = 3
        with ops.Profiler() as prof:
            for _ in range(reps):
                step()
        kernels = {}
        for k, v in prof.rows.items():
            ms = v[0] / reps
            kernels[k] = {"ms_per_step": round(ms, 4), "launches_per_step": v[1] // reps,
                          "tflops": round(v[2] / reps / (ms * 1e-3) / 1e12, 1) if v[2] else None,
                          "gbs": round(v[3] / reps / (ms * 1e-3) / 1e9, 1) if v[3] else None}
        dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
        kd, raw = kernels[dom], prof.rows[dom]
        # HBM traffic per launch of that class: PMC FETCH_SIZE/WRITE_SIZE passes of this same command, measured with
        # rocprofv3 (cannot run inside the benchmark), corrected per MI355X_MICROARCH.md, committed under profiles/
        traffic = None
        import glob
        tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic*.json")))      # latest round / version
        tfile = tfiles[-1] if tfiles else ""
        if workload == "sdxl" and n == 8 and tfile:
            with open(tfile) as fh:
                traffic = json.load(fh)["classes"].get(dom, {}).get("hbm_bytes_per_launch_corrected")
        if raw[2] > 0:      # MFMA-bound class (implicit GEMM / flash attention): algorithmic FLOPs / HIP-event time
            roof = {"kernel": dom, "bound": "mfma", "achieved": kd["tflops"], "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(kd["tflops"] / MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                    "flops_per_launch": raw[2] / raw[1], "avg_launch_ms": raw[0] / raw[1], "launches_per_step": kd["launches_per_step"]}
        else:
            roof = {"kernel": dom, "bound": "hbm", "achieved": kd["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(kd["gbs"] / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "bytes_per_launch": raw[3] / raw[1], "avg_launch_ms": raw[0] / raw[1], "launches_per_step": kd["launches_per_step"]}

    # ---- cpu_baseline leg: the fp32 oracle on the host cores, bounded sample (rank 0, N=1 only) ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and workload == "sdxl":
        import cases  # noqa: F401
        from oracle.init import seeded_init
        from oracle.controlnet import ControlNetOracle
        from oracle.adapter import ControlNetAdapterOracle
        cores = min(os.cpu_count() or 1, 32)        # more threads than this slow the fp32 oracle down (NUMA / oversubscription)
        torch.set_num_threads(cores)
        oc = seeded_init(ControlNetOracle(cross_attention_dim=768).eval(), seed=11)
        oa = seeded_init(ControlNetAdapterOracle(**SDXL_ADAPTER).eval(), seed=22)
        xc = {k: v[:1].float().cpu() for k, v in x.items()}

        def cpu_step():
            s = torch.nn.functional.adaptive_avg_pool2d(xc["latents"], (64, 64))
            d, m = oc(s, torch.tensor(499.0), xc["ehs_c"], xc["cond"])
            return oa(d, num_frames=1, timestep=torch.tensor(499.0), encoder_hidden_states=xc["ehs_a"])
        c0 = time.perf_counter()
        reps = 4                                    # bounded sample: one image of the batch, ~10-15 s of CPU work
        for _ in range(reps):
            cpu_step()
        per_img = (time.perf_counter() - c0) / reps
        cpu = {"value": round(1.0 / (per_img * n), 5), "unit": "denoise-steps/s", "cores": cores, "kind": "port",
               "sample": "1 image of the batch of %d (1/%d step) x %d reps, fp32 PyTorch oracle; value = 1/(%d x %.2f s)"
                         % (n, n, reps, n, per_img)}

    if rank == 0:
        flops_step = (0.2835 + 2.258) * 1e12 * n if workload == "sdxl" else None
        line = {
            "metric": "denoise-steps/s (ControlNet+adapter fwd) SDXL 1024^2 b=8" if workload == "sdxl"
                      else "denoise-steps/s (ControlNet+adapter fwd) SVD 16-frame clip (CFG pair, 32 frames)",
            "value": round(value, 3), "unit": "denoise-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16",    # MFMA operands fp16; fp32 accumulate / statistics / softmax / residual streams
            "data": "synthetic latents, prompts, condition images; seeded random weights",
            "config": {"workload": "SDXL depth 1024^2 batch=8 per GPU (N=8 images enter ControlNet+adapter, no CFG doubling); "
                                   "pool->ControlNet(SD1.5, 64x64 latents, 512^2 cond)->Ctrl-Adapter(A,B,C x3, up x2)"
                                   if workload == "sdxl" else "SVD depth, 16 frames, CFG pair (N=32 frames), skip_conv_in",
                       "batch_per_gpu": n, "parallelism": "dp%d (images sharded, no collective)" % world, "launch": mode,
                       "call_form": "controlnet(...) ; adapter(...)" if not args.fused else
                                    "controlled_step(controlnet, adapter, ...) = both forwards, overlapped (bit-identical results)"},
            "algorithmic_tflop_per_step": round(flops_step / 1e12, 2) if flops_step else None,
            "mfma_frac_whole_step": round(flops_step / (ms_per_step * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4) if flops_step else None,
            "roofline": roof, "cpu_baseline": cpu, "kernels": kernels,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
