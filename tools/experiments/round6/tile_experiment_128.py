"""Round 6: every forced tile family (CTRL_IGEMM_FORCE) against the dispatcher's choice on the short-K token GEMMs of the 128^2 level with
the epilogues they have TODAY (fp16 token stream: fp16 residual in, fp16 out; video: fp32 residual in, fp32 master + fp16 mirror out).
HIP events on the current stream, median of 5 x 8 launches.  python tools/experiments/round6/tile_experiment_128.py > out.txt"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import ctrl_adapter_amd  # noqa
from ctrl_adapter_amd import ops

TILES = ("", "wide", "128x256", "256x128", "4w128x256", "4w256x128", "128x128x32", "128x128x64", "128x64x64", "64x64x64")


def timeit(fn, iters=8, warm=2, reps=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / iters)
    return sorted(out)[len(out) // 2]


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    R = lambda *s: torch.randn(*s, generator=g).half().to(dev)
    # name, M, N, K, residual kind (None | "h" fp16 stream | "f" fp32 stream + mirror)
    cases = [("attn out 320->512 + fp16 stream", 131072, 512, 320, "h"), ("attn out 320->512 + fp32 stream", 131072, 512, 320, "f"),
             ("proj_in 320->512 fp16", 131072, 512, 320, None), ("q|k|v 512->960 fp16", 131072, 960, 512, None),
             ("q 512->320 fp16", 131072, 320, 512, None), ("q|k 512->640 fp16", 131072, 640, 512, None),
             ("attn out 640->512 + fp16 stream @64^2", 32768, 512, 640, "h"), ("q|k|v 512->1920 @64^2", 32768, 1920, 512, None),
             ("cn proj 320->320 @64^2 b8", 32768, 320, 320, "f"), ("cn q|k|v 320->960 b8", 32768, 960, 320, None)]
    for name, M, N, K, rk in cases:
        x, w, b = R(M, K), R(N, K), torch.randn(N, generator=g).to(dev)
        res = None if rk is None else (torch.randn(M, N, generator=g).to(dev) if rk == "f" else R(M, N))
        out = torch.empty(M, N, dtype=torch.float32 if rk == "f" else torch.float16, device=dev)
        mir = torch.empty(M, N, dtype=torch.float16, device=dev) if rk == "f" else None
        wp = ops.pack_linear_w(w, geglu=False)
        bp = ops.pack_vec(b, geglu=False)

        def run():
            ops.igemm(x, K, wp, M, N, K, bias=bp, res=res, ldres=N, segs=[(out, N, 0, N, ops.SEG_ROW, 1)], out16=mir, ld16=N)
        ref = None
        line = "%-40s M%-6d N%-4d K%-4d" % (name, M, N, K)
        best = (1e9, "")
        for tile in TILES:
            ops.set_policy("CTRL_IGEMM_FORCE", tile or None)
            try:
                ms = timeit(run)
            except Exception as e:          # a tile that does not take this shape
                line += "  %s --" % tile
                continue
            o = out.float().clone()
            if ref is None:
                ref = o
            else:
                assert (o - ref).abs().max().item() <= 2e-2 * ref.abs().max().item(), (name, tile)
            line += "  %s %.3f" % (tile or "default", ms)
            if ms < best[0]:
                best = (ms, tile or "default")
        print(line + "   | best %s %.3f ms (%.0f TFLOP/s)" % (best[1], best[0], 2.0 * M * N * K / best[0] / 1e9), flush=True)
    ops.set_policy("CTRL_IGEMM_FORCE", None)


if __name__ == "__main__":
    main()
