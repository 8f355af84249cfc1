"""A/B of igemm tile configurations on the token GEMMs of the SDXL / video adapters with their REAL epilogues (fp32 residual
in, fp32 master + fp16 mirror out; GEGLU), CTRL_IGEMM_FORCE=<tile> vs the dispatcher's choice.  HIP events, current stream."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctrl_adapter_amd  # noqa
from ctrl_adapter_amd import ops


def timeit(fn, iters=8, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    R = lambda *s: torch.randn(*s, generator=g).half().to(dev)
    cases = [("out-proj  f32 stream", 131072, 512, 320, False, True), ("proj_in   f32 stream", 131072, 512, 320, False, True),
             ("ff2       f32 stream", 131072, 512, 2048, False, True), ("qk        fp16 out", 131072, 640, 512, False, False),
             ("ff1 geglu fp16 out", 131072, 4096, 512, True, False), ("ff1 geglu M32768", 32768, 4096, 512, True, False),
             ("out-proj  M32768", 32768, 512, 640, False, True), ("to_q 512->512 fp16", 131072, 512, 512, False, False),
             ("q/v 512->320 fp16", 131072, 320, 512, False, False), ("qk 512->1280 fp16", 32768, 1280, 512, False, False),
             ("ff2 M32768 f32", 32768, 512, 2048, False, True), ("cn ff1 geglu 320", 32768, 2560, 320, True, False)]
    for name, M, N, K, geglu, f32 in cases:
        x, w, b = R(M, K), R(N, K), torch.randn(N, generator=g).to(dev)
        on = N // 2 if geglu else N
        res = torch.randn(M, on, generator=g).to(dev) if f32 else None
        out = torch.empty(M, on, dtype=torch.float32 if f32 else torch.float16, device=dev)
        mir = torch.empty(M, on, dtype=torch.float16, device=dev) if f32 else None
        wp = ops.pack_linear_w(w, geglu=geglu)
        bp = ops.pack_vec(b, geglu=geglu)

        def run():
            ops.igemm(x, K, wp, M, N, K, bias=bp, res=res, ldres=on, geglu=geglu, segs=[(out, on, 0, on, ops.SEG_ROW, 1)],
                      out16=mir, ld16=on)
        line = "%-22s M%d N%d K%d:" % (name, M, N, K)
        for tile in ("", "128x256", "256x128", "128x320"):
            ops.set_policy("CTRL_IGEMM_FORCE", tile or None)
            ms = timeit(run)
            line += "  %s %.3f ms %.0f TF" % (tile or "default", ms, 2.0 * M * N * K / ms / 1e9)
        print(line)
    ops.set_policy("CTRL_IGEMM_FORCE", None)


if __name__ == "__main__":
    main()
