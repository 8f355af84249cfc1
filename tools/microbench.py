"""Micro-benchmarks of the dominant kernels at the SDXL b=8 shapes (HIP events on the current stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctrl_adapter_amd  # noqa
from ctrl_adapter_amd import ops


def timeit(fn, iters=10, warm=3):
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
    g = torch.Generator(device="cpu").manual_seed(0)
    R = lambda *s: (torch.randn(*s, generator=g)).half().to(dev)
    print("== flash attention (adapter self-attn) ==")
    for (B, heads, L) in [(8, 5, 16384), (8, 10, 4096), (8, 20, 1024), (8, 5, 4096)]:
        Cc = heads * 64
        q, k = R(B * L, Cc), R(B * L, Cc)
        vt = R(B, Cc, L)
        ms = timeit(lambda: ops.flash_attn(q, Cc, k, Cc, vt, L, B, heads, 64, L, L), iters=5)
        fl = 4.0 * B * heads * L * L * 64
        print("attn B%d h%d L%d: %.3f ms  %.1f TFLOP/s" % (B, heads, L, ms, fl / ms / 1e9))
    print("== cross attention Lk=77 ==")
    B, heads, L = 8, 5, 16384
    Cc = heads * 64
    q, k, vt = R(B * L, Cc), R(B * 77, Cc), R(B, Cc, 128)
    ms = timeit(lambda: ops.flash_attn(q, Cc, k, Cc, vt, 128, B, heads, 64, L, 77), iters=5)
    print("xattn L%d: %.3f ms" % (L, ms))
    print("== GEMMs ==")
    for (M, N, K, geglu) in [(131072, 512, 320, False), (131072, 960, 512, False), (131072, 4096, 512, True),
                             (131072, 512, 2048, False), (32768, 8192, 1280, False), (32768, 1280, 1280, False),
                             (8192, 1280, 1280, False), (512, 1280, 1280, False)]:
        x = R(M, K)
        w = R(N, K)
        ms = timeit(lambda: ops.linear(x, w, geglu=geglu), iters=5)
        print("gemm M%d N%d K%d geglu=%d: %.3f ms  %.1f TFLOP/s" % (M, N, K, geglu, ms, 2.0 * M * N * K / ms / 1e9))
    print("== conv3x3 ==")
    for (n, c, co, h, up) in [(8, 320, 320, 128, 1), (8, 320, 320, 64, 2), (8, 640, 640, 64, 1), (8, 1280, 1280, 32, 1),
                              (8, 320, 320, 64, 1), (8, 1280, 1280, 8, 1), (8, 1280, 1280, 16, 1), (8, 640, 640, 32, 1)]:
        x = R(n, h, h, c)
        w = R(co, 9 * c)
        ms = timeit(lambda: ops.conv2d(x, w, co, taps=9, up=up), iters=5)
        ho = h * up
        print("conv3x3 n%d %d->%d @%d(up%d): %.3f ms  %.1f TFLOP/s" % (n, c, co, ho, up, ms, 2.0 * n * ho * ho * co * 9 * c / ms / 1e9))
        if n * ho * ho <= 32768:
            ws = torch.empty(16 * n * ho * ho * co, dtype=torch.float32, device=dev)
            ms = timeit(lambda: ops.conv2d(x, w, co, taps=9, up=up, splitk_ws=ws), iters=5)
            print("   + split-K scratch: %.3f ms  %.1f TFLOP/s" % (ms, 2.0 * n * ho * ho * co * 9 * c / ms / 1e9))
    print("== norms (HBM) ==")
    for (n, hw, c) in [(8, 16384, 320), (8, 4096, 640)]:
        x = R(n, hw, c)
        gm, bt = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        ms = timeit(lambda: ops.groupnorm(x, gm, bt, n, hw, silu=True), iters=5)
        print("groupnorm(+alloc) n%d hw%d c%d: %.3f ms  %.1f GB/s (3 passes)" % (n, hw, c, ms, 3 * x.numel() * 2 / ms / 1e6))
    x = R(131072, 512)
    gm, bt = torch.ones(512, device=dev), torch.zeros(512, device=dev)
    ms = timeit(lambda: ops.layernorm(x, gm, bt), iters=5)
    print("layernorm 131072x512: %.3f ms  %.1f GB/s" % (ms, 2 * x.numel() * 2 / ms / 1e6))


if __name__ == "__main__":
    main()
