"""Runs a few launches of one dominant kernel shape (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctrl_adapter_amd  # noqa
from ctrl_adapter_amd import ops

which = sys.argv[1] if len(sys.argv) > 1 else "gemm"
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g).half().to(dev)
if which == "gemm":
    x, w = R(131072, 512), R(4096, 512)
    for _ in range(5):
        ops.linear(x, w, geglu=True)
elif which == "gemm2":
    x, w = R(131072, 2048), R(512, 2048)
    for _ in range(5):
        ops.linear(x, w)
elif which == "conv":
    x, w = R(8, 128, 128, 320), R(320, 9 * 320)
    for _ in range(5):
        ops.conv2d(x, w, 320, taps=9)
else:
    B, heads, L = 8, 5, 16384
    Cc = heads * 64
    q, k, vt = R(B * L, Cc), R(B * L, Cc), R(B, Cc, L)
    for _ in range(3):      # "attn_fold": the product path's form (K pre-scaled, maximum folded into the accumulator init)
        ops.flash_attn(q, Cc, k, Cc, vt, L, B, heads, 64, L, L, k_prescaled=(which == "attn_fold"))
torch.cuda.synchronize()
