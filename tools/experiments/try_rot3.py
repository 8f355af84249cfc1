"""One correctness check + one timing of the rot3 attention experiment (CTRL_ATTN_ROT3=1 selects it)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ctrl_adapter_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g).half().to(dev)
B, heads, L = 1, 2, 2100
C = heads * 64
Lp = (L + 63) // 64 * 64
q, k, v = R(B * L, C), R(B * L, C), R(B * L, C)
vt = torch.zeros(B, C, Lp, dtype=torch.float16, device=dev); vt[:, :, :L] = v.view(B, L, C).permute(0, 2, 1)
o = ops.flash_attn(q, C, k, C, vt, Lp, B, heads, 64, L, L)
qq, kk, vv = (x.float().view(B, L, heads, 64).permute(0, 2, 1, 3) for x in (q, k, v))
ref = torch.nn.functional.scaled_dot_product_attention(qq, kk, vv).permute(0, 2, 1, 3).reshape(B * L, C)
print("rel_inf", ((o.float() - ref).abs().max() / ref.abs().max()).item())
B, heads, L = 8, 5, 16384
C = heads * 64
q, k, vt = R(B * L, C), R(B * L, C), R(B, C, L)
for _ in range(2): ops.flash_attn(q, C, k, C, vt, L, B, heads, 64, L, L)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): ops.flash_attn(q, C, k, C, vt, L, B, heads, 64, L, L)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
print("L16384 %.3f ms %.1f TFLOP/s" % (ms, 4.0 * B * heads * L * L * 64 / ms / 1e9))
