"""ControlNet-only vs adapter-only time at the bench shapes (HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch, bench
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
P, cn, ad = bench.build_models(dev, "sdxl")
x = bench.make_inputs(dev, "sdxl", 8, 1234)
t = torch.tensor([499.0], device=dev)
s = P.pool_latents(x["latents"], (64, 64))
down, mid = cn(s, t, x["ehs_c"], x["cond"], return_dict=False)
def tm(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("controlnet ms", tm(lambda: cn(s, t, x["ehs_c"], x["cond"], return_dict=False)))
print("adapter    ms", tm(lambda: ad(down, num_frames=1, timestep=t, encoder_hidden_states=x["ehs_a"])))
from ctrl_adapter_amd import ops
with ops.Profiler() as prof:
    cn(s, t, x["ehs_c"], x["cond"], return_dict=False)
print("controlnet classes:", {k: (round(v[0], 3), v[1]) for k, v in sorted(prof.rows.items(), key=lambda kv: -kv[1][0])})
