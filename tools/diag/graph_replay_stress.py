"""Diagnostic (GPU): captured adapter forward replayed many times against the eager result -- which outputs differ, how often, with
the fused feed-forward on / off (argv[1] = value of CTRL_FF_FUSED or "-"), after dirtying the allocator with garbage."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]
import torch
import cases
import ctrl_adapter_amd as P
from ctrl_adapter_amd import ops
from ctrl_adapter_amd.synthetic import seeded_init, seeded_tensor
torch.set_grad_enabled(False)
gpu = torch.device("cuda:0")
if len(sys.argv) > 1 and sys.argv[1] != "-":
    ops.set_policy("CTRL_FF_FUSED", sys.argv[1])
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
# dirty memory: fill a few GB with NaN patterns and free them (recycled blocks then hold garbage, as in the middle of the test suite)
junk = [torch.full((256, 1024, 1024), float("nan"), dtype=torch.float16, device=gpu) for _ in range(6)]
del junk
ad = seeded_init(P.ControlNetAdapter(**cases.ADAPTER_SDXL), seed=22).to(gpu)
downs, mid = cases.pyramid_inputs(N=N, h0=32, seed=900, with_mid=False)
kw = dict(num_frames=1, timestep=torch.tensor(499.0).to(gpu), encoder_hidden_states=seeded_tensor((N, 77, 2048), 990).half().to(gpu))
ins = [d.half().to(gpu) for d in downs]
def fwd():
    o, m = ad(ins, **kw)
    return list(o)
fwd()
with ops.Profiler():
    ref = [x.clone() for x in fwd()]
torch.cuda.synchronize()
for k in range(3):
    e = fwd(); torch.cuda.synchronize()
    print("eager %d (lanes on) differing outputs:" % k, [i for i, (a, b) in enumerate(zip(e, ref)) if not torch.equal(a, b)])
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    fwd()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    outs = fwd()
bad_total = {}
for k in range(int(os.environ.get('REPLAYS', '40'))):
    g.replay(); torch.cuda.synchronize()
    bad = [i for i, (a, b) in enumerate(zip(outs, ref)) if not torch.equal(a, b)]
    for i in bad:
        d = (outs[i].float() - ref[i].float()).abs()
        nz = (d > 0).nonzero()
        box = [(int(nz[:, j].min()), int(nz[:, j].max())) for j in range(nz.shape[1])]
        cols = sorted(set(nz[:, 1].tolist()))
        bad_total.setdefault(i, []).append((k, int((d > 0).sum().item()), float(d.max().item()), "box n,c,y,x=%s channels=%s" % (box, cols[:12])))
print("FF_FUSED=%s N=%d: replays with differences per output: %s" % (sys.argv[1] if len(sys.argv) > 1 else "-", N, {i: len(v) for i, v in bad_total.items()}))
for i, v in bad_total.items():
    print("  output %d: first cases (replay, #values, max abs): %s" % (i, v[:4]))
