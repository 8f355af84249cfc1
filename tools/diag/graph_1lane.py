"""Diagnostic (GPU): graph replays of the adapter forward with CTRL_ADAPTER_LANES=1 -- what breaks from the second replay on?"""
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
ad = seeded_init(P.ControlNetAdapter(**cases.ADAPTER_SDXL), seed=22).to(gpu)
downs, mid = cases.pyramid_inputs(N=4, h0=32, seed=900, with_mid=False)
kw = dict(num_frames=1, timestep=torch.tensor(499.0).to(gpu), encoder_hidden_states=seeded_tensor((4, 77, 2048), 990).half().to(gpu))
ins = [d.half().to(gpu) for d in downs]
ins_copy = [x.clone() for x in ins]
ops.set_group_launches(int(os.environ.get("GRP", "1")))
def fwd():
    o, m = ad(ins, **kw)
    return list(o)
ref = [x.clone() for x in fwd()]
torch.cuda.synchronize()
ref2 = [x.clone() for x in fwd()]
torch.cuda.synchronize()
print("eager twice equal:", all(torch.equal(a, b) for a, b in zip(ref, ref2)))
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    fwd()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    outs = fwd()
pre = [torch.empty_like(x) for x in outs]          # destinations allocated BEFORE any replay: no allocation between replays
for k in range(4):
    g.replay(); torch.cuda.synchronize()
    bad = [i for i, (a, b) in enumerate(zip(outs, ref)) if not torch.equal(a, b)]
    print("replay %d (no allocation in between): differing outputs %s; inputs intact %s" % (k, bad, all(torch.equal(a, b) for a, b in zip(ins, ins_copy))))
for k in range(3):
    g.replay(); torch.cuda.synchronize()
    c = [x.clone() for x in outs]
    bad = [i for i, (a, b) in enumerate(zip(c, ref)) if not torch.equal(a, b)]
    print("replay %d (clone after): differing outputs %s" % (k, bad))
# back-to-back replays without sync
for k in range(3):
    g.replay()
torch.cuda.synchronize()
print("3 back-to-back replays: differing", [i for i, (a, b) in enumerate(zip(outs, ref)) if not torch.equal(a, b)])
e = [x.clone() for x in fwd()]; torch.cuda.synchronize()
print("eager after the replays equal:", all(torch.equal(a, b) for a, b in zip(e, ref)))
