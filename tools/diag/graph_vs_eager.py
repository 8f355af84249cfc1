"""Diagnostic (GPU): is the adapter forward bit-reproducible across eager runs / graph replays, with and without grouped launches?"""
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
which = sys.argv[1] if len(sys.argv) > 1 else "sdxl"
if which == "sdxl":
    ad = seeded_init(P.ControlNetAdapter(**cases.ADAPTER_SDXL), seed=22).to(gpu)
    downs, mid = cases.pyramid_inputs(N=4, h0=32, seed=900, with_mid=False)
    kw = dict(num_frames=1, timestep=torch.tensor(499.0).to(gpu), encoder_hidden_states=seeded_tensor((4, 77, 2048), 990).half().to(gpu))
else:
    ad = seeded_init(P.ControlNetAdapter(**cases.ADAPTER_VIDEO), seed=33).to(gpu)
    downs, mid = cases.pyramid_inputs(N=8, h0=16, seed=910, with_mid=True)
    kw = dict(num_frames=4, timestep=torch.tensor(961.0).to(gpu), encoder_hidden_states=seeded_tensor((1, 1, 1024), 991).half().to(gpu),
              mid_block_res_sample=mid.half().to(gpu))
ins = [d.half().to(gpu) for d in downs]

def fwd():
    o, m = ad(ins, **kw)
    return [x.clone() for x in list(o) + ([m] if m is not None else [])]

def diff(a, b):
    return [(i, (x.float() - y.float()).abs().max().item()) for i, (x, y) in enumerate(zip(a, b)) if not torch.equal(x, y)]

for mode in (0, 1, 2):
    ops.set_group_launches(mode)
    fwd(); torch.cuda.synchronize()
    with ops.Profiler():
        ref = fwd()                       # one lane (the profiler switches the lanes off)
    torch.cuda.synchronize()
    eager = []
    for _ in range(4):
        eager.append(fwd()); torch.cuda.synchronize()
    print("mode %d eager(lanes) vs one-lane:" % mode, [diff(e, ref) for e in eager])
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fwd()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        o, m = ad(ins, **kw)
    outs = list(o) + ([m] if m is not None else [])
    reps = []
    for _ in range(6):
        g.replay(); torch.cuda.synchronize()
        reps.append([x.clone() for x in outs])
    print("mode %d graph replays vs one-lane:" % mode, [diff(r, ref) for r in reps])
    del g
ops.set_group_launches(1)
