"""Clip-split multi-GPU mode on CPU: the transports the native side calls back into
(ctrl-adapter_amd/clip_parallel.py) -- torch.distributed over gloo with world_size 2 (the RCCL path runs the same
Python, backend "nccl"), and the thread loopback used by the single-GPU parity test -- plus the frame sharding helpers.
The HIP compute of the sharded forward is covered by tests/test_gpu_e2e.py::test_clip_sharded_* (virtual ranks)."""
import os
import socket
import subprocess
import sys
import textwrap

import torch

from helpers import ROOT

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch
    import torch.distributed as dist
    import ctrl_adapter_amd.dp as dp
    from ctrl_adapter_amd.clip_parallel import TorchDistTransport, shard_frames
    rank, world = dp.init("gloo")
    t = TorchDistTransport()
    assert (t.rank, t.world) == (rank, 2) and t.ws.device.type == "cpu"
    cs = t.c_struct()
    assert cs.rank == rank and cs.world == 2 and cs.ws_bytes == t.ws.numel()
    # all-gather of K|V rows: recv[r] = rank r's send
    n = 4096
    t.view(0, n).copy_(torch.full((n,), 10 + rank, dtype=torch.uint8))
    assert cs.all_gather(None, 0, 8192, n, None) == 0
    got = t.view(8192, 2 * n)
    assert got[:n].eq(10).all() and got[n:].eq(11).all()
    # all-reduce of GroupNorm partial sums
    v = t.view(0, 4 * 64, torch.float32)
    v.copy_(torch.arange(64, dtype=torch.float32) * (rank + 1))
    assert cs.all_reduce_sum_f32(None, 0, 64, None) == 0
    assert torch.equal(t.view(0, 4 * 64, torch.float32), torch.arange(64, dtype=torch.float32) * 3)
    # halo: my first frame -> previous rank's "next" slot, my last frame -> next rank's "prev" slot
    blk = 1024
    t.view(0 * blk, blk).fill_(100 + rank)      # send_prev (my first frame)
    t.view(1 * blk, blk).fill_(200 + rank)      # send_next (my last frame)
    t.view(2 * blk, blk).fill_(7)               # recv_prev
    t.view(3 * blk, blk).fill_(7)               # recv_next
    assert cs.halo_exchange(None, 0, blk, 2 * blk, 3 * blk, blk, None) == 0
    if rank == 0:
        assert t.view(2 * blk, blk).eq(7).all()             # no previous rank: untouched
        assert t.view(3 * blk, blk).eq(101).all()           # rank 1's first frame
    else:
        assert t.view(2 * blk, blk).eq(200).all()           # rank 0's last frame
        assert t.view(3 * blk, blk).eq(7).all()
    # all-to-all (frame shards <-> pixel shards): block r of my send area -> rank r; block r of my recv area <- rank r
    t.view(0, 2 * blk).copy_(torch.cat([torch.full((blk,), 30 + 10 * rank + r, dtype=torch.uint8) for r in range(2)]))
    assert cs.all_to_all(None, 0, 4 * blk, blk, None) == 0
    got = t.view(4 * blk, 2 * blk)
    assert got[:blk].eq(30 + rank).all() and got[blk:].eq(40 + rank).all()
    t.use_all_to_all = False
    assert not t.c_struct().all_to_all                        # NULL pointer: the native side falls back to the K|V all-gather
    t.use_all_to_all = True
    assert t.bytes_sent > 0
    # a failing exchange is reported, never swallowed
    assert cs.all_gather(None, 0, 0, 1 << 40, None) == 1 and t.error is not None
    x = torch.arange(2 * 4 * 3).reshape(8, 3)                # 2 clips x 4 frames
    mine = shard_frames(x, 4, rank, 2)
    assert mine.shape == (4, 3) and mine[0, 0].item() == rank * 2 * 3 and mine[2, 0].item() == (4 + rank * 2) * 3
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_torch_distributed_transport_two_rank_gloo(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o)
        assert "ok" in o


def test_loopback_transport_and_frame_sharding():
    from ctrl_adapter_amd.clip_parallel import LoopbackWorld, run_virtual_ranks, shard_frames, unshard_frames
    W = 4
    lw = LoopbackWorld(W, "cpu")
    ts = [lw.transport(r) for r in range(W)]

    def body(r):
        t = ts[r]
        t.view(0, 16).fill_(r + 1)
        t.all_gather(0, 256, 16)
        g = t.view(256, 16 * W).clone()
        f = t.view(0, 4 * 8, torch.float32)
        f.copy_(torch.full((8,), float(r)))
        t.all_reduce_sum_f32(0, 8)
        s = t.view(0, 4 * 8, torch.float32).clone()
        t.view(1024, 8).fill_(50 + r); t.view(1032, 8).fill_(60 + r); t.view(1040, 16).fill_(0)
        t.halo_exchange(1024, 1032, 1040, 1048, 8)
        h = t.view(1040, 16).clone()
        for k in range(W):
            t.view(2048 + 8 * k, 8).fill_(10 * r + k)           # block k of my send area goes to rank k
        t.all_to_all(2048, 4096, 8)
        return g, s, h, t.view(4096, 8 * W).clone()
    res = run_virtual_ranks(W, body)
    for r, (g, s, h, a2a) in enumerate(res):
        assert all(a2a[8 * k:8 * (k + 1)].eq(10 * k + r).all() for k in range(W))
        assert all(g[16 * k:16 * (k + 1)].eq(k + 1).all() for k in range(W))
        assert s.eq(0.0 + 1 + 2 + 3).all()
        assert h[:8].eq(60 + r - 1 if r > 0 else 0).all() and h[8:].eq(50 + r + 1 if r < W - 1 else 0).all()
    x = torch.randn(3 * 8, 5, 2)
    parts = [shard_frames(x, 8, r, W) for r in range(W)]
    assert parts[0].shape == (6, 5, 2) and torch.equal(unshard_frames(parts, 8), x)
