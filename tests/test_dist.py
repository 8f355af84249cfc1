"""N>1 path on CPU: world_size-2 gloo processes exercise the sharding and the max-over-ranks timing that bench.py uses
on RCCL.  The data path itself has no collective (images / whole clips are independent)."""
import os
import socket
import subprocess
import sys
import textwrap

from helpers import ROOT

WORKER = textwrap.dedent("""
    import os, sys, time
    sys.path.insert(0, %r)
    import torch
    import ctrl_adapter_amd.dp as dp
    rank, world = dp.init("gloo")
    assert world == 2
    b, e = dp.shard(17, rank, world)
    assert (b, e) == ((0, 9) if rank == 0 else (9, 17)), (b, e)
    # a rank that is slower by construction must define the reported time
    def work():
        time.sleep(0.02 * (rank + 1))
    el = dp.timed_region(work, 5, device="cpu")
    assert 0.19 < el < 0.6, el
    thr = dp.aggregate_throughput(8, 5, el, world)
    assert abs(thr - 2 * 8 * 5 / el) < 1e-9
    # every rank sees the same maximum
    t = torch.tensor([el], dtype=torch.float64)
    import torch.distributed as dist
    lst = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(lst, t)
    assert abs(lst[0].item() - lst[1].item()) < 1e-12
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_two_rank_gloo_sharding_and_timing(tmp_path):
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
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o)
        assert "ok" in o


def test_shard_covers_everything():
    import ctrl_adapter_amd.dp as dp
    for total in (1, 7, 8, 64, 1001):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                b, e = dp.shard(total, r, world)
                cover += list(range(b, e))
            assert cover == list(range(total))


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` without a launcher must start N ranks itself (it asserted on WORLD_SIZE).
    There is no GPU here, so every rank stops at the loud no-GPU error.  That N ranks were launched with the torchrun
    environment (RANK / WORLD_SIZE / MASTER_ADDR = 127.0.0.1) is read from the marker file every rank writes first thing
    (CTRL_BENCH_MARKER_DIR) -- not from the ranks' output: torchrun tears the sibling rank down as soon as the first one
    fails, so how many of them get to print their error is a race."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["CTRL_BENCH_MARKER_DIR"] = str(tmp_path)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = p.stdout.decode()
    markers = sorted(f.name for f in tmp_path.iterdir())
    assert markers == ["rank0_of_2", "rank1_of_2"], (markers, out[-2000:])
    assert all((tmp_path / m).read_text().strip() == "127.0.0.1" for m in markers)
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        assert p.returncode == 0 and '"n_gpus": 2' in out, out[-2000:]
    else:
        assert p.returncode != 0                     # the parent relays the ranks' failure
        assert out.count("bench.py needs a GPU") >= 1 or out.count("has no GPU") >= 1, out[-2000:]
