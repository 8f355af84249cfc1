"""One clip split across GPUs by frames (BASELINE.json config 4 with fewer clips than GPUs).

Every rank holds F / world consecutive frames of every clip and runs the whole hot path on them; only the three frame-mixing
ops of the adapter exchange data (include/ctrl_hip.h, ctrl_clip_comm):
    temporal transformer all-to-all: frame shards <-> pixel shards around    (model/adapter_spatial_temporal.py:280)
                         the block (or, without it, an all-gather of the K|V rows over the frame axis)
    Conv3d (3,1,1)       +-1-frame halo with the neighbour ranks            (TemporalResnetBlock, :226)
    temporal GroupNorm   all-reduce of the 2 x 32 x clips partial sums      (TemporalResnetBlock, :226)
The native side calls back into a *transport* with byte offsets into an exchange workspace the transport owns:

    RcclTransport        the production transport: RCCL over xGMI enqueued on the forward's stream from C++ (csrc/clip_rccl.cpp), one
                         process per GPU; the sharded forward is hipGraph-capturable with it.  torch.distributed (any backend) only
                         carries the 128-byte communicator id at construction.
    TorchDistTransport   torch.distributed collectives through Python callbacks: backend "nccl" on the GPUs, "gloo" in the CPU
                         tests of the callback contract.  Exchanges are enqueued on the current stream; eager launches only.
    LoopbackTransport    `world` threads of ONE process (virtual ranks on one GPU): used by the single-GPU parity test
                         that proves frame-sharded == unsharded results.

Usage (per rank):  comm = TorchDistTransport(group=None);  adapter(down, mid, num_frames=F_local, ..., clip_comm=comm)
"""
import ctypes as C
import threading

import torch

from . import _lib as L


def shard_frames(x, num_frames, rank, world):
    """[(b f) ...] -> the rank's [(b f_local) ...] slice: frames [rank*Fl, (rank+1)*Fl) of every clip"""
    n = x.shape[0]
    if n % num_frames or num_frames % world:
        raise ValueError("frames per clip (%d) must divide the batch (%d) and be a multiple of world (%d)" % (num_frames, n, world))
    fl = num_frames // world
    v = x.reshape(n // num_frames, num_frames, *x.shape[1:])[:, rank * fl:(rank + 1) * fl]
    return v.reshape(-1, *x.shape[1:]).contiguous()


def unshard_frames(parts, num_frames):
    """inverse of shard_frames over the list of every rank's tensor (rank order)"""
    world = len(parts)
    fl = num_frames // world
    b = parts[0].shape[0] // fl
    return torch.cat([p.reshape(b, fl, *p.shape[1:]) for p in parts], dim=1).reshape(b * num_frames, *parts[0].shape[1:])


class ClipTransport:
    """Exchange workspace + the ctypes callbacks handed to ctrl_adapter_forward_clip_sharded."""

    def __init__(self, rank, world, device):
        self.rank, self.world, self.device = int(rank), int(world), torch.device(device)
        self.ws = None
        self.error = None
        self._cb = (L.CB_GATHER(self._c_gather), L.CB_REDUCE(self._c_reduce), L.CB_HALO(self._c_halo), L.CB_A2A(self._c_a2a))
        self.use_all_to_all = True       # False: the K|V all-gather form of the temporal attention (ctrl_clip_comm::all_to_all = NULL)
        self.bytes_sent = 0              # payload this rank handed to the transport (diagnostics / bench)
        self.ensure(1 << 20)

    def ensure(self, nbytes):
        if self.ws is None or self.ws.numel() < nbytes:
            self.ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)     # torch allocations are >= 256-B aligned
        return self.ws

    def view(self, off, nbytes, dtype=torch.uint8):
        return self.ws[off:off + nbytes].view(dtype)

    def c_struct(self):
        s = L.ClipComm()
        s.rank, s.world = self.rank, self.world
        s.ws, s.ws_bytes = self.ws.data_ptr(), self.ws.numel()
        s.all_gather, s.all_reduce_sum_f32, s.halo_exchange = self._cb[:3]
        s.all_to_all = self._cb[3] if self.use_all_to_all else L.CB_A2A()
        s.user, s.ws_needed = None, 0
        return s

    # ---- C callbacks: never let an exception cross the C frame ----
    def _guard(self, stream, fn, *a):
        """runs one exchange with `stream` (the hipStream_t the forward enqueues on) as torch's current stream, so that the
        collective is ordered exactly where the native side placed it even when the caller's torch stream is another one"""
        try:
            ctx = None
            if self.ws.is_cuda and stream and int(stream) != torch.cuda.current_stream(self.device).cuda_stream:
                ctx = torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=self.device))
            if ctx is None:
                fn(*a)
            else:
                with ctx:
                    fn(*a)
            return 0
        except BaseException as e:      # noqa: BLE001 - reported to the caller of the forward
            self.error = e
            return 1

    def _c_gather(self, user, send_off, recv_off, nbytes, stream):
        self.bytes_sent += nbytes
        return self._guard(stream, self.all_gather, send_off, recv_off, nbytes)

    def _c_reduce(self, user, off, count, stream):
        return self._guard(stream, self.all_reduce_sum_f32, off, count)

    def _c_halo(self, user, sp, sn, rp, rn, nbytes, stream):
        self.bytes_sent += nbytes * ((self.rank > 0) + (self.rank < self.world - 1))
        return self._guard(stream, self.halo_exchange, sp, sn, rp, rn, nbytes)

    def _c_a2a(self, user, send_off, recv_off, nbytes, stream):
        self.bytes_sent += nbytes * (self.world - 1)
        return self._guard(stream, self.all_to_all, send_off, recv_off, nbytes)


class TorchDistTransport(ClipTransport):
    """torch.distributed transport: `group` = the ranks that share one clip (None = the default group)."""

    def __init__(self, group=None, device=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else "cpu"
        super().__init__(rank, world, device)

    def _peer(self, r):
        return self.dist.get_global_rank(self.group, r) if self.group is not None else r

    def all_gather(self, send_off, recv_off, nbytes):
        send = self.view(send_off, nbytes)
        recv = self.view(recv_off, nbytes * self.world)
        try:
            self.dist.all_gather_into_tensor(recv, send, group=self.group)
        except (RuntimeError, NotImplementedError):        # backends without the flat form
            self.dist.all_gather([recv[r * nbytes:(r + 1) * nbytes] for r in range(self.world)], send, group=self.group)

    def all_reduce_sum_f32(self, off, count):
        self.dist.all_reduce(self.view(off, 4 * count, torch.float32), op=self.dist.ReduceOp.SUM, group=self.group)

    def all_to_all(self, send_off, recv_off, nbytes):
        send = self.view(send_off, nbytes * self.world)
        recv = self.view(recv_off, nbytes * self.world)
        try:
            self.dist.all_to_all_single(recv, send, group=self.group)      # RCCL: every peer pair over its own xGMI link
        except (RuntimeError, NotImplementedError):        # backends without it (gloo on some builds): pairwise exchange
            d, ops = self.dist, []
            recv[self.rank * nbytes:(self.rank + 1) * nbytes].copy_(send[self.rank * nbytes:(self.rank + 1) * nbytes])
            for r in range(self.world):
                if r != self.rank:
                    ops += [d.P2POp(d.isend, send[r * nbytes:(r + 1) * nbytes], self._peer(r), self.group),
                            d.P2POp(d.irecv, recv[r * nbytes:(r + 1) * nbytes], self._peer(r), self.group)]
            for req in d.batch_isend_irecv(ops):
                req.wait()

    def halo_exchange(self, sp, sn, rp, rn, nbytes):
        d, ops = self.dist, []
        if self.rank > 0:
            ops += [d.P2POp(d.isend, self.view(sp, nbytes), self._peer(self.rank - 1), self.group),
                    d.P2POp(d.irecv, self.view(rp, nbytes), self._peer(self.rank - 1), self.group)]
        if self.rank < self.world - 1:
            ops += [d.P2POp(d.isend, self.view(sn, nbytes), self._peer(self.rank + 1), self.group),
                    d.P2POp(d.irecv, self.view(rn, nbytes), self._peer(self.rank + 1), self.group)]
        if ops:
            for req in d.batch_isend_irecv(ops):
                req.wait()            # nccl: orders the current stream behind the transfer (no host block)


class RcclTransport(ClipTransport):
    """Native RCCL transport (include/ctrl_hip.h: ctrl_rccl_*).  `group`: the torch.distributed group of the ranks that share one clip
    (None = the default group; any backend) -- used once, to broadcast the communicator ids from the group's rank 0.  With world == 1
    and no process group at all pass rank=0, world=1.
    lanes: communicators (each with its own exchange workspace) chained through ctrl_clip_comm::next_lane -- the adapter runs its four
    pyramid levels on four HIP streams and lane l exchanges on communicator l; lanes=1 (the default) keeps the whole forward on the
    caller's stream.  Several lanes are an opt-in: collectives of different communicators issued from different streams are not
    guaranteed to co-run, so a multi-GPU job should soak-test lanes > 1 before relying on it (only world 1 has run on hardware); under
    hipGraph capture the forward uses one lane whatever the chain length (RCCL refuses a second communicator inside one capture)."""

    def __init__(self, group=None, device=None, rank=None, world=None, lanes=1):
        import torch.distributed as dist
        if rank is None:
            rank, world = dist.get_rank(group), dist.get_world_size(group)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.error = None
        self.rank, self.world, self.device = int(rank), int(world), torch.device(device)
        self.use_all_to_all = True
        self.handles, self.wss, self._chain = [], [], []
        for _ in range(max(1, int(lanes))):
            idbuf = (C.c_uint8 * 128)()
            h = C.c_void_p()
            with torch.cuda.device(self.device):
                if self.rank == 0:
                    L.check(L.lib().ctrl_rccl_unique_id(idbuf))
                if self.world > 1:
                    t = torch.tensor(list(bytes(idbuf)), dtype=torch.uint8)
                    t = t.to(self.device) if dist.get_backend(group) == "nccl" else t
                    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                    idbuf = (C.c_uint8 * 128)(*t.cpu().tolist())
                L.check(L.lib().ctrl_rccl_comm_create(idbuf, self.rank, self.world, C.byref(h)))      # collective
            self.handles.append(h)
        self.ws = None
        self.ensure(1 << 20)

    def ensure(self, nbytes):
        if self.ws is None or self.ws.numel() < nbytes:
            self.wss = [torch.empty(int(nbytes), dtype=torch.uint8, device=self.device) for _ in self.handles]
            self.ws = self.wss[0]
        return self.ws

    @property
    def bytes_sent(self):
        return sum(int(L.lib().ctrl_rccl_comm_bytes_sent(h)) for h in self.handles if h)

    def c_struct(self):
        """the head of the lane chain (the structs of the chain are kept alive on self)"""
        self._chain = [L.ClipComm() for _ in self.handles]
        with torch.cuda.device(self.device):
            for cs, h, ws in zip(self._chain, self.handles, self.wss):
                L.check(L.lib().ctrl_rccl_comm_bind(h, ws.data_ptr(), ws.numel(), 1 if self.use_all_to_all else 0, C.byref(cs)))
        for cs, nxt in zip(self._chain, self._chain[1:]):
            cs.next_lane = C.addressof(nxt)
        return self._chain[0]

    def close(self):
        for h in getattr(self, "handles", []):
            if h:
                L.lib().ctrl_rccl_comm_destroy(h)
        self.handles = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LoopbackWorld:
    """`world` virtual ranks = threads of this process; make one LoopbackTransport per thread with .transport(rank)."""

    def __init__(self, world, device):
        self.world, self.device = world, torch.device(device)
        self.barrier = threading.Barrier(world, timeout=600)     # a failed rank must not hang the others
        self.peers = [None] * world

    def transport(self, rank):
        t = LoopbackTransport(rank, self)
        self.peers[rank] = t
        return t


class LoopbackTransport(ClipTransport):
    def __init__(self, rank, lw):
        self.lw = lw
        super().__init__(rank, lw.world, lw.device)

    def _sync(self):
        if self.ws.is_cuda:
            torch.cuda.current_stream(self.ws.device).synchronize()

    def _rendezvous(self, copy):
        self._sync()                     # my send buffers are complete
        self.lw.barrier.wait()           # ... and so are everyone's
        copy()
        self._sync()
        self.lw.barrier.wait()           # nobody re-uses a send buffer before every reader is done

    def all_gather(self, send_off, recv_off, nbytes):
        def copy():
            for r, p in enumerate(self.lw.peers):
                self.view(recv_off + r * nbytes, nbytes).copy_(p.view(send_off, nbytes))
        self._rendezvous(copy)

    def all_reduce_sum_f32(self, off, count):
        def copy():
            parts = [p.view(off, 4 * count, torch.float32).clone() for p in self.lw.peers]     # rank order: deterministic
            self._tmp = sum(parts[1:], parts[0])
        self._rendezvous(copy)
        self.view(off, 4 * count, torch.float32).copy_(self._tmp)      # after the second barrier: every rank has read the inputs

    def all_to_all(self, send_off, recv_off, nbytes):
        def copy():
            for r, p in enumerate(self.lw.peers):
                self.view(recv_off + r * nbytes, nbytes).copy_(p.view(send_off + self.rank * nbytes, nbytes))
        self._rendezvous(copy)

    def halo_exchange(self, sp, sn, rp, rn, nbytes):
        def copy():
            if self.rank > 0:
                self.view(rp, nbytes).copy_(self.lw.peers[self.rank - 1].view(sn, nbytes))
            if self.rank < self.world - 1:
                self.view(rn, nbytes).copy_(self.lw.peers[self.rank + 1].view(sp, nbytes))
        self._rendezvous(copy)


def run_virtual_ranks(world, fn):
    """runs fn(rank) on `world` threads (each with its own CUDA stream when a GPU is present); returns the results in rank
    order and re-raises the first exception"""
    out, err = [None] * world, [None] * world

    def body(r):
        try:
            if torch.cuda.is_available():
                with torch.cuda.stream(torch.cuda.Stream()):
                    out[r] = fn(r)
                    torch.cuda.current_stream().synchronize()
            else:
                out[r] = fn(r)
        except BaseException as e:      # noqa: BLE001
            err[r] = e
    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in err:
        if e is not None:
            raise e
    return out
