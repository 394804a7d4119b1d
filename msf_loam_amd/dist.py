"""Multi-GPU plumbing: scans shard embarrassingly across ranks (one process per GPU), the map is
replicated, and the only data-path collective is the pose gather (RCCL all_gather over xGMI;
`gloo` in the CPU tests).  SURVEY.md §8e.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_items, rank, world_size):
    """Contiguous block partition [lo, hi) of n_items for `rank`; sizes differ by at most one."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_offsets(off, lo, hi):
    """Slice a B+1 prefix-offset array to scans [lo, hi): (new offsets starting at 0, point range)."""
    off = np.asarray(off)
    return (off[lo:hi + 1] - off[lo]).astype(np.int32), (int(off[lo]), int(off[hi]))


def broadcast_map(corner, surf, src=0, device=None, force=False):
    """Replicate the local map (float32 (n,4) arrays) from `src` to every rank.
    force: run the collectives even in a one-rank group (the 1-rank RCCL dry run of tests/test_gpu_bench_ranks.py)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return corner, surf
    dev = device or torch.device("cpu")
    rank = dist.get_rank()
    sizes = torch.tensor([len(corner) if rank == src else 0, len(surf) if rank == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(sizes, src)
    out = []
    for arr, n in ((corner, int(sizes[0])), (surf, int(sizes[1]))):
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to(dev) if rank == src else torch.empty((n, 4), dtype=torch.float32, device=dev)
        dist.broadcast(t, src)
        out.append(t)
    return out[0], out[1]


class PoseGather:
    """all_gather of per-rank (B,7) f64 poses + (B,) i32 status.  ONE collective per call: status
    rides as an 8th double next to the pose (56 KB + 8 KB per 1024 scans: latency-bound on xGMI, so
    one launch instead of two).  Equal B on every rank (weak scaling); use gather_ragged otherwise."""

    def __init__(self, B, device):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.B = B
        self.send = torch.empty((B, 8), dtype=torch.float64, device=device)
        self.recv = torch.empty((self.world, B, 8), dtype=torch.float64, device=device)

    def all_gather(self, poses, status):
        self.send[:, :7].copy_(poses.reshape(self.B, 7))
        self.send[:, 7].copy_(status.reshape(self.B))
        if not dist.is_initialized():
            self.recv[0].copy_(self.send)
        else:
            dist.all_gather_into_tensor(self.recv.view(-1), self.send.view(-1))
        return self.recv[:, :, :7], self.recv[:, :, 7]      # status as f64 view; cast when consumed

    # --- pipelined form: the collective of batch k runs on a side stream under the compute of batch k+1 ---
    def all_gather_async(self, poses, status):
        """Pack on the current stream, gather on a side stream.  Results are valid after wait().  Two send /
        receive buffers alternate, so the caller may overwrite `poses` right away and the previous result
        stays readable until the call after next."""
        if self.send.device.type != "cuda":
            return self.all_gather(poses, status)
        if not hasattr(self, "_side"):
            self._side = torch.cuda.Stream(device=self.send.device)
            self._bufs = [(self.send, self.recv), (torch.empty_like(self.send), torch.empty_like(self.recv))]
            self._done = [None, None]
            self._k = 0
        k = self._k
        self._k ^= 1
        send, recv = self._bufs[k]
        cur = torch.cuda.current_stream(self.send.device)
        if self._done[k] is not None:
            cur.wait_event(self._done[k])                     # the collective that last read this send buffer
        send[:, :7].copy_(poses.reshape(self.B, 7))
        send[:, 7].copy_(status.reshape(self.B))
        packed = torch.cuda.Event()
        packed.record(cur)
        with torch.cuda.stream(self._side):
            self._side.wait_event(packed)
            if not dist.is_initialized():
                recv[0].copy_(send)
            else:
                dist.all_gather_into_tensor(recv.view(-1), send.view(-1))
            self._done[k] = torch.cuda.Event()
            self._done[k].record(self._side)
        self._last = recv
        return recv[:, :, :7], recv[:, :, 7]

    def wait(self):
        """Join the side stream: everything gathered so far is visible to the current stream afterwards."""
        if hasattr(self, "_side"):
            torch.cuda.current_stream(self.send.device).wait_stream(self._side)


def gather_ragged(poses, status, n_total, force=False):
    """Gather block-partitioned results of uneven shards back into scan order on every rank (force: also in a one-rank group)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1 and not (force and dist.is_initialized()):
        return poses, status
    dev = poses.device
    cap = (n_total + world - 1) // world
    pad_p = torch.zeros((cap, 7), dtype=torch.float64, device=dev); pad_p[:len(poses)] = poses
    pad_s = torch.zeros((cap,), dtype=torch.int32, device=dev); pad_s[:len(status)] = status
    all_p = torch.empty((world, cap, 7), dtype=torch.float64, device=dev)
    all_s = torch.empty((world, cap), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(all_p.view(-1), pad_p.view(-1))
    dist.all_gather_into_tensor(all_s.view(-1), pad_s.view(-1))
    out_p, out_s = [], []
    for r in range(world):
        lo, hi = shard_bounds(n_total, r, world)
        out_p.append(all_p[r, :hi - lo]); out_s.append(all_s[r, :hi - lo])
    return torch.cat(out_p), torch.cat(out_s)


def register_sharded(register_fn, corner, corner_off, surf, surf_off, guesses):
    """Shard B scans over the ranks, run `register_fn(corner, corner_off, surf, surf_off, guesses)
    -> (poses (b,7), status (b,))` on the local block and gather everything in scan order.
    `register_fn` is the device call (capi.Handle.match_scan2map_batch) in production."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    B = len(guesses)
    lo, hi = shard_bounds(B, rank, world)
    co, (c0, c1) = shard_offsets(corner_off, lo, hi)
    so, (s0, s1) = shard_offsets(surf_off, lo, hi)
    poses, status = register_fn(corner[c0:c1], co, surf[s0:s1], so, guesses[lo:hi])
    p = torch.as_tensor(np.asarray(poses, dtype=np.float64).reshape(-1, 7))
    s = torch.as_tensor(np.asarray(status, dtype=np.int32).reshape(-1))
    gp, gs = gather_ragged(p, s, B)
    return gp.numpy(), gs.numpy()


def register_pairs_sharded(register_fn, map_corner, map_corner_off, map_surf, map_surf_off, corner, corner_off, surf, surf_off, guesses):
    """"Many map-submap pairs" across GPUs: the P pairs are block-partitioned over the ranks, every rank indexes and
    registers ITS pairs only -- nothing is broadcast, the maps never leave the rank that owns them -- and the poses are gathered in pair
    order.  `register_fn(map_corner, mc_off, map_surf, ms_off, corner, c_off, surf, s_off, guesses) -> (poses (p,7), status (p,))` is
    capi.Handle.match_pairs_batch in production."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    P = len(guesses)
    lo, hi = shard_bounds(P, rank, world)
    mco, (a0, a1) = shard_offsets(map_corner_off, lo, hi)
    mso, (b0, b1) = shard_offsets(map_surf_off, lo, hi)
    co, (c0, c1) = shard_offsets(corner_off, lo, hi)
    so, (s0, s1) = shard_offsets(surf_off, lo, hi)
    poses, status = register_fn(map_corner[a0:a1], mco, map_surf[b0:b1], mso, corner[c0:c1], co, surf[s0:s1], so, guesses[lo:hi])[:2]
    p = torch.as_tensor(np.asarray(poses, dtype=np.float64).reshape(-1, 7))
    s = torch.as_tensor(np.asarray(status, dtype=np.int32).reshape(-1))
    gp, gs = gather_ragged(p, s, P)
    return gp.numpy(), gs.numpy()
