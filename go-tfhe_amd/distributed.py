"""Batch sharding across the GPUs of one node.

Gate bootstraps are independent units over a read-only cloud key (trgsw.go:234-252 is already
one goroutine per input), so multi-GPU is: replicate the key on every GPU, split the batch into
contiguous index ranges, run the single-GPU path on each shard, gather in index order.  The
only collectives are the batch scatter and gather (RCCL over xGMI with backend "nccl"; "gloo"
in the CPU tests) -- there is no exchange step inside the path, so nothing is reduced.
"""
import numpy as np


def shard_bounds(total, world, rank):
    """Contiguous range [lo, hi) of rank's shard: [g*B/G, (g+1)*B/G)."""
    return (total * rank) // world, (total * (rank + 1)) // world


def shard_sizes(total, world):
    return [shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world)]


class ShardedGates:
    """gates.Batch* over a process group: rank `root` holds the full batch, every rank computes
    its shard with `compute(ops, a, b, c) -> out` (the local single-GPU path), root gets the
    outputs back in order.  Tensors are torch tensors on the backend's device (GPU for nccl)."""

    def __init__(self, compute, n_plus_1, group=None, device="cpu"):
        import torch.distributed as dist
        self.dist = dist
        self.compute = compute
        self.n1 = n_plus_1
        self.group = group
        self.device = device
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def _scatter(self, full, total, width, dtype, root):
        """Scatter rows of `full` (root only) in shard order; shards are padded to equal length."""
        import torch
        sizes = shard_sizes(total, self.world)
        cap = max(max(sizes), 1)
        mine = torch.empty((cap, width), dtype=dtype, device=self.device)
        chunks = None
        if self.rank == root:
            chunks = []
            for r in range(self.world):
                lo, hi = shard_bounds(total, self.world, r)
                buf = torch.zeros((cap, width), dtype=dtype, device=self.device)
                buf[: hi - lo] = full[lo:hi]
                chunks.append(buf)
        self.dist.scatter(mine, chunks, src=root, group=self.group)
        return mine[: sizes[self.rank]]

    def gate_batch(self, ops, a, b, c=None, total=None, root=0):
        """ops: str (uniform) or uint8 tensor [B] on root; a, b, c: int32/uint32-bit tensors [B][n+1]
        on root (None elsewhere).  `total` (batch size) must be given on non-root ranks."""
        import torch
        if self.rank == root:
            total = a.shape[0]
        meta = [total, c is not None, ops if isinstance(ops, str) else None] if self.rank == root else [None] * 3
        self.dist.broadcast_object_list(meta, src=root, group=self.group)
        total, has_c, uniform = meta
        dt = torch.int32
        la = self._scatter(a, total, self.n1, dt, root)
        lb = self._scatter(b, total, self.n1, dt, root)
        lc = self._scatter(c, total, self.n1, dt, root) if has_c else None
        lops = uniform
        if uniform is None:
            lops = self._scatter(ops.reshape(-1, 1) if self.rank == root else None, total, 1, torch.uint8, root).reshape(-1)
        lout = self.compute(lops, la, lb, lc) if la.shape[0] else la.clone()
        # gather (padded) and re-assemble in index order
        sizes = shard_sizes(total, self.world)
        cap = max(max(sizes), 1)
        pad = torch.zeros((cap, self.n1), dtype=dt, device=self.device)
        pad[: lout.shape[0]] = lout
        bufs = [torch.empty_like(pad) for _ in range(self.world)] if self.rank == root else None
        self.dist.gather(pad, bufs, dst=root, group=self.group)
        if self.rank != root:
            return None
        return torch.cat([bufs[r][: sizes[r]] for r in range(self.world)], dim=0)


def gpu_compute(ctx):
    """Local compute callable over a Context using the device-pointer ABI (torch GPU tensors)."""
    import torch

    def run(ops, a, b, c):
        out = torch.empty_like(a)
        a, b = a.contiguous(), b.contiguous()
        c = c.contiguous() if c is not None else None
        o = ops if isinstance(ops, str) else ops.contiguous()
        ctx.gate_batch_dev(o, a, b, c, out, torch.cuda.current_stream())
        return out

    return run
