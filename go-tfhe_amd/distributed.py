"""Batch sharding across the GPUs of one node.

Gate bootstraps are independent units over a read-only cloud key (trgsw.go:234-252 is already
one goroutine per input), so multi-GPU is: replicate the key on every GPU, split the batch into
contiguous index ranges, run the single-GPU path on each shard, gather in index order.  The
only collectives are the key broadcast (once) and the batch scatter and gather (RCCL over xGMI with
backend "nccl"; "gloo" in the CPU tests) -- there is no exchange step inside the path, so nothing is
reduced.  Dependent workloads (circuits) are sharded by circuit, so carries never leave their GPU.
"""
import time

import numpy as np

from ._binding import OPS

OP_NAMES = {v: k for k, v in OPS.items()}


def shard_bounds(total, world, rank):
    """Contiguous range [lo, hi) of rank's shard: [g*B/G, (g+1)*B/G)."""
    return (total * rank) // world, (total * (rank + 1)) // world


def shard_sizes(total, world):
    return [shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world)]


def _bcast_header(dist, words, root, group, device):
    """The few integers every rank needs before a scatter (batch size, plane count, uniform op code ...) as ONE fixed
    4-word int32 tensor broadcast on the backend's own device -- not broadcast_object_list, which pickles, moves the
    bytes through two collectives and synchronises the host on every call."""
    import torch
    h = torch.zeros(4, dtype=torch.int32, device=device)
    if words is not None:
        h[: len(words)] = torch.tensor(list(words), dtype=torch.int32)
    dist.broadcast(h, src=root, group=group)
    return [int(x) for x in h.tolist()]


def broadcast_cloud_key(ctx, src=0, group=None, via_host=False):
    """Replicate the cloud key loaded (or generated) in rank `src`'s context into every other rank's context:
    one broadcast per key of the engine's device-layout blob (tfhe_key_export_dev / tfhe_key_import_dev; 68.8 MB +
    77.9 MB at 128-bit) over xGMI, instead of every rank uploading or regenerating it.  The receiving
    library checks each blob's header (parameter set, key kind, layout version, length) before installing it.
    via_host: stage the blobs through host memory (backends that move CPU tensors: the gloo dry runs and CPU tests)."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    import torch
    for which in (0, 1):
        if via_host:
            if rank == src:
                blob = torch.from_numpy(ctx.key_export(which))
            else:
                blob = torch.empty(ctx.key_size(which), dtype=torch.uint8)
            dist.broadcast(blob, src=src, group=group)
            if rank != src:
                ctx.key_import(which, blob.numpy())
            continue
        if rank == src:
            blob = ctx.key_export_dev(which, torch.cuda.current_stream())
        else:
            blob = torch.empty(ctx.key_size(which), dtype=torch.uint8, device=torch.device("cuda", ctx.device))
        dist.broadcast(blob, src=src, group=group)
        if rank != src:
            ctx.key_import_dev(which, blob, torch.cuda.current_stream())
    if not via_host:
        torch.cuda.current_stream().synchronize()


class ShardedGates:
    """gates.Batch* over a process group: rank `root` holds the full batch, every rank computes
    its shard with `compute(ops, a, b, c) -> out` (the local single-GPU path), root gets the
    outputs back in order.  Tensors are torch tensors on the backend's device (GPU for nccl).

    One scatter and one gather per call: root packs the operands into ONE contiguous buffer laid out
    [rank][plane][row][n+1] (planes: a, b[, c]; + one int32 per row carrying the op code when ops are per item),
    so every rank receives its planes contiguous and ready for the kernels, without per-rank temporaries on root.
    `last_timing` holds the wall time of the three phases of the most recent call (seconds, this rank)."""

    def __init__(self, compute, n_plus_1, group=None, device="cpu"):
        import torch.distributed as dist
        self.dist = dist
        self.compute = compute
        self.n1 = n_plus_1
        self.group = group
        self.device = device
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.last_timing = None

    def _sync(self):
        import torch
        if torch.device(self.device).type == "cuda":
            torch.cuda.synchronize()

    def pack(self, ops, a, b, c=None):
        """Root-side packing (separate so that a caller can keep it out of a timed region when the batch is produced
        in this layout in the first place).  Returns (buffer, meta)."""
        import torch
        total, n1, W = a.shape[0], self.n1, self.world
        sizes = shard_sizes(total, W)
        cap = max(max(sizes), 1)
        planes = 2 + (c is not None)
        per_item = not isinstance(ops, str)
        stride = planes * cap * n1 + (cap if per_item else 0)          # int32 words per rank
        buf = torch.zeros((W, stride), dtype=torch.int32, device=self.device)
        even = total == cap * W
        for pl, src in enumerate((a, b, c)[:planes]):
            if even:                                                    # one strided copy per plane
                buf[:, pl * cap * n1:(pl + 1) * cap * n1].unflatten(1, (cap, n1)).copy_(src.view(W, cap, n1))
            else:
                for r in range(W):
                    lo, hi = shard_bounds(total, W, r)
                    buf[r, pl * cap * n1: pl * cap * n1 + (hi - lo) * n1].view(hi - lo, n1).copy_(src[lo:hi])
        if per_item:
            o32 = ops.to(torch.int32)
            for r in range(W):
                lo, hi = shard_bounds(total, W, r)
                buf[r, planes * cap * n1: planes * cap * n1 + (hi - lo)] = o32[lo:hi]
        return buf, [total, planes, -1 if per_item else OPS[ops]]

    def gate_batch(self, ops, a, b, c=None, total=None, root=0, packed=None):
        """ops: str (uniform) or uint8 tensor [B] on root; a, b, c: int32/uint32-bit tensors [B][n+1]
        on root (None elsewhere).  packed: result of pack() (root) to skip the packing step."""
        import torch
        dist, W, n1 = self.dist, self.world, self.n1
        t0 = time.perf_counter()
        if self.rank == root:
            buf, meta = packed if packed is not None else self.pack(ops, a, b, c)
        else:
            buf, meta = None, None
        total, planes, ucode, _ = _bcast_header(dist, meta, root, self.group, self.device)
        uniform = None if ucode < 0 else OP_NAMES[ucode]
        sizes = shard_sizes(total, W)
        cap = max(max(sizes), 1)
        mine_n = sizes[self.rank]
        stride = planes * cap * n1 + (cap if uniform is None else 0)
        mine = torch.empty(stride, dtype=torch.int32, device=self.device)
        dist.scatter(mine, list(buf.unbind(0)) if self.rank == root else None, src=root, group=self.group)
        self._sync()
        t1 = time.perf_counter()
        pl = [mine[k * cap * n1:(k + 1) * cap * n1].view(cap, n1)[:mine_n] for k in range(planes)]
        lops = uniform if uniform is not None else mine[planes * cap * n1: planes * cap * n1 + mine_n].to(torch.uint8)
        out = torch.zeros((cap, n1), dtype=torch.int32, device=self.device)
        if mine_n:
            out[:mine_n] = self.compute(lops, pl[0], pl[1], pl[2] if planes == 3 else None)
        self._sync()
        t2 = time.perf_counter()
        gathered = torch.empty((W, cap, n1), dtype=torch.int32, device=self.device) if self.rank == root else None
        dist.gather(out, list(gathered.unbind(0)) if self.rank == root else None, dst=root, group=self.group)
        self._sync()
        t3 = time.perf_counter()
        self.last_timing = {"scatter_s": t1 - t0, "compute_s": t2 - t1, "gather_s": t3 - t2}
        if self.rank != root:
            return None
        if total == cap * W:
            return gathered.view(total, n1)
        return torch.cat([gathered[r, : sizes[r]] for r in range(W)], dim=0)


class ShardedCircuits:
    """A levelised circuit over C independent instances, sharded BY CIRCUIT (carries stay on the
    GPU that owns the circuit, no traffic between levels): root scatters the input wires of each rank's
    circuits, every rank runs the whole circuit on its share (`run_local(wires)`, e.g. CircuitExecutor.run), root
    gathers the requested output wires."""

    def __init__(self, run_local, n_wires, n_plus_1, group=None, device="cpu"):
        import torch.distributed as dist
        self.dist, self.run_local, self.n_wires, self.n1 = dist, run_local, n_wires, n_plus_1
        self.group, self.device = group, device
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.last_timing = None

    def _sync(self):
        import torch
        if torch.device(self.device).type == "cuda":
            torch.cuda.synchronize()

    def run(self, in_wires, out_wires, inputs=None, root=0):
        """inputs (root): int32 tensor [len(in_wires)][C][n+1]; returns (root) [len(out_wires)][C][n+1].
        Any C: rank r owns the contiguous circuits shard_bounds(C, world, r) (shares differ by at most one; the
        scatter and gather buffers are padded to the largest share, a rank without circuits computes nothing)."""
        import torch
        dist, W, n1 = self.dist, self.world, self.n1
        t0 = time.perf_counter()
        C = _bcast_header(dist, [inputs.shape[1]] if self.rank == root else None, root, self.group, self.device)[0]
        sizes = shard_sizes(C, W)
        cap, I = max(max(sizes), 1), len(in_wires)
        mine_n = sizes[self.rank]
        mine = torch.empty((I, cap, n1), dtype=torch.int32, device=self.device)
        chunks = None
        if self.rank == root:
            if C == cap * W:                                  # [I][W][cap][n1] -> [W][I][cap][n1], one copy
                chunks = list(inputs.view(I, W, cap, n1).permute(1, 0, 2, 3).contiguous().unbind(0))
            else:
                pad = torch.zeros((W, I, cap, n1), dtype=torch.int32, device=self.device)
                for r in range(W):
                    lo, hi = shard_bounds(C, W, r)
                    pad[r, :, : hi - lo] = inputs[:, lo:hi]
                chunks = list(pad.unbind(0))
        dist.scatter(mine, chunks, src=root, group=self.group)
        self._sync()
        t1 = time.perf_counter()
        O = len(out_wires)
        res = torch.zeros((O, cap, n1), dtype=torch.int32, device=self.device)
        if mine_n:
            wires = torch.zeros((self.n_wires, mine_n, n1), dtype=torch.int32, device=self.device)
            wires[torch.tensor(list(in_wires), device=self.device)] = mine[:, :mine_n]
            ret = self.run_local(wires)              # in place, or returns the finished wire tensor
            if ret is not None:
                wires = ret
            res[:, :mine_n] = wires[torch.tensor(list(out_wires), device=self.device)]
        self._sync()
        t2 = time.perf_counter()
        gathered = torch.empty((W, O, cap, n1), dtype=torch.int32, device=self.device) if self.rank == root else None
        dist.gather(res, list(gathered.unbind(0)) if self.rank == root else None, dst=root, group=self.group)
        self._sync()
        t3 = time.perf_counter()
        self.last_timing = {"scatter_s": t1 - t0, "compute_s": t2 - t1, "gather_s": t3 - t2}
        if self.rank != root:
            return None
        if C == cap * W:
            return gathered.permute(1, 0, 2, 3).reshape(O, C, n1)
        return torch.cat([gathered[r, :, : sizes[r]] for r in range(W)], dim=1)


def gpu_compute(ctx):
    """Local compute callable over a Context using the device-pointer ABI (torch GPU tensors)."""
    import torch

    def run(ops, a, b, c):
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        c = c.contiguous() if c is not None else None
        o = ops if isinstance(ops, str) else ops.contiguous()
        ctx.gate_batch_dev(o, a, b, c, out, torch.cuda.current_stream())
        return out

    return run
