"""Multi-GPU harness: one process per GPU, `torch.distributed` over RCCL/xGMI.

The denoising path shards by independent units (prompt/seed index i -> rank i mod N, the
reference idiom `indices[args.rank::args.num_processes]`, videocrafter_main.py:181): there is
NO collective inside the denoising loop.  Two collectives exist in a job:
  C1  broadcast of the UNet parameters from rank 0 (each reference process loads the checkpoint
      itself, videocrafter_main.py:71-74; here one rank materialises them and the others receive
      them over xGMI in a few large flat buckets), and
  C2  a final gather of the result latents to rank 0.
On CPU test runs the same code runs on the `gloo` backend."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

BUCKET_BYTES = 1 << 30   # 1 GiB flat buckets: few, large messages (xGMI ring/tree is per-link bound)


def init_from_env(backend=None):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_indices(n_items, rank, world):
    """videocrafter_main.py:181 -- strided prompt sharding"""
    return list(range(n_items))[rank::world]


@torch.no_grad()
def broadcast_parameters(module, src=0, bucket_bytes=BUCKET_BYTES):
    """C1: flat-bucketed broadcast of every parameter/buffer of `module` from `src`."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    tensors = [p for p in module.parameters()] + [b for b in module.buffers()]
    sent = 0
    i = 0
    while i < len(tensors):
        dtype, dev = tensors[i].dtype, tensors[i].device
        group, nbytes = [], 0
        while i < len(tensors) and tensors[i].dtype == dtype and (not group or nbytes + tensors[i].numel() * tensors[i].element_size() <= bucket_bytes):
            group.append(tensors[i])
            nbytes += tensors[i].numel() * tensors[i].element_size()
            i += 1
        flat = torch.empty(sum(t.numel() for t in group), dtype=dtype, device=dev)
        if dist.get_rank() == src:
            off = 0
            for t in group:
                flat[off:off + t.numel()].copy_(t.reshape(-1))
                off += t.numel()
        dist.broadcast(flat, src=src)
        if dist.get_rank() != src:
            off = 0
            for t in group:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
        sent += nbytes
        del flat
    for m in module.modules():
        if hasattr(m, "_invalidate"):
            m._invalidate()
    return sent


@torch.no_grad()
def gather_results(t, dst=0):
    """C2: gather equally-shaped result tensors on `dst`; returns the list there, None elsewhere."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [t]
    world = dist.get_world_size()
    if dist.get_backend() == "nccl":
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t.contiguous())      # RCCL has no native gather-to-one faster than this at 0.3 MB
        return out if dist.get_rank() == dst else None
    out = [torch.empty_like(t) for _ in range(world)] if dist.get_rank() == dst else None
    dist.gather(t.contiguous(), out, dst=dst)
    return out


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(x: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shutdown():
    """tear the process group down (every rank, after its last collective; no barrier: a failed rank must not hang the rest)"""
    if dist.is_initialized():
        dist.destroy_process_group()
