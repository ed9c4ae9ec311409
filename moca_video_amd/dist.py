"""Multi-GPU harness: one process per GPU, `torch.distributed` over RCCL/xGMI.

The denoising path shards by independent units (prompt/seed index i -> rank i mod N, the
reference idiom `indices[args.rank::args.num_processes]`, videocrafter_main.py:181): there is
NO collective inside the denoising loop.  Two collectives exist in a job:
  C1  broadcast of the UNet's PACKED operand set (2.83 GB fp16 weights + fp32 biases / norm parameters) from rank 0 (each
      reference process loads the checkpoint itself, videocrafter_main.py:71-74; here one rank materialises and packs them,
      the others receive them over xGMI in a few large flat buckets, in place, and never hold or pack fp32 masters), and
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
            # "nccl" IS RCCL on ROCm.  MOCA_DIST_BACKEND=gloo: rehearsal of an N-rank job on ONE GPU (RCCL refuses two ranks per device)
            backend = os.environ.get("MOCA_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_indices(n_items, rank, world):
    """videocrafter_main.py:181 -- strided prompt sharding"""
    return list(range(n_items))[rank::world]


@torch.no_grad()
def broadcast_parameters(module, src=0, bucket_bytes=BUCKET_BYTES):
    """flat-bucketed broadcast of every parameter/buffer of `module` from `src` (the fp32 masters: 5.65 GB for the UNet; the job's
    C1 is `broadcast_packed`, which moves the 2.83 GB the kernels read)"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    sent = broadcast_tensors([p for p in module.parameters()] + [b for b in module.buffers()], src, bucket_bytes)
    for m in module.modules():
        if hasattr(m, "_invalidate"):
            m._invalidate()
    return sent


def _walk_operands(v, pws, tensors):
    from .ops import PackedWeight
    if isinstance(v, PackedWeight):
        if not v.derived:              # (per-group weights of a folded GroupNorm are WRITTEN by a launch of the forward: not an operand to send)
            pws.setdefault(id(v), v)
    elif torch.is_tensor(v):
        tensors.append(v)
    elif isinstance(v, (tuple, list)):
        for u in v:
            _walk_operands(u, pws, tensors)


def packed_operands(packed, plans):
    """The packed operand set the recorded launches of `plans` read: the fp16 [N][K] weight matrices (plain, LayerNorm-folded, per-head
    q|k|v, `Upsample` phases -- whichever form each launch uses), their fp32 biases / folded row sums, and the fp32 norm parameters.
    Order = first use in the launch sequences, so every rank that built the same plans enumerates the same list (the dict keys of
    `packed` are process-local ids and play no role)."""
    pws, seen_t = {}, []
    for plan in plans:
        for st in plan.steps:
            for v in list(getattr(st, "args", ())) + list(getattr(st, "keywords", {}).values()):
                _walk_operands(v, pws, seen_t)
    norm_ptrs = {}
    for v in packed.values():                      # fp32 norm parameters live in `packed` as (gamma, beta) tuples
        if isinstance(v, tuple) and all(torch.is_tensor(t) for t in v):
            for t in v:
                norm_ptrs[t.data_ptr()] = t
    out, done = [], set()

    def add(t):
        if t is not None and t.data_ptr() not in done:
            done.add(t.data_ptr())
            out.append(t)
    for pw in pws.values():
        add(pw.w); add(pw.bias); add(pw.wsum)
    for t in seen_t:
        if t.data_ptr() in norm_ptrs:
            add(norm_ptrs[t.data_ptr()])
    return out


@torch.no_grad()
def broadcast_packed(model, plans, src=0, bucket_bytes=BUCKET_BYTES, drop_masters=True):
    """C1: rank `src` has packed its weights (`model._packed`); every rank has BUILT the same plans (the others from
    placeholder parameters: packing is shape-driven), so all ranks hold the same list of operand tensors.  Those are broadcast in
    a few large flat buckets -- 2.83 GB for the UNet instead of the 5.65 GB of fp32 masters, and no re-pack on the receivers (the
    recorded launches keep their pointers: the data arrives in place).  Receivers then drop their fp32 masters (`drop_masters`) and
    refuse to build further plans (they have no parameters to pack from).  Returns (bytes, tensors)."""
    tensors = packed_operands(model._packed, plans)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0, tensors
    sig = [(tuple(t.shape), str(t.dtype)) for t in tensors]
    ref = [sig if dist.get_rank() == src else None]
    dist.broadcast_object_list(ref, src=src)
    if ref[0] != sig:
        raise RuntimeError("broadcast_packed: this rank's packed operand list differs from rank %d's (different plans built?)" % src)
    sent = broadcast_tensors(tensors, src, bucket_bytes)
    if dist.get_rank() != src:
        model._packed_only = True
        if drop_masters:
            for p in model.parameters():
                p.data = torch.empty(0, dtype=p.dtype, device=p.device)
    return sent, tensors


@torch.no_grad()
def broadcast_tensors(tensors, src=0, bucket_bytes=BUCKET_BYTES):
    """flat-bucketed broadcast (buckets hold one dtype): returns the bytes moved"""
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
