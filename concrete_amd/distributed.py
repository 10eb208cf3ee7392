"""Multi-GPU plumbing: one process per GPU, ciphertext batches sharded by index, the server
key replicated once. There is no collective in the steady state: every bootstrap reads only the
shared read-only key material and its own ciphertext (the reference exploits the same
independence with rayon, concrete-integer/src/server_key/radix_parallel/mul.rs:157-181).

torch.distributed supplies the process group (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in CPU tests); the key travels as one byte blob (Fourier bootstrap key + keyswitch key).
"""
from __future__ import annotations

from typing import Tuple


def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous index range [begin, end) of ``rank``: sizes differ by at most one."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(int(total), int(world))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def broadcast_blob(blob, meta, src: int = 0):
    """Broadcast the key blob. ``meta`` is an int64 tensor [blob_bytes, variant] valid on ``src``;
    ``blob`` is a uint8 tensor on ``src`` and None elsewhere. Returns (blob, meta) on every rank."""
    import torch
    import torch.distributed as dist

    dist.broadcast(meta, src)
    if dist.get_rank() != src:
        blob = torch.empty(int(meta[0].item()), dtype=torch.uint8, device=meta.device)
    dist.broadcast(blob, src)
    return blob, meta


def replicate_server_key(ctx, params, device_key, torus_bits: int = 64, src: int = 0):
    """Give every rank a DeviceServerKey equal to ``device_key`` of rank ``src`` (None elsewhere).

    One broadcast of the device-resident blob; the forward FFT of the bootstrap key ran once, on
    ``src``. Replaces the per-device key conversion of
    concrete-boolean/src/engine/bootstrapping/cuda.rs:123-137 for the multi-GPU case."""
    import torch
    import torch.distributed as dist

    from .backend import DeviceServerKey

    if dist.get_backend() != "nccl":
        # no device-to-device transport (gloo: ranks that share a GPU, CPU rehearsals): the blob
        # travels through host memory, still as one broadcast
        meta = torch.zeros(2, dtype=torch.int64)
        blob = None
        if dist.get_rank() == src:
            host = device_key.export_blob_host()
            meta[0], meta[1] = host.size, device_key.variant
            blob = torch.from_numpy(host)
        blob, meta = broadcast_blob(blob, meta, src)
        if dist.get_rank() == src:
            return device_key
        return DeviceServerKey.from_host_blob(ctx, params, blob.numpy(), int(meta[1].item()), torus_bits=torus_bits)

    device = torch.device("cuda", ctx.device_id)
    meta = torch.zeros(2, dtype=torch.int64, device=device)
    blob = None
    if dist.get_rank() == src:
        meta[0], meta[1] = device_key.blob_size(), device_key.variant
        blob = torch.empty(device_key.blob_size(), dtype=torch.uint8, device=device)
        device_key.export_blob(blob, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    blob, meta = broadcast_blob(blob, meta, src)
    torch.cuda.synchronize()
    if dist.get_rank() == src:
        return device_key
    return DeviceServerKey.from_device_blob(ctx, params, blob, int(meta[1].item()), torus_bits=torus_bits)
