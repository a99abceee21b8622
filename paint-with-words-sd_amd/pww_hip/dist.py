"""Multi-GPU plumbing for the PwW path: one process per GPU, torch.distributed over RCCL/xGMI
(backend "nccl" on ROCm), gloo on CPU for tests.

The path shards by INDEPENDENT IMAGES: no collective sits on the data path.
Communication is (a) one broadcast of the model parameters at start-up, flattened into a single
buffer per dtype so it is one large RCCL call instead of ~700 small ones (xGMI is point-to-point and
per-link bound: few big transfers), (b) per request a broadcast of the raw color map + region table
(786 KB; every rank then runs the mask kernel itself instead of receiving 1.7 MB of weight maps), and
(c) an optional gather of the final latents (64 KB per image).
"""
import os
import pickle

import numpy as np
import torch
import torch.distributed as dist


def init_from_env(device_type="cuda"):
    """Initialise torch.distributed from torchrun's env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    Returns (rank, world_size, local_rank). World size 1 needs no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # under a launcher (RANK set) the group is created even for one rank, so a single-GPU torchrun
    # exercises the same RCCL calls as an 8-GPU one
    if (world > 1 or "RANK" in os.environ) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = "nccl" if device_type == "cuda" else "gloo"
        # smoke-test hook for boxes with ONE GPU: PWW_DIST_ONE_DEVICE=1 puts every rank on cuda:0 and rendezvous over gloo
        # (RCCL refuses two ranks on one device). Same code path above the backend; not a production mode.
        one_device = device_type == "cuda" and os.environ.get("PWW_DIST_ONE_DEVICE") == "1"
        if one_device:
            backend, local = "gloo", 0
        if device_type == "cuda":
            torch.cuda.set_device(local)
        if device_type == "cuda" and not one_device:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_items, rank, world):
    """Contiguous split of n_items over ranks (first n_items % world ranks get one more)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def image_seeds(base_seed, n_global, rank, world):
    """seed_i = base_seed + global image index: results do not depend on the GPU count."""
    lo, hi = shard_range(n_global, rank, world)
    return [base_seed + i for i in range(lo, hi)]


def broadcast_module(module, src=0, group=None):
    """Make every rank's parameters and buffers equal to rank `src`'s with ONE broadcast per dtype."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    total = 0
    for dtype, ts in sorted(by_dtype.items(), key=lambda kv: str(kv[0])):
        flat = torch.cat([t.reshape(-1) for t in ts])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for t in ts:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
        total += flat.numel() * flat.element_size()
    return total


def build_and_broadcast(build_fn, skeleton_fn, device, dtype, src=0, group=None):
    """Model set-up for image-sharded inference: rank `src` builds the real module (`build_fn()`: seeded
    random init or a checkpoint load), every other rank only allocates the structure (`skeleton_fn()` under
    the meta device, then `to_empty`) and receives the values through ONE flat broadcast per dtype.
    Returns (module, bytes_broadcast)."""
    rank = dist.get_rank(group) if (dist.is_available() and dist.is_initialized()) else 0
    if rank == src:
        module = build_fn().to(device=device, dtype=dtype)
    else:
        with torch.device("meta"):
            module = skeleton_fn()
        module = module.to_empty(device=device).to(dtype)
    module = module.eval().requires_grad_(False)
    nbytes = broadcast_module(module, src=src, group=group)
    return module, nbytes


def broadcast_request(payload, device, src=0, group=None):
    """Broadcast a request from `src`: every numpy array in the payload dict (the uint8 color map 'rgb', and for
    inpainting the mask and the init image) travels as a tensor broadcast, everything else (color_context, prompt ...)
    pickled in one byte tensor. payload on src: dict; None elsewhere. Returns the dict on every rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return payload
    rank = dist.get_rank(group)
    if rank == src:
        arrays = {k: np.ascontiguousarray(v) for k, v in payload.items() if isinstance(v, np.ndarray)}
        meta = {k: v for k, v in payload.items() if k not in arrays}
        meta["_arrays"] = [(k, tuple(a.shape), str(a.dtype)) for k, a in arrays.items()]
        blob = pickle.dumps(meta)
        head = torch.tensor([len(blob)], dtype=torch.int64, device=device)
    else:
        head = torch.zeros(1, dtype=torch.int64, device=device)
    dist.broadcast(head, src=src, group=group)
    if rank == src:
        meta_t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    else:
        meta_t = torch.empty(int(head.item()), dtype=torch.uint8, device=device)
    dist.broadcast(meta_t, src=src, group=group)
    meta = pickle.loads(meta_t.cpu().numpy().tobytes())
    for key, shape, dtype in meta.pop("_arrays"):
        if rank == src:
            t = torch.from_numpy(arrays[key]).to(device).contiguous()
        else:
            t = torch.empty(shape, dtype=getattr(torch, dtype), device=device)
        dist.broadcast(t, src=src, group=group)
        meta[key] = t.cpu().numpy()
    return meta


def gather_latents(latents, dst=0, group=None):
    """Gather every rank's final latents on `dst` (list in rank order there, None elsewhere)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [latents]
    world = dist.get_world_size(group)
    counts = [torch.zeros(1, dtype=torch.int64, device=latents.device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([latents.shape[0]], dtype=torch.int64, device=latents.device), group=group)
    nmax = int(max(c.item() for c in counts))
    pad = torch.zeros((nmax,) + tuple(latents.shape[1:]), dtype=latents.dtype, device=latents.device)
    pad[: latents.shape[0]] = latents
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    if dist.get_rank(group) != dst:
        return None
    return [o[: int(c.item())] for o, c in zip(out, counts)]


def max_over_ranks(value, device, group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def barrier(device=None, group=None):
    if dist.is_available() and dist.is_initialized():
        if device is not None and torch.device(device).type == "cuda":
            dist.barrier(group=group, device_ids=[torch.device(device).index or 0])
        else:
            dist.barrier(group=group)
