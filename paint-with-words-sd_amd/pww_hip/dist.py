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
        if world > 1:
            # N ranks of one node would otherwise run MIOpen's find mode against ONE shared user perf-db file (file-lock
            # contention, N x the same search racing on the same records): one user db per local rank. Must be in the environment
            # before the process's first MIOpen call, i.e. here.
            if "MIOPEN_USER_DB_PATH" not in os.environ:
                base = os.environ.get("PWW_MIOPEN_DB_BASE", os.path.join(os.path.expanduser("~"), ".config", "miopen"))
                path = os.path.join(base, "pww_rank%d" % local)
                if not os.path.isdir(path):
                    # first use: seed the rank's db with the user's existing one (a tuned perf-db keeps paying off: no rank
                    # repeats a find search whose answer is already on disk)
                    os.makedirs(path, exist_ok=True)
                    import shutil
                    for name in (os.listdir(base) if os.path.isdir(base) else []):
                        src_file = os.path.join(base, name)
                        if os.path.isfile(src_file):
                            try:
                                shutil.copy2(src_file, os.path.join(path, name))
                            except OSError:
                                pass
                os.environ["MIOPEN_USER_DB_PATH"] = path
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


def miopen_db_path(local_rank):
    """The MIOpen user-db directory init_from_env gives local rank `local_rank` of a multi-rank job (None when the user set
    MIOPEN_USER_DB_PATH themselves: every rank then shares that one)."""
    own = os.environ.get("MIOPEN_USER_DB_PATH")
    if own and os.path.basename(os.path.normpath(own)).startswith("pww_rank"):
        return os.path.join(os.path.dirname(os.path.normpath(own)), "pww_rank%d" % local_rank)
    return None


def adopt_miopen_db(src_local_rank=0):
    """Copy local rank `src_local_rank`'s MIOpen user db (find-db and perf-db records of its warm-up) into this rank's own directory: the
    convolutions of this rank's warm-up then find their answers on disk instead of repeating the find-mode search (~25 s for the SD1.5
    UNet) on every rank of the node. Call it BEFORE this process's first convolution and after a barrier behind the source rank's warm-up.
    Same node only (a shared file system is assumed); returns the number of files copied (0: nothing to adopt, the rank searches itself)."""
    import shutil
    src, dst = miopen_db_path(src_local_rank), os.environ.get("MIOPEN_USER_DB_PATH")
    if not src or not dst or not os.path.isdir(src) or os.path.normpath(src) == os.path.normpath(dst):
        return 0
    n = 0
    for name in os.listdir(src):
        f = os.path.join(src, name)
        if os.path.isfile(f) and not name.endswith(".lock"):
            try:
                shutil.copy2(f, os.path.join(dst, name))
                n += 1
            except OSError:
                pass
    return n


def backend_name():
    """"nccl" (= RCCL on ROCm) / "gloo" of the default process group, or "none" without one."""
    return dist.get_backend() if (dist.is_available() and dist.is_initialized()) else "none"


def shard_range(n_items, rank, world):
    """Contiguous split of n_items over ranks (first n_items % world ranks get one more)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def image_seeds(base_seed, n_global, rank, world):
    """seed_i = base_seed + global image index: results do not depend on the GPU count."""
    lo, hi = shard_range(n_global, rank, world)
    return [base_seed + i for i in range(lo, hi)]


BROADCAST_BUCKET_BYTES = 256 << 20     # per collective: large enough for xGMI's per-link rate, small enough to bound the staging buffer


def broadcast_module(module, src=0, group=None, bucket_bytes=None):
    """Make every rank's parameters and buffers equal to rank `src`'s with a few LARGE broadcasts: tensors are packed per dtype
    into one reusable staging buffer of at most `bucket_bytes` (default 256 MiB), broadcast, and unpacked -- 7 RCCL calls for the
    1.72 GB SD1.5 UNet instead of ~700 small ones, without the transient second copy of all weights a single flat
    `torch.cat` needs. A tensor larger than the bucket travels on its own. Returns the bytes broadcast."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    bucket_bytes = bucket_bytes or BROADCAST_BUCKET_BYTES
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    rank = dist.get_rank()       # GLOBAL rank: `src` of dist.broadcast is a global rank too (with a sub-group the group-local rank differs)
    total = 0
    for dtype, ts in sorted(by_dtype.items(), key=lambda kv: str(kv[0])):
        cap = max(1, bucket_bytes // ts[0].element_size())
        stage = None
        i = 0
        while i < len(ts):
            if ts[i].numel() >= cap:            # bigger than a bucket: in place if contiguous
                t = ts[i]
                flat = t.reshape(-1) if t.is_contiguous() else t.contiguous().reshape(-1)
                dist.broadcast(flat, src=src, group=group)
                if not t.is_contiguous():
                    t.copy_(flat.view_as(t))
                total += flat.numel() * flat.element_size()
                i += 1
                continue
            j, n = i, 0
            while j < len(ts) and ts[j].numel() < cap and n + ts[j].numel() <= cap:
                n += ts[j].numel()
                j += 1
            if stage is None:
                stage = torch.empty(min(cap, sum(t.numel() for t in ts)), dtype=dtype, device=ts[0].device)
            buf = stage[:n]
            if rank == src:
                off = 0
                for t in ts[i:j]:
                    buf[off:off + t.numel()].copy_(t.reshape(-1))
                    off += t.numel()
            dist.broadcast(buf, src=src, group=group)
            if rank != src:
                off = 0
                for t in ts[i:j]:
                    t.copy_(buf[off:off + t.numel()].view_as(t))
                    off += t.numel()
            total += n * buf.element_size()
            i = j
    return total


def build_and_broadcast(build_fn, skeleton_fn, device, dtype, src=0, group=None, timing=None):
    """Model set-up for image-sharded inference: rank `src` builds the real module (`build_fn()`: seeded
    random init or a checkpoint load), every other rank only allocates the structure (`skeleton_fn()` under
    the meta device, then `to_empty`) and receives the values through ONE flat broadcast per dtype.
    Returns (module, bytes_broadcast); `timing` (a dict, optional) receives build_s / broadcast_s."""
    import time
    rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0      # global, like `src`
    t0 = time.time()
    if rank == src:
        module = build_fn().to(device=device, dtype=dtype)
    else:
        with torch.device("meta"):
            module = skeleton_fn()
        module = module.to_empty(device=device).to(dtype)
    module = module.eval().requires_grad_(False)
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize()
    barrier(device, group)            # the broadcast time below is the collective's, not rank 0's model construction
    t1 = time.time()
    nbytes = broadcast_module(module, src=src, group=group)
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize()
    if timing is not None:
        timing.update(build_s=round(t1 - t0, 3), broadcast_s=round(time.time() - t1, 3))
    return module, nbytes


def broadcast_request(payload, device, src=0, group=None):
    """Broadcast a request from `src`: every numpy array in the payload dict (the uint8 color map 'rgb', and for
    inpainting the mask and the init image) travels as a tensor broadcast, everything else (color_context, prompt ...)
    pickled in one byte tensor. payload on src: dict; None elsewhere. Returns the dict on every rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return payload
    rank = dist.get_rank()       # global, like `src`
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
    if dist.get_rank() != dst:      # (`dst`: a global rank)
        return None
    return [o[: int(c.item())] for o, c in zip(out, counts)]


def max_over_ranks(value, device, group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def all_ranks(value, device, group=None):
    """[value of rank 0, rank 1, ...] on every rank (one all_gather of a float64): per-rank timings of a benchmark line, so that a slow
    rank is visible in the record instead of hiding behind the max."""
    if not (dist.is_available() and dist.is_initialized()):
        return [float(value)]
    world = dist.get_world_size(group)
    out = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(world)]
    dist.all_gather(out, torch.tensor([float(value)], dtype=torch.float64, device=device), group=group)
    return [float(t.item()) for t in out]


def barrier(device=None, group=None):
    if dist.is_available() and dist.is_initialized():
        if device is not None and torch.device(device).type == "cuda":
            dist.barrier(group=group, device_ids=[torch.device(device).index or 0])
        else:
            dist.barrier(group=group)
