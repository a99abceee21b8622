"""N > 1 path on CPU: world_size-2 gloo processes exercise the sharding / broadcast / gather plumbing
of pww_hip.dist (the data path itself has no collective: images are independent)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "paint-with-words-sd_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from pww_hip import dist as pdist
    from sd_standin import build_unet, TINY_CONFIG
    r, w, _ = pdist.init_from_env("cpu")
    assert (r, w) == (rank, world)
    dev = torch.device("cpu")
    # (a) weights: rank 0 has the seeded model, the others garbage; one flat broadcast per dtype
    unet = build_unet(TINY_CONFIG, seed=1234 if rank == 0 else 999)
    nbytes = pdist.broadcast_module(unet, src=0)
    ref = build_unet(TINY_CONFIG, seed=1234)
    assert nbytes == sum(p.numel() * p.element_size() for p in ref.parameters())
    for a, b in zip(unet.parameters(), ref.parameters()):
        assert torch.equal(a, b)
    # (a'') the same through many small buckets (64 KiB: several tensors per bucket, and tensors larger than a bucket on their own)
    unet_b = build_unet(TINY_CONFIG, seed=1234 if rank == 0 else 555)
    assert any(p.numel() * 4 > (1 << 16) for p in unet_b.parameters()) and any(p.numel() * 4 < (1 << 12) for p in unet_b.parameters())
    assert pdist.broadcast_module(unet_b, src=0, bucket_bytes=1 << 16) == nbytes
    for a, b in zip(unet_b.parameters(), ref.parameters()):
        assert torch.equal(a, b)
    # (a') what bench.py does: rank 0 builds, the others allocate a meta skeleton and receive the values
    from sd_standin import UNet2DConditionModel
    timing = {}
    m2, nb2 = pdist.build_and_broadcast(lambda: build_unet(TINY_CONFIG, seed=77), lambda: UNet2DConditionModel(**TINY_CONFIG),
                                        dev, torch.float32, src=0, timing=timing)
    assert set(timing) == {"build_s", "broadcast_s"}
    assert os.environ.get("MIOPEN_USER_DB_PATH", "").endswith("pww_rank%d" % rank)      # one MIOpen user db per local rank
    ref2 = build_unet(TINY_CONFIG, seed=77)
    assert nb2 == nbytes and not any(p.is_meta for p in m2.parameters())
    for a, b in zip(m2.parameters(), ref2.parameters()):
        assert torch.equal(a, b)
    x = torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(1))
    assert torch.equal(m2(x, torch.tensor(10.0), torch.zeros(1, 77, 64)).sample, ref2(x, torch.tensor(10.0), torch.zeros(1, 77, 64)).sample)
    # (b) request broadcast: color map + region table
    payload = None
    rgb = (np.arange(64 * 48 * 3) % 251).astype(np.uint8).reshape(64, 48, 3)
    if rank == 0:
        payload = {"rgb": rgb, "context": {(1, 2, 3): "dog,1.0"}, "prompt": "a dog"}
    got = pdist.broadcast_request(payload, dev, src=0)
    assert got["prompt"] == "a dog" and got["context"] == {(1, 2, 3): "dog,1.0"}
    assert got["rgb"].shape == (64, 48, 3) and int(got["rgb"].sum()) == int(((np.arange(64 * 48 * 3) % 251).astype(np.uint8)).sum())
    # (b') an inpainting request: color map + mask + init image travel as arrays, the rest pickled (bench.py --config 4)
    payload = None
    if rank == 0:
        payload = {"rgb": rgb, "mask": (np.arange(64 * 48) % 256).astype(np.uint8).reshape(64, 48), "init": rgb[::-1].copy(),
                   "context": {(1, 2, 3): "dog,1.0,7"}, "prompt": "a dog", "seeds": [81, 82]}
    got = pdist.broadcast_request(payload, dev, src=0)
    assert set(got) == {"rgb", "mask", "init", "context", "prompt", "seeds"} and got["seeds"] == [81, 82]
    assert got["mask"].dtype == np.uint8 and got["mask"].shape == (64, 48) and int(got["mask"][1, 5]) == (48 + 5) % 256
    assert got["init"].shape == (64, 48, 3) and np.array_equal(got["init"][::-1], got["rgb"])
    # (c) image sharding is contiguous, disjoint, complete, and seeds do not depend on the world size
    seeds = pdist.image_seeds(100, 5, rank, world)
    lat = torch.stack([torch.full((4, 2, 2), float(s)) for s in seeds]) if seeds else torch.zeros(0, 4, 2, 2)
    gathered = pdist.gather_latents(lat, dst=0)
    t = pdist.max_over_ranks(1.0 + rank, dev)
    assert t == float(world)
    assert pdist.all_ranks(10.0 + rank, dev) == [10.0 + r for r in range(world)]        # per-rank timings of a bench line (bench.py config.per_rank)
    if rank == 0:
        allv = torch.cat(gathered)[:, 0, 0, 0].tolist()
        assert allv == [100.0, 101.0, 102.0, 103.0, 104.0]
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    pdist.barrier()
    dist.destroy_process_group()


def test_gloo_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").read_text() == "ok"


def test_shard_range_properties():
    from pww_hip.dist import shard_range, image_seeds
    for n in (0, 1, 7, 8, 64):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    assert sum((image_seeds(10, 64, r, 8) for r in range(8)), []) == list(range(10, 74))
