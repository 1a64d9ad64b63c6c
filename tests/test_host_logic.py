"""Host-side logic that needs neither the reference nor a GPU: synthetic inputs, sharding
and the world_size-2 gather (gloo)."""
import os
import sys

import numpy as np
import pytest

from sift_amd import abi, shard, synth


def test_lcg_matches_scalar_recurrence():
    st, want = 777, []
    for _ in range(500):
        st = (st * 1664525 + 1013904223) & 0xFFFFFFFF
        want.append((st >> 8) / 2 ** 24)
    assert np.array_equal(np.array(want, np.float32), synth.lcg_uniform(777, 500))


def test_synthetic_images_are_deterministic_and_in_range():
    for kind in ("smooth", "white"):
        a = synth.make(kind, 160, 90, 5)
        b = synth.make(kind, 160, 90, 5)
        c = synth.make(kind, 160, 90, 6)
        assert a.dtype == np.float32 and a.shape == (90, 160)
        assert np.array_equal(a, b) and not np.array_equal(a, c)
        assert 0.0 <= a.min() and a.max() < 1.0


def test_shard_bounds_cover_batch_in_order():
    for n in (0, 1, 7, 16, 4096):
        for world in (1, 2, 3, 4, 8):
            spans = [shard.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[k][1] == spans[k + 1][0] for k in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, n_images, out_path):
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import pyoracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = shard.shard_bounds(n_images, world, rank)
    local = []
    for i in range(b, e):
        rec = pyoracle.run(synth.white(96, 72, seed=50 + i), keep_ss=False)
        k = rec["keys"]
        arr = np.zeros(len(k["ints"]), abi.KEYPOINT_STD_DTYPE)
        arr["x"], arr["y"] = k["flts"][:, 0], k["flts"][:, 1]
        arr["scale"], arr["orientation"] = k["flts"][:, 2], k["flts"][:, 3]
        arr["descriptor"] = k["descr"].astype(np.uint8)
        local.append((i, arr))
    got = shard.gather_keypoints(local, dist, dst=0)
    if rank == 0:
        np.save(out_path, np.concatenate(got).view(np.uint8))
        np.save(out_path + ".counts.npy", np.array([len(a) for a in got]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_gather_is_independent_of_world_size(tmp_path, world):
    """The per-image keypoints gathered from `world` ranks equal the single-process result."""
    import torch.multiprocessing as mp
    n_images = 5
    out = str(tmp_path / "gathered.npy")
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, n_images, out), nprocs=world, join=True)
    got = np.load(out)
    counts = np.load(out + ".counts.npy")
    # single process
    from oracle import pyoracle
    want = []
    for i in range(n_images):
        rec = pyoracle.run(synth.white(96, 72, seed=50 + i), keep_ss=False)
        k = rec["keys"]
        arr = np.zeros(len(k["ints"]), abi.KEYPOINT_STD_DTYPE)
        arr["x"], arr["y"] = k["flts"][:, 0], k["flts"][:, 1]
        arr["scale"], arr["orientation"] = k["flts"][:, 2], k["flts"][:, 3]
        arr["descriptor"] = k["descr"].astype(np.uint8)
        want.append(arr)
    assert counts.tolist() == [len(a) for a in want]
    assert np.array_equal(got, np.concatenate(want).view(np.uint8))


def test_launch_plan_covers_a_shard_without_straddling_the_pool():
    for begin, end, batch in ((0, 0, 8), (0, 4096, 128), (1365, 2731, 128), (5, 6, 4), (3, 29, 8), (512, 1024, 128)):
        plan = shard.launch_plan(begin, end, batch)
        covered = [g for g0, n in plan for g in range(g0, g0 + n)]
        assert covered == list(range(begin, end))
        assert all(1 <= n <= batch and g0 % batch + n <= batch for g0, n in plan)
    assert shard.launch_plan(0, 4096, 128) == [(k * 128, 128) for k in range(32)]


def _plan_worker(rank, world, port, total, batch, out_path):
    """The strong-scaling loop of bench.py with a stand-in for the detector: image g of the batch
    is pool image g % batch, and its result depends on the pool image only."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pool_result = (np.arange(batch) * 7919 + 13) % 1000
    lo, hi = shard.shard_bounds(total, world, rank)
    counts = np.zeros(hi - lo, np.int64)
    for g0, n in shard.launch_plan(lo, hi, batch):
        counts[g0 - lo:g0 - lo + n] = pool_result[g0 % batch:g0 % batch + n]
    got = shard.gather_counts(lo, counts, dist)
    if rank == 0:
        np.save(out_path, got)
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_strong_scaling_counts_are_independent_of_world_size(tmp_path, world):
    import torch.multiprocessing as mp
    total, batch = 101, 8
    out = str(tmp_path / "counts.npy")
    port = 31500 + (os.getpid() % 2000) + world
    mp.spawn(_plan_worker, args=(world, port, total, batch, out), nprocs=world, join=True)
    want = ((np.arange(total) % batch) * 7919 + 13) % 1000
    assert np.array_equal(np.load(out), want)
    assert np.array_equal(shard.gather_counts(0, want), want)       # single process: passthrough
