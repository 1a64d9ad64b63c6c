"""Sharding of an image batch over the GPUs of one node.

Images are independent (sift_anatomy keeps no state between calls), so a batch is cut into
contiguous blocks, one per rank, and processed without any collective on the data path.
The only cross-rank step is a host-side gather of the per-image keypoint arrays, in image
order, so that the result does not depend on the number of GPUs.
"""
import numpy as np


def shard_bounds(n_images, world, rank):
    """[begin, end) of the contiguous block of images owned by `rank`; blocks differ by at
    most one image and cover 0..n_images in rank order."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(n_images, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def gather_keypoints(local, dist=None, dst=0):
    """Gathers per-image keypoint arrays to rank `dst`.

    local: list of (image_index, structured array) owned by this rank.
    Returns, on rank dst, the list of arrays ordered by image index; None elsewhere.
    """
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [a for _, a in sorted(local, key=lambda t: t[0])]
    world = dist.get_world_size()
    bucket = [None] * world if dist.get_rank() == dst else None
    dist.gather_object([(i, a.tobytes(), a.dtype.descr) for i, a in local], bucket, dst=dst)
    if dist.get_rank() != dst:
        return None
    flat = [t for part in bucket for t in part]
    flat.sort(key=lambda t: t[0])
    return [np.frombuffer(b, dtype=np.dtype([tuple(f) for f in d])).copy() for _, b, d in flat]


def launch_plan(begin, end, batch):
    """Cuts the images [begin, end) of a fixed batch into launches of at most `batch` images that
    never straddle a multiple of `batch` (a launch reads a contiguous piece of a pool of `batch`
    resident images, image g being pool image g % batch).  Returns [(first image, count)]."""
    if batch < 1 or end < begin:
        raise ValueError("bad range/batch")
    plan, g = [], begin
    while g < end:
        n = min(batch - g % batch, end - g)
        plan.append((g, n))
        g += n
    return plan


def gather_counts(begin, counts, dist=None, dst=0):
    """Per-image integer results of contiguous shards -> one array in image order on rank `dst`
    (None elsewhere)."""
    counts = np.asarray(counts, np.int64)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return counts
    bucket = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object((int(begin), counts.tobytes()), bucket, dst=dst)
    if dist.get_rank() != dst:
        return None
    bucket.sort(key=lambda t: t[0])
    return np.concatenate([np.frombuffer(b, np.int64) for _, b in bucket])
