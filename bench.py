#!/usr/bin/env python3
"""Throughput benchmark of the SIFT hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the whole pipeline (scale-space, DoG, extrema, refinement,
orientation, descriptors, host gather of the keypoints) over one batch of synthetic
1920x1080 images per GPU; the images are resident in HBM before the timed region starts.
For N > 1 the driver launches one process per GPU with torch.distributed.run; images are
sharded by rank with no collective on the data path (weak scaling: every GPU processes
the same number of images per step).  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W_IM, H_IM = 1920, 1080
HBM_PEAK_GBS = 8000.0


def cpu_baseline(n_cores):
    """Times the CPU implementation on the host cores, one process per core, one 1080p image
    each.  Uses the compiled reference (oracle/_ref) when present, else the oracle port.
    Must run before this process touches the GPU (workers are forked)."""
    import multiprocessing as mp
    from oracle import refrun, pyoracle
    kind = "reference" if refrun.have_ref() else "port"
    if kind == "port":
        pyoracle.build()
    ctx = mp.get_context("fork")
    t0 = time.time()
    with ctx.Pool(n_cores) as pool:
        nkeys = pool.map(_cpu_one, [(kind, 12345 + b) for b in range(n_cores)])
    dt = time.time() - t0
    return {"value": n_cores / dt, "unit": "images/s", "cores": n_cores, "kind": kind,
            "sample": f"{n_cores} synthetic 1920x1080 S1 images (seeds 12345..{12345 + n_cores - 1}), "
                      f"one process per core, sift_anatomy with default parameters, "
                      f"{dt:.1f} s wall, image generation excluded from neither side",
            "keypoints": int(sum(nkeys))}


def _cpu_one(arg):
    kind, seed = arg
    from sift_amd import synth
    img = synth.smooth(W_IM, H_IM, seed)
    if kind == "reference":
        from oracle import refrun
        rec = refrun.run_sift_anatomy(refrun.ref_lib(), img, keep_ss=False)
    else:
        from oracle import pyoracle
        rec = pyoracle.run(img, keep_ss=False)
    return len(rec["keys"]["ints"])


def load_traffic():
    """HBM bytes per launch of the dominant kernel from the PMC passes (profiles/), if any."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(path):
        try:
            with open(path) as f:
                return json.load(f)
        except Exception:
            return None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--kind", default="smooth", choices=["smooth", "white"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--host-input", action="store_true",
                    help="diagnostic: images start in pinned host memory and are copied to the GPU "
                         "inside the timed region (PCIe-inclusive rate; not the headline value)")
    ap.add_argument("--pipeline", type=int, default=4,
                    help="contexts/streams used alternately (1 = strictly one batch at a time)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run "
                     "(one process per GPU)")
        args.gpus = world

    cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu:
        n_cores = min(16, len(os.sched_getaffinity(0)))
        cpu = cpu_baseline(n_cores)

    import numpy as np
    import torch
    import sift_amd
    from sift_amd import api, synth

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU")
    # SIFT_AMD_BENCH_REHEARSAL=1: every rank uses GPU 0 and the ranks meet over gloo, to rehearse
    # the multi-process path on a one-GPU machine (not a measurement)
    rehearsal = os.environ.get("SIFT_AMD_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    red_dev = "cpu" if rehearsal else "cuda"

    B = args.batch
    # shard of the synthetic batch owned by this rank: image b of rank r has seed 12345 + r*B + b
    host = np.stack([synth.make(args.kind, W_IM, H_IM, seed=12345 + rank * B + b) for b in range(B)])
    d_imgs = torch.from_numpy(host).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    # Several contexts (one pyramid pool each) on their own streams, used in turn: the latency-bound
    # tail of one batch (orientation, descriptors, host gather) overlaps the bandwidth-bound
    # scale-space of the next.  All batches read the same resident input images.
    n_ctx = max(1, args.pipeline)
    ctxs = [sift_amd.Context(W_IM, H_IM, max_batch=B, device=local_rank) for _ in range(n_ctx)]
    streams = [torch.cuda.Stream() for _ in range(n_ctx)]
    ctx = ctxs[0]

    h_pinned = torch.from_numpy(host).pin_memory() if args.host_input else None
    d_stage = [torch.empty_like(d_imgs) for _ in range(n_ctx)] if args.host_input else None

    def run_steps(n_steps):
        """Queues n_steps batches; the gather of batch i is issued after batch i+1 is queued."""
        last = None
        pending = []
        for i in range(n_steps):
            k = i % n_ctx
            if len(pending) == n_ctx:
                last = pending.pop(0).gather()
            src = d_imgs
            if args.host_input:
                with torch.cuda.stream(streams[k]):
                    d_stage[k].copy_(h_pinned, non_blocking=True)
                src = d_stage[k]
            ctxs[k].detect_device(src.data_ptr(), B, streams[k].cuda_stream)
            pending.append(ctxs[k])
        for c in pending:
            last = c.gather()
        return last

    def step():
        ctx.detect_device(d_imgs.data_ptr(), B, stream)
        return ctx.gather()          # host-side gather of this rank's keypoints

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    keys, counts = run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0

    # Roofline leg: the same steps again with the library's per-launch event timers on the
    # dominant kernel (hipEvents on the stream the kernels run on).  Timed launches are issued
    # one by one instead of replaying the cached hipGraph, so this leg is kept out of `value`.
    fused = os.environ.get("SIFT_AMD_BLUR", "fused") != "split"
    dom = api.K_BLUR_FUSED if fused else api.K_BLUR_H
    ctx.profile_enable([dom])
    ctx.profile_read(dom, -1, reset=True)
    roof_steps = min(args.steps, 10)
    for _ in range(roof_steps):
        step()
    torch.cuda.synchronize()
    ctx.profile_enable([])

    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    nk = torch.tensor([float(counts.sum())], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(nk, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())

    if rank == 0:
        total_images = B * world * args.steps
        # Dominant kernel: the Gaussian blur with DoG epilogue on octave 0 (n_sca-1 = 5 launches per
        # step; the seed launch, which evaluates the bilinear oversampling on the fly and writes no
        # DoG, is a different instantiation and is timed separately).
        # Algorithmic bytes per octave-0 pixel and launch, fused kernel: read the source image (4 B)
        # and write the blurred image (4 B) = 8 B (DoG images are not stored: the extrema scan and
        # the refinement form DoG values from the Gaussian stack); the launch that also emits the
        # next octave's first image adds 1 B (a quarter of the pixels).
        # Two-kernel form, row pass: read the column-pass image, write the Gaussian image = 8 B.
        n0 = ctx.dims[0][0] * ctx.dims[0][1]
        ms, launches = ctx.profile_read(dom, 0)
        per_step = ctx.n_sca - 1
        if fused:
            bytes_per_step = B * n0 * (8 * (ctx.n_sca - 1) + (1 if ctx.n_oct > 1 else 0))
            kname = "k_blur_fused<R,false> (column+row pass, contrast marks), octave 0, R = 5,7,8,10,13"
        else:
            per_step = ctx.n_sca                     # the seed row pass is timed in the same class
            bytes_per_step = B * n0 * 8 * ctx.n_sca
            kname = "k_blur_h (row pass + DoG epilogue), octave 0"
        steps_timed = launches / per_step if per_step else 0
        achieved = (bytes_per_step * steps_timed) / (ms * 1e-3) / 1e9 if ms > 0 else None
        traffic = load_traffic()
        out = {
            "metric": "1080p images/sec",
            "value": total_images / elapsed,
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"batch of {B} synthetic 1920x1080 images per GPU per step "
                            f"({args.kind} S{'1' if args.kind == 'smooth' else '2'}, seeds 12345+), "
                            "sift_anatomy default parameters, inputs resident in HBM, "
                            "keypoints gathered to host memory inside the timed region",
                "image": "1920x1080",
                "batch_per_gpu": B,
                "batches_in_flight": n_ctx,
                "octaves": ctx.n_oct,
                "keypoints_per_image": float(nk.item()) / (B * world),
                "sharding": "by image, no collective on the data path",
                "input": "pinned host memory, H2D copy inside the timed region" if args.host_input
                         else "resident in HBM",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kname,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                # PMC passes are taken at some batch size; HBM traffic of this kernel is proportional
                # to the number of images per launch
                "traffic": traffic["bytes_per_launch"] * B / traffic.get("images_per_launch", 16) if traffic else None,
                "launches_timed": launches,
                "measured": f"{roof_steps} extra steps right after the timed region, per-launch hipEvents",
                "avg_launch_ms": (ms / launches) if launches else None,
                "algorithmic_bytes_per_launch": bytes_per_step / per_step,
            },
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    for c in ctxs:
        c.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
