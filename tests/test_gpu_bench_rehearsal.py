"""GPU: bench.py end to end on small images - the single-process line, and the strong-scaling
mode (`--total-images`) with two ranks sharing GPU 0 over gloo (SIFT_AMD_BENCH_REHEARSAL=1): the
per-image keypoint counts gathered from two ranks equal the single-process ones."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--width", "320", "--height", "240", "--batch", "6", "--pipeline", "2", "--no-cpu", "--no-extras"]


def _run(cmd, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_line_on_small_images(hip):
    out = _run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--repeats", "2"] + SMALL)
    assert out["n_gpus"] == 1 and out["scaling"] == "weak" and out["value"] > 0
    assert out["repeats"]["n"] == 2 and out["config"]["keypoints_per_image"] > 10
    r = out["roofline"]
    # six blur launches of octave 0 at this batch size; the one of the last image, evaluated only where it
    # is read, is listed apart and kept out of the roofline figure
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1
    assert len(r["per_launch"]) == 5 and r["sparse_launch"]["scale"] == 5 and not r["sparse_launch"]["in_roofline"]
    assert out.get("verified") is None or out["verified"]["ok"]      # (no fixture of this size)
    assert all(e["binding"] in ("hbm", "valu") for e in r["per_launch"])


def test_strong_scaling_two_ranks_equal_one(hip):
    one = _run([sys.executable, "bench.py", "--total-images", "29", "--repeats", "1"] + SMALL)
    port = 32500 + os.getpid() % 2000
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2",
                "--total-images", "29", "--repeats", "1"] + SMALL,
               {"SIFT_AMD_BENCH_REHEARSAL": "1"})
    assert one["scaling"] == two["scaling"] == "strong" and two["n_gpus"] == 2
    assert one["config"]["keypoints_total"] == two["config"]["keypoints_total"] > 29 * 10
    assert one["config"]["per_image_counts_sha1"] == two["config"]["per_image_counts_sha1"]
