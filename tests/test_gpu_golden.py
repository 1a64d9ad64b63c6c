"""GPU: the HIP library against the golden fixtures produced by the compiled reference,
including the full-size 1080p and 4K inputs (bit-level digests of every scale-space octave,
per-octave stage sizes, digests of the final keypoint table)."""
import json
import os

import numpy as np
import pytest

import sift_amd
from sift_amd import abi, api, synth

from test_oracle_golden import GOLD, INDEX, digest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(INDEX))
def test_hip_reproduces_golden(hip, name):
    e = INDEX[name]
    img = synth.make(e["kind"], e["w"], e["h"], e["seed"])
    assert digest(img) == e["input_sha1"]
    p = abi.SiftParameters.default().copy(**e["params"])
    n_oct = e["n_oct"]
    with sift_amd.Context(e["w"], e["h"], 1, p, keep_all_extrema=True, keep_scalespace=True) as ctx:
        ctx.detect_host(img)
        assert ctx.n_oct == n_oct and [list(d) for d in ctx.dims] == e["dims"]
        sizes, ext = ctx.stage_counts(0)
        assert ext.tolist() == e["stage_counts_per_octave"]["A"], "discrete extrema per octave"
        seed, ref, stage = ctx.candidates(0)
        for k, s in enumerate("ABCDEF"):
            per_oct = np.bincount(seed["o"][stage >= k], minlength=n_oct)[:n_oct].tolist()
            assert per_oct == e["stage_counts_per_octave"][s], f"stage {s} per octave"
            assert sizes[k] == sum(per_oct)
        for o in range(n_oct):
            assert digest(ctx.scalespace(0, 0, o)) == e["gss_sha1"][o], f"gss octave {o} bits"
            assert digest(ctx.scalespace(0, 1, o)) == e["dog_sha1"][o], f"dog octave {o} bits"
        std, keys = ctx.keypoints(0)
        orih, descr, src = ctx.extras(0)
    assert len(std) == e["n_keys"]
    # the throughput configuration (first contrast test fused into the scan, DoG images never
    # stored) must give the same keypoint records
    with sift_amd.Context(e["w"], e["h"], 1, p) as fast:
        fast.detect_host(img)
        std_fast, keys_fast = fast.keypoints(0)
        with pytest.raises(sift_amd.api.SiftAmdError):
            fast.scalespace(0, 1, 0)                     # DoG images are not stored on this path
        # (its last image of a stack is computed only where the scan and the refinement read it)
        for o in (0, n_oct - 1):
            stack = fast.scalespace(0, 0, o)
            assert [digest(im) for im in stack[:-1]] == e["gss_image_sha1"][o][:-1]
    assert std_fast.tobytes() == std.tobytes() and keys_fast.tobytes() == keys.tobytes()
    ints = np.stack([keys["o"], keys["s"], keys["i"], keys["j"]], 1).astype(np.int32)
    assert digest(ints) == e["keys_sha1"]["ints"]
    path = os.path.join(GOLD, name + ".npz")
    if os.path.exists(path):
        g = np.load(path)
        xyst = np.stack([std["x"], std["y"], std["scale"], std["orientation"]], 1)
        assert np.abs(xyst - g["flts"][:, :4]).max(initial=0) <= 1e-3           # stated tolerance
        assert np.abs(std["descriptor"].astype(int) - g["descr"].astype(int)).max(initial=0) <= 1
        assert np.array_equal(std["descriptor"], g["descr"])                     # and in fact exact
        assert np.array_equal(xyst.view(np.uint32), g["flts"][:, :4].view(np.uint32))
        assert np.array_equal(orih.view(np.uint32), g["orihist"].view(np.uint32))
    else:
        xyst = np.ascontiguousarray(np.stack([std["x"], std["y"], std["scale"], std["orientation"]], 1))
        exact_xyst = digest(xyst) == e["keys_sha1"]["xyst"]
        exact_descr = digest(np.ascontiguousarray(std["descriptor"])) == e["keys_sha1"]["descr_u8"]
        print(name, "keys", len(std), "xyst exact", exact_xyst, "descr exact", exact_descr)
        assert exact_xyst and exact_descr


# Diagnostic switches, read when a context is created.  Each selects another route through the
# same arithmetic and must give the same bytes as the default route.
SWITCHES = [{"SIFT_AMD_CELLS": "0"},      # extrema scan without contrast marks
            {"SIFT_AMD_SPARSE": "0"},     # the last image of every stack computed everywhere
            {"SIFT_AMD_SPARSE_WIDE": "1", "SIFT_AMD_STRIP": "72"},   # ... only where needed, 480-column tiles, short strips
            {"SIFT_AMD_SPARSE_WIDE": "0", "SIFT_AMD_STRIP": "40"},   # ... 224-column tiles
            {"SIFT_AMD_PAIR": "0"},       # every blur a launch of its own (no two-blur launches)
            {"SIFT_AMD_PAIR_STRIP": "64"},   # two-blur launches forced, in short strips
            # single-blur launches in strips of 9 iterations; on octaves that take 480-column tiles
            # (the 480x270 cases: 960 and 480 wide) rows are then fetched two iterations ahead and the
            # last iteration of a strip is an empty one
            {"SIFT_AMD_STRIP": "72"},
            {"SIFT_AMD_PAIR": "0", "SIFT_AMD_STRIP": "544"},   # the same path with tall strips, every radius
            {"SIFT_AMD_PAIR": "0", "SIFT_AMD_STRIP": "72"},
            {"SIFT_AMD_WIDE_MIN": "300", "SIFT_AMD_STRIP": "64"},   # 480-column tiles forced on every octave >= 300 wide
            {"SIFT_AMD_FORK": "0"},       # one chain of launches (small batches alternate the octaves between two
                                          # streams by default: parallel branches of the graph)
            {"SIFT_AMD_FORK": "1", "SIFT_AMD_GRAPH": "0"},
            {"SIFT_AMD_GRAPH": "0"},      # no hipGraph replay
            {"SIFT_AMD_DEBUG": "1"},      # generic descriptor instantiation
            {"SIFT_AMD_DEBUG": "2"},      # descriptor contributions listed per cell instead of per bin
            {"SIFT_AMD_DEBUG": "3"}]      # per-cell lists with the generic list walk


@pytest.mark.parametrize("env", SWITCHES, ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
@pytest.mark.parametrize("name", ["white_200x150", "natural_800x640_a", "white_333x257_spo4", "white_480x270",
                                  "natural_480x270"])
def test_diagnostic_switches_give_the_same_records(hip, monkeypatch, name, env):
    e = INDEX[name]
    img = synth.make(e["kind"], e["w"], e["h"], e["seed"])
    p = abi.SiftParameters.default().copy(**e["params"])
    g = np.load(os.path.join(GOLD, name + ".npz"))
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    api.launch_stats(reset=True)
    with sift_amd.Context(e["w"], e["h"], 2, p) as ctx:
        for rep in range(3):                                  # plain launches, graph build, replay
            ctx.detect_host(np.stack([img, img]))
            if rep == 0:
                # the switch reached the route it is meant to exercise
                st = api.launch_stats(reset=True)
                wide = any(((w + 479) // 480) * 512 < ((w + 223) // 224) * 256 for w, _ in ctx.dims)
                if "SIFT_AMD_WIDE_MIN" in env:
                    wide = ctx.dims[0][0] >= 300
                if int(env.get("SIFT_AMD_STRIP", 0)) >= 64 and wide:
                    assert st["wide_ahead2"] > 0 and st["wide_ahead2_strip_max"] == ((int(env["SIFT_AMD_STRIP"]) + 7) & ~7), st
                sparse_ok = any(w % 4 == 0 for w, _ in ctx.dims)       # the sparse form takes 16-byte items only
                assert (st["sparse"] > 0) == (sparse_ok and env.get("SIFT_AMD_SPARSE") != "0"
                                              and env.get("SIFT_AMD_CELLS") != "0"), st
                if env.get("SIFT_AMD_PAIR") == "0":
                    assert st["pair"] == 0 and st["pair_seed"] == 0, st
                if "SIFT_AMD_PAIR_STRIP" in env and ctx.dims[0][0] >= 900 and ctx.dims[0][0] % 4 == 0:
                    assert st["pair_seed"] == 1 and st["pair"] >= 1 and st["pair_strip_max"] == 64, st
            for b in range(2):
                std, keys = ctx.keypoints(b)
                orih, descr, _ = ctx.extras(b)
                assert len(std) == e["n_keys"]
                xyst = np.stack([std["x"], std["y"], std["scale"], std["orientation"]], 1)
                assert np.array_equal(xyst.view(np.uint32), g["flts"][:, :4].view(np.uint32))
                assert np.array_equal(std["descriptor"], g["descr"])
                assert np.array_equal(orih.view(np.uint32), g["orihist"].view(np.uint32))
        sizes, _ = ctx.stage_counts(0)
        assert sizes[1:].tolist() == [sum(e["stage_counts_per_octave"][s]) for s in "BCDEF"]


@pytest.mark.parametrize("name,env", [("natural_800x640_a", {}), ("natural_480x270", {"SIFT_AMD_SPARSE_WIDE": "1"}),
                                      ("smooth_800x640", {"SIFT_AMD_STRIP": "512"})])
def test_last_stack_image_where_it_is_read(hip, monkeypatch, name, env):
    """The batched pipeline computes the last Gaussian image of an octave only near samples that
    pass the first contrast test.  Everything within 6 samples of an 8 x 64 cell that holds such a
    sample (every place the scan and the refinement can read) equals the oracle's image bit for bit."""
    from oracle import pyoracle
    e = INDEX[name]
    img = synth.make(e["kind"], e["w"], e["h"], e["seed"])
    want = pyoracle.run(img, keep_ss=True)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    api.launch_stats(reset=True)
    with sift_amd.Context(e["w"], e["h"], 3) as ctx:
        ctx.detect_host(np.stack([img, img, img]))
        assert api.launch_stats()["sparse"] == sum(w % 4 == 0 for w, _ in ctx.dims)    # (16-byte items only)
        thr = ctx.geometry.thresh_pre
        checked = 0
        for o in range(ctx.n_oct):
            dog = want["dog"][o]
            h, w = dog.shape[1:]
            hit = (np.abs(dog[1:-1]) > np.float32(thr)).any(0)
            ch, cw = (h + 7) // 8, (w + 63) // 64
            pad = np.zeros((ch * 8, cw * 64), bool)
            pad[:h, :w] = hit
            cells = np.repeat(np.repeat(pad.reshape(ch, 8, cw, 64).any((1, 3)), 8, 0), 64, 1)[:h, :w]
            near = np.zeros_like(cells)
            ys, xs = np.nonzero(cells)
            for dy in (-6, 0, 6):                       # cells are 8 x 64 blocks: their corners moved by 6 cover the dilation
                for dx in (-6, 0, 6):
                    near[np.clip(ys + dy, 0, h - 1), np.clip(xs + dx, 0, w - 1)] = True
            for b in (0, 2):
                got = ctx.scalespace(b, 0, o)[-1]
                ref = want["gss"][o][-1]
                assert np.array_equal(got.view(np.uint32)[near], ref.view(np.uint32)[near]), f"octave {o} image {b}"
            checked += int(near.sum())
        assert checked > 1000
        std, _ = ctx.keypoints(1)
    assert len(std) == e["n_keys"]
