"""`--cuda_device 0,1` fans out by itself like the reference's CLI (inference_cli.py:1048-1209): the parent starts one
child per listed device, the children run the product's multi-rank path over gloo on the CPU (tests/cli_cpu_child.py
substitutes only the GPU-bound pieces), and the file on disk equals the reference's multi-GPU semantics: streaming
segment split (:1082-1097), per-segment chunk streaming (:566-654), boundary blend on the final frames (:1163-1198)."""

import os
import sys
import time

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = [sys.executable, os.path.join(HERE, "cli_cpu_child.py")]


def _cli(pkg):
    import importlib
    return importlib.import_module("seedvr2_amd.inference_cli")


def _clip(n, h=6, w=8, seed=0):
    g = np.random.default_rng(seed)
    return g.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)


@pytest.mark.parametrize("ndev,total,overlap,chunk", [(2, 23, 4, 0), (3, 20, 2, 5), (2, 9, 0, 4)])
def test_cuda_device_list_fans_out_and_matches_reference_semantics(pkg, tmp_path, monkeypatch, ndev, total, overlap, chunk):
    from cli_cpu_child import fake_upscale
    from seedvr2_ref import video_io as ovio
    from test_cpu_parallel import _reference_concat
    cli = _cli(pkg)
    clip = _clip(total)
    src, dst = tmp_path / "in.npy", tmp_path / "out.npy"
    np.save(src, clip)
    argv = [str(src), "--output", str(dst), "--cuda_device", ",".join(str(i) for i in range(ndev)), "--batch_size", "5",
            "--temporal_overlap", str(overlap), "--chunk_size", str(chunk), "--color_correction", "none"]
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    rc = cli.main(argv, runtime=cli.Runtime(child_cmd=CHILD))
    assert rc == 0
    got = np.load(dst)
    # expectation: every rank streams its own segment in chunks with raw context frames, results blended at the boundaries
    frames = ovio.u8_to_unit(clip)
    ranges = pkg.parallel.segment_ranges(total, ndev, overlap, 5, True)
    parts = []
    for r, (s, e) in enumerate(ranges):
        outs = ovio.stream_chunks(frames[s:e], chunk, overlap, 0, lambda x, p, r=r: fake_upscale(x, r))
        parts.append(torch.cat(outs, dim=0))
    want = ovio.unit_to_u8(_reference_concat(parts, overlap).to(torch.bfloat16))
    assert got.shape == want.shape == (total, 12, 16, 3)
    assert np.array_equal(got, want)


def test_a_dead_worker_ends_the_run_instead_of_hanging(pkg, tmp_path, monkeypatch):
    cli = _cli(pkg)
    src = tmp_path / "in.npy"
    np.save(src, _clip(10))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setenv("SVR2_TEST_FAIL_RANK", "1")
    t0 = time.time()
    rc = cli.main([str(src), "--output", str(tmp_path / "o.npy"), "--cuda_device", "0,1", "--color_correction", "none"],
                  runtime=cli.Runtime(child_cmd=CHILD))
    assert rc == 1 and time.time() - t0 < 90
    assert not (tmp_path / "o.npy").exists()


def test_any_batch_size_is_accepted_and_padded(pkg):
    """The reference pads every batch to 4n+1 (generation_phases.py:109-124); its own examples use --batch_size 31."""
    cli = _cli(pkg)
    a = cli.parse_arguments(["in.mp4", "--batch_size", "31"])
    assert a.batch_size == 31
    b, _ = pkg.pipeline.plan_batches(40, 31, 0)
    assert [len(x.index) for x in b] == [33, 9] and b[0].index[31:] == [29, 28]
    with pytest.raises(SystemExit):
        cli.parse_arguments(["in.mp4", "--batch_size", "0"])


def test_single_device_selection_and_text_embedding_lookup(pkg, tmp_path, monkeypatch):
    """--cuda_device 3 selects cuda:3 for a single rank (and sets it as the current device before any kernel); the node
    resolves pos_emb.pt by itself when execute() gets only the schema inputs."""
    cli = _cli(pkg)
    chosen = []
    rt = cli.Runtime(set_device=lambda dev: chosen.append(dev))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    rc = cli.main([str(tmp_path / "missing.mp4"), "--cuda_device", "3"], runtime=rt)
    assert rc == 1 and chosen == ["cuda:3"]
    # text embedding: found in the model directory, loaded as a [tokens, dim] tensor; a clear error otherwise
    with pytest.raises(FileNotFoundError) as ei:
        pkg.weights.load_text_embedding("pos_emb.pt", str(tmp_path))
    assert "pos_emb.pt" in str(ei.value)
    torch.save(torch.ones(1, 58, 16), tmp_path / "pos_emb.pt")
    t = pkg.weights.load_text_embedding("pos_emb.pt", str(tmp_path))
    assert tuple(t.shape) == (58, 16)
    monkeypatch.setenv("SEEDVR2_MODEL_DIR", str(tmp_path))
    seen = {}

    def fake_find(name, model_dir=None):
        return None

    monkeypatch.setattr(pkg.weights, "find_model_file", fake_find)
    with pytest.raises(RuntimeError) as ei:      # reaches the checkpoint lookup: the embedding was resolved without the caller
        pkg.nodes.SeedVR2VideoUpscaler.execute(torch.zeros(1, 8, 8, 3), {"model": "m", "device": "cuda:0"},
                                               {"model": "v", "device": "cuda:0"}, color_correction="none")
    assert "model files not found" in str(ei.value)
