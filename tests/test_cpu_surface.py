"""Drop-in surface: CLI flags and node inputs are the reference's (names extracted from the reference into
tests/golden/surface.json); unsupported low-VRAM options are rejected loudly, never silently ignored."""

import json
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def surface():
    with open(os.path.join(ROOT, "tests", "golden", "surface.json")) as fh:
        return json.load(fh)


def _cli(pkg):
    import importlib.util
    spec = importlib.util.spec_from_file_location("svr2_cli", os.path.join(ROOT, "comfyui-seedvr2_amd", "inference_cli.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_cli_flags_match_reference(pkg, surface):
    cli = _cli(pkg)
    a = cli.parse_arguments(["in.png"])
    have = {("input" if k == "input" else "--" + k) for k in vars(a)}
    assert have == set(surface["cli_flags"])
    assert (a.resolution, a.batch_size, a.seed, a.temporal_overlap, a.color_correction) == (1080, 5, 42, 0, "lab")
    with pytest.raises(SystemExit):
        cli.parse_arguments(["in.png", "--blocks_to_swap", "4"])
    with pytest.raises(SystemExit):
        cli.parse_arguments(["in.png", "--batch_size", "4"])


def test_node_inputs_and_loader_dicts(pkg, surface):
    N = pkg.nodes
    assert N.UPSCALER_INPUTS == surface["upscaler_inputs"]
    import inspect
    params = list(inspect.signature(N.SeedVR2VideoUpscaler.execute).parameters)[: len(N.UPSCALER_INPUTS)]
    assert params == N.UPSCALER_INPUTS
    dit_cfg, = N.SeedVR2LoadDiTModel.execute()
    vae_cfg, = N.SeedVR2LoadVAEModel.execute()
    assert list(dit_cfg) == N.DIT_KEYS and list(vae_cfg) == N.VAE_KEYS
    with pytest.raises(NotImplementedError):
        N.SeedVR2LoadVAEModel.execute(decode_tiled=True)
    with pytest.raises(NotImplementedError):
        N.SeedVR2VideoUpscaler.execute(torch.zeros(1, 8, 8, 3), dit_cfg, vae_cfg, color_correction="lab", pipeline=object())


def test_runner_condition_matches_reference_semantics(pkg):
    r = pkg.infer.VideoDiffusionInfer()
    lat = torch.randn(2, 3, 4, 16)
    cond = r.get_condition(torch.zeros_like(lat), lat, "sr")
    assert cond.shape == (2, 3, 4, 17) and torch.equal(cond[..., :16], lat) and (cond[..., 16] == 1).all()
    with pytest.raises(NotImplementedError):
        pkg.infer.VideoDiffusionInfer(vae_tiling_enabled=True)
