"""CPU tests of the spatial VAE tiling: the oracle against the vectors produced by the reference's own
tiled_encode / tiled_decode (oracle/pin_against_reference.py G13), and the product's tile plan / fade weights against the
oracle.  The blend kernels themselves are covered on the GPU (tests/test_gpu_tiling.py)."""

import json
import os

import pytest
import torch
from safetensors import safe_open

from seedvr2_ref import tiling as otile

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiling_pins.safetensors")


def _gold():
    with safe_open(GOLD, "pt") as f:
        cases = json.loads(f.metadata()["cases"])
        return cases, {k: f.get_tensor(k) for k in f.keys()}


def test_oracle_tiling_matches_reference_vectors():
    cases, t = _gold()
    n = 0
    for tag in ("f32", "bf16"):
        for i in range(6):
            c = cases[f"enc.{i}"]
            got = otile.tiled_encode(t[f"G13.enc.{tag}.{i}.x"].clone(), otile.synthetic_enc_codec, tuple(c["tile"]), tuple(c["overlap"]))
            assert torch.equal(got, t[f"G13.enc.{tag}.{i}.out"]), (tag, i)
            c = cases[f"dec.{i}"]
            got = otile.tiled_decode(t[f"G13.dec.{tag}.{i}.z"].clone(), otile.synthetic_dec_codec, tuple(c["tile"]), tuple(c["overlap"]))
            assert torch.equal(got, t[f"G13.dec.{tag}.{i}.out"]), (tag, i)
            n += 2
    assert n == 24


def test_tiling_changes_the_result_near_seams_only_when_tiles_disagree():
    """Sanity of the fixture itself: tiled != untiled for the border-sensitive codec, equal for a pointwise codec."""
    cases, t = _gold()
    x = t["G13.enc.f32.0.x"]
    assert not torch.equal(t["G13.enc.f32.0.out"], otile.synthetic_enc_codec(x))
    point = lambda v: (v[:, ::4, ::8, ::8] * 2).contiguous()
    got = otile.tiled_encode(x.clone(), point, (32, 32), (8, 8))
    assert torch.allclose(got, point(x), atol=1e-6)


@pytest.mark.parametrize("hl,wl,tile,ov", [(6, 8, (4, 4), (1, 1)), (6, 8, (4, 5), (2, 1)), (5, 9, (3, 3), (2, 1)),
                                           (270, 480, (128, 128), (16, 16)), (7, 6, (2, 6), (0, 3)), (6, 10, (4, 4), (3, 3)),
                                           (1, 1, (1, 1), (0, 0)), (9, 9, (1, 1), (0, 0))])
def test_product_tile_plan_matches_oracle(pkg, hl, wl, tile, ov):
    plan = pkg.vae.VideoVAEHip.tile_plan(hl, wl, tile, ov)
    assert plan == otile.tile_plan(hl, wl, tile, ov)
    covered = torch.zeros(hl, wl, dtype=torch.bool)
    for y, ye, x, xe in plan:
        covered[y:ye, x:xe] = True
    assert covered.all()


def test_product_fade_weights_match_oracle(pkg):
    fade = pkg.vae.VideoVAEHip._fade
    for n, ov, steps in ((16, 4, 4), (9, 8, 16), (5, 0, 0), (3, 2, 2), (64, 16, 16)):
        r = otile.ramp(steps, torch.bfloat16) if steps else None
        for head in (False, True):
            for tail in (False, True):
                assert torch.equal(fade(n, ov, r, head, tail), otile.edge_weights(n, ov, r, head, tail, torch.bfloat16))
