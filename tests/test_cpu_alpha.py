"""CPU tests of the RGBA alpha path: the oracle against the vectors produced by the reference's own
edge_guided_alpha_upscale (oracle/pin_against_reference.py G14; cv2's two calls served by the oracle's restatement of
OpenCV's documented arithmetic), and known answers for those two restated functions."""

import os

import numpy as np
import torch
from safetensors.torch import load_file

from seedvr2_ref import alpha as oalpha

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "alpha_pins.safetensors")


def test_oracle_alpha_matches_reference_vectors():
    t = load_file(GOLD)
    for tag in ("binary", "gradient", "matte"):
        got = oalpha.edge_guided_alpha_upscale(t[f"G14.{tag}.alpha"], t[f"G14.{tag}.rgb"])
        assert torch.equal(got, t[f"G14.{tag}.out"]), tag
    assert torch.equal(oalpha.detect_edges((t["G14.binary.rgb"].float() + 1) / 2), t["G14.binary.edges"])
    # the two fixtures take the two branches of the reference
    out_m, out_g = t["G14.matte.out"], t["G14.gradient.out"]
    assert ((out_m == 0) | (out_m == 1)).float().mean() > 0.5 and ((out_g > 0.02) & (out_g < 0.98)).float().mean() > 0.5


def test_opencv_restatements_known_answers():
    # RGB2GRAY: 14-bit fixed point, weights sum to 1 << 14
    px = np.array([[[255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 20, 30]]], dtype=np.uint8)
    assert oalpha.rgb_to_gray_u8(px).tolist() == [[255, 76, 150, 29, 18]]
    # Sobel on a vertical step edge: 4 * step in the middle, reflect-101 border keeps the edge columns symmetric
    g = np.zeros((4, 6), np.uint8)
    g[:, 3:] = 10
    sx, sy = oalpha.sobel_f64(g, 1, 0), oalpha.sobel_f64(g, 0, 1)
    assert sx.dtype == np.float64 and sx.tolist()[0] == [0, 0, 40, 40, 0, 0] and not sy.any()
    ramp = np.arange(5, dtype=np.uint8)[None].repeat(3, 0) * 7
    assert oalpha.sobel_f64(ramp, 1, 0)[1].tolist() == [0, 56, 56, 56, 0]      # reflect-101: p[-1] = p[1]
    flat = torch.full((1, 3, 4, 4), 0.5)
    assert oalpha.detect_edges(flat).abs().max() == 0                            # 0/0 frame -> zeros
