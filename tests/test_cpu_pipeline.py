"""CPU tests of the product's host-side pipeline logic against the oracle / torch's own ops, the VAE
oracle against the reference-derived golden vectors, and a world_size-2 gloo run of the frame sharding."""

import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from safetensors.torch import load_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("hin,win,res", [(20, 36, 48), (72, 128, 216), (27, 40, 54), (48, 48, 32)])
def test_resize_taps_match_aten_antialias_bicubic(pkg, hin, win, res):
    """The host-computed separable taps reproduce torch's _upsample_bicubic2d_aa (what torchvision's
    resize(..., BICUBIC, antialias=True) calls) - up- and down-scaling."""
    P = pkg.pipeline
    rh, rw, *_ = P.target_dims(hin, win, res)
    x = torch.rand(2, 3, hin, win, generator=torch.Generator().manual_seed(0))
    ref = F.interpolate(x, size=(rh, rw), mode="bicubic", align_corners=False, antialias=True)
    wy, ymin, ycnt = P.aa_bicubic_taps(hin, rh)
    wx, xmin, xcnt = P.aa_bicubic_taps(win, rw)
    My = np.zeros((rh, hin), np.float32)
    Mx = np.zeros((rw, win), np.float32)
    for i in range(rh):
        My[i, ymin[i]:ymin[i] + ycnt[i]] = wy[i, :ycnt[i]]
    for i in range(rw):
        Mx[i, xmin[i]:xmin[i] + xcnt[i]] = wx[i, :xcnt[i]]
    got = torch.einsum("yh,nchw,xw->ncyx", torch.from_numpy(My), x, torch.from_numpy(Mx))
    assert (got - ref).abs().max() < 2e-5


def test_resize_taps_carry_host_side_bounds(pkg):
    """The tap tables travel with the maxima the resize wrappers check against the input size (hip.ResizeTaps), taken on
    the host when the tables are built: reading them back from the device on every call would synchronise every batch."""
    import numpy as np
    import torch
    wy, ymin, ycnt = pkg.pipeline.aa_bicubic_taps(37, 90)
    wx, xmin, xcnt = pkg.pipeline.aa_bicubic_taps(53, 40)
    bounds = (int((ymin + ycnt).max()), int((xmin + xcnt).max()), int(ycnt.max()), int(xcnt.max()))
    taps = pkg.hip.ResizeTaps((torch.from_numpy(np.ascontiguousarray(a)) for a in (wy, ymin, ycnt, wx, xmin, xcnt)), bounds)
    assert len(taps) == 6 and taps.bounds == bounds and pkg.hip._taps_bounds(taps) == bounds
    assert bounds[0] <= 37 and bounds[1] <= 53 and bounds[2] <= wy.shape[1] and bounds[3] <= wx.shape[1]
    assert pkg.hip._taps_bounds(tuple(taps)) == bounds          # a plain tuple falls back to reducing the tables


def test_target_dims_and_oracle_agree(pkg):
    from seedvr2_ref import pipeline as opipe
    for (h, w, r) in [(720, 1280, 2160), (540, 960, 1080), (256, 256, 512), (1280, 720, 1080), (101, 57, 90)]:
        assert pkg.pipeline.target_dims(h, w, r) == opipe.target_dims(h, w, r)
    assert pkg.pipeline.target_dims(720, 1280, 2160) == (2160, 3840, 2160, 3840, 2160, 3840)
    assert pkg.pipeline.target_dims(540, 960, 1080) == (1080, 1920, 1080, 1920, 1088, 1920)


@pytest.mark.parametrize("total,bs,ov,uniform", [(64, 21, 4, False), (8, 9, 0, False), (1, 1, 0, False),
                                                 (23, 5, 2, False), (10, 9, 3, True), (3, 5, 0, True),
                                                 (2, 1, 0, False), (7, 5, 5, False), (30, 13, 1, True)])
def test_batch_plan_matches_oracle_padding(pkg, total, bs, ov, uniform):
    """Index plan == the oracle's restatement of pad_video_temporal / the batching loop."""
    from seedvr2_ref import pipeline as opipe
    batches, eff = pkg.pipeline.plan_batches(total, bs, ov, uniform)
    ranges, eff_o = opipe.batch_ranges(total, bs, ov)
    assert eff == eff_o and [(b.start, b.end) for b in batches] == ranges
    frames = torch.arange(total).float().view(total, 1, 1, 1)
    for b in batches:
        v = frames[b.start:b.end]
        if uniform and v.shape[0] < bs:
            v = opipe.pad_video_temporal(v, bs - v.shape[0])
        v = opipe.pad_video_temporal(v)
        assert v.flatten().long().tolist() == b.index
        assert len(b.index) % 4 == 1 and b.ori_len == b.end - b.start
    if (total, bs, ov) == (64, 21, 4):
        assert [(b.start, b.end) for b in batches] == [(0, 21), (17, 38), (34, 55), (51, 64)]


def test_blend_weights(pkg):
    from seedvr2_ref import pipeline as opipe
    for ov in (1, 2, 3, 4, 8):
        a, b = torch.rand(ov, 2, 3, 3).bfloat16(), torch.rand(ov, 2, 3, 3).bfloat16()
        w = pkg.pipeline.blend_weights(ov).view(ov, 1, 1, 1)
        mine = (a.float() * w).bfloat16().float() + (b.float() * (1 - w).bfloat16().float()).bfloat16().float()
        ref = opipe.blend_overlapping_frames(a, b, ov)
        assert (mine.bfloat16().float() - ref.float()).abs().max() <= 2 ** -7


def test_vae_oracle_golden():
    """Causal conv / per-frame GroupNorm of the oracle vs vectors produced by the reference's own
    InflatedCausalConv3d / causal_norm_wrapper (oracle/pin_against_reference.py, G10)."""
    from seedvr2_ref import vae as ovae
    t = load_file(os.path.join(ROOT, "tests", "golden", "vae_pins.safetensors"))
    for tag, stride, pad in (("k3s1", (1, 1, 1), (1, 1, 1, 1)), ("k3s2", (2, 2, 2), (1, 1, 1, 1)),
                             ("k133", (1, 2, 2), (0, 0, 0, 0)), ("k1", (1, 1, 1), (0, 0, 0, 0))):
        p = {"c.weight": t[f"G10.{tag}.w"], "c.bias": t[f"G10.{tag}.b"]}
        x = t[f"G10.{tag}.x"]
        assert (ovae.causal_conv(p, "c", x, stride=stride, pad_hw=pad) - t[f"G10.{tag}.y"]).abs().max() < 1e-5
        st = ovae.State()
        outs = [ovae.causal_conv(p, "c", x[:, :5], stride=stride, pad_hw=pad, st=st)]
        st.active = True
        for a in (5, 7):
            outs.append(ovae.causal_conv(p, "c", x[:, a:a + 2], stride=stride, pad_hw=pad, st=st))
        assert (torch.cat(outs, dim=1) - t[f"G10.{tag}.y"]).abs().max() < 1e-5
    y = ovae.group_norm_per_frame({"n.weight": t["G10.gn.w"], "n.bias": t["G10.gn.b"]}, "n", t["G10.gn.x"], 4, 1e-6)
    assert (y - t["G10.gn.y"]).abs().max() < 1e-6


def test_vae_oracle_sliced_equals_unsliced():
    from seedvr2_ref import vae as ovae
    cfg = ovae.VAEConfig(block_out_channels=(32, 32, 64, 64))
    p = ovae.synthetic_params(cfg, 0)
    x = torch.randn(3, 9, 16, 24, generator=torch.Generator().manual_seed(1)).clamp(-1, 1)
    m = ovae.encoder(p, x, cfg)
    assert m.shape == (32, 3, 2, 3)
    assert (m - ovae.encode_sliced(p, x, cfg, 4)).abs().max() < 1e-4
    y = ovae.decoder(p, m[:16], cfg)
    assert y.shape == (3, 9, 16, 24)
    assert (y - ovae.decode_sliced(p, m[:16], cfg, 1)).abs().max() < 1e-4


def test_oracle_pipeline_runs_tiny():
    """Config-1-shaped plumbing run of the oracle end to end (tiny dims): shapes, range, determinism."""
    from seedvr2_ref import dit as odit, pipeline as opipe, vae as ovae
    vcfg = ovae.VAEConfig(block_out_channels=(32, 32, 64, 64))
    vp = ovae.synthetic_params(vcfg, 0)
    dcfg = odit.DiTConfig(vid_dim=64, heads=2, head_dim=32, num_layers=2, mm_layers=1, txt_in_dim=32,
                          mlp_multiple_of=32)
    dp = odit.synthetic_params(dcfg, 0)
    frames = torch.rand(3, 16, 24, 3, generator=torch.Generator().manual_seed(2))
    txt = torch.randn(58, 32, generator=torch.Generator().manual_seed(3))
    out = opipe.generate(frames, vp, vcfg, dp, dcfg, txt, resolution=32, batch_size=5, seed=7)
    assert out.shape == (3, 32, 48, 3) and out.min() >= 0 and out.max() <= 1
    out2 = opipe.generate(frames, vp, vcfg, dp, dcfg, txt, resolution=32, batch_size=5, seed=7)
    assert torch.equal(out, out2)


def test_color_oracle_matches_reference_golden():
    """Phase-4 colour correction: the oracle's adain / wavelet / lab against outputs of the
    reference's own color_fix.py (tests/golden/color_pins.safetensors, written by oracle/pin_against_reference.py)."""
    from seedvr2_ref import color as ocol
    t = load_file(os.path.join(ROOT, "tests", "golden", "color_pins.safetensors"))
    for tag, tol in (("f32", 2e-5), ("bf16", 1.6e-2)):      # bf16: one ulp at |v| ~ 1 (conv accumulation order of the host)
        c, s_ = t[f"G12.{tag}.content"], t[f"G12.{tag}.style"]
        for name, fn in (("adain", ocol.adaptive_instance_normalization), ("wavelet", ocol.wavelet_reconstruction),
                         ("lab", ocol.lab_color_transfer), ("hsv", ocol.hsv_saturation_histogram_match),
                         ("wavelet_adaptive", ocol.wavelet_adaptive_color_correction)):
            got = fn(c.clone(), s_.clone()).float()
            ref = t[f"G12.{tag}.{name}"].float()
            assert got.shape == ref.shape
            assert (got - ref).abs().max().item() <= tol, (tag, name)
    lab = ocol.rgb_to_lab((t["G12.f32.style"] + 1) / 2)
    assert (lab - t["G12.lab_of_style"]).abs().max().item() <= 1e-4
    # histogram matching moves the reference's value multiset onto the source ranks
    g = torch.Generator().manual_seed(3)
    src, ref = torch.randn(5, 7, 9, generator=g), torch.randn(5, 7, 9, generator=g) * 2 + 1
    out = ocol.histogram_match(src, ref)
    assert torch.equal(out.flatten().sort().values, ref.flatten().sort().values)
    assert torch.equal(out.flatten().argsort(), src.flatten().argsort())


def test_oracle_noise_injection_semantics():
    """input noise / latent blur of the oracle: identity at scale 0, the reference's dtype flow (bf16 in, bf16 out, fp32
    timestep), the (h, w, 16)-as-(frames, height, width) quirk of the shift, and schedule.forward's lerp."""
    from seedvr2_ref import pipeline as opipe
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 5, 8, 8, generator=g).to(torch.bfloat16)
    n = torch.randn(3, 5, 8, 8, generator=g).to(torch.bfloat16)
    y = opipe.apply_input_noise(x, n, 0.5)
    assert y.dtype == torch.bfloat16 and (y.float() - x.float()).abs().max() < 0.25 * 0.05 * 5 + 0.02
    lat = torch.randn(2, 4, 6, 16, generator=g).to(torch.bfloat16)
    b1, b2 = (torch.randn(2, 4, 6, 16, generator=g).to(torch.bfloat16) for _ in range(2))
    assert opipe.latent_blur(lat, b1, b2, 0.0) is lat
    t = opipe.blur_timestep(lat.shape, 0.3)
    assert t.dtype == torch.float32 and t.shape == (1,)
    # shift from (frames, h, w) = ((4-1)*4+1, 6*8, 16*8): vid branch
    m = (5.0 - 1.0) / (1280 * 720 * 145 - 256 * 256 * 37)
    shift = m * (48 * 128 * 13) + (1.0 - m * 256 * 256 * 37)
    t0 = float((torch.tensor([1000.0], dtype=torch.bfloat16) * 0.3) / 1000.0)     # still bf16 here: 0.30078125
    assert t0 == 0.30078125
    assert abs(float(t) / 1000.0 - shift * t0 / (1 + (shift - 1) * t0)) < 1e-6
    out = opipe.latent_blur(lat, b1, b2, 0.3)
    aug = b1 * 0.1 + b2 * 0.05
    want = ((1 - t / 1000.0) * lat.float() + (t / 1000.0) * aug.float()).to(torch.bfloat16)
    assert out.dtype == torch.bfloat16 and torch.equal(out, want)


def test_piecewise_executor_of_the_oracle_decoder_matches_the_plain_oracle():
    """tests/piecewise_oracle.py runs the oracle's decoder one latent frame per chunk with library calls on pieces (one output
    frame x a strip of rows per conv3d call, one frame per group_norm call): same result as the plain oracle to fp32 rounding
    (limits small enough to cut every conv); in bf16 the two are bf16 noise apart."""
    import piecewise_oracle
    from seedvr2_ref import vae as ovae
    cfg = ovae.VAEConfig()
    p = ovae.synthetic_params(cfg, 0)
    z = torch.randn(2, 4, 6, 16, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        ref = ovae.decode(p, z, cfg, torch.float32).permute(1, 2, 3, 0)
        got = piecewise_oracle.decode_chunked(p, z, ovae, torch.float32, max_elems=20000)
        ref16 = ovae.decode(p, z, cfg, torch.bfloat16).float().permute(1, 2, 3, 0)
        got16 = piecewise_oracle.decode_chunked(p, z, ovae, torch.bfloat16, max_elems=20000).float()
    assert got.shape == ref.shape == (5, 32, 48, 3)
    assert (got - ref).abs().max() <= 2e-5 * ref.abs().max()
    assert (got16 - ref16).pow(2).mean().sqrt() <= 1.5 * (ref16 - ref).pow(2).mean().sqrt()
    assert ovae.F is torch.nn.functional        # the executor restores the module
    x = torch.rand(3, 5, 32, 48, generator=torch.Generator().manual_seed(1)) * 2 - 1
    with torch.no_grad():
        e_ref = ovae.encode(p, x, cfg, torch.float32)
        e_got = piecewise_oracle.encode_piecewise(p, x, ovae, torch.float32, max_elems=20000)
    assert e_got.shape == e_ref.shape == (2, 4, 6, 16)
    assert (e_got - e_ref).abs().max() <= 2e-5 * e_ref.abs().max()
