"""The oracle's MODEL-LEVEL forwards against vectors produced by the reference's own model files
(oracle/pin_models.py: attn_video_vae.py, dit_3b/nadit.py, dit_7b/nadit.py imported and run as they are with the
stand-ins of oracle/ref_standins.py; generation_utils.py helpers taken from the syntax tree).  No reference at
test time: fixtures are inputs, (small-model) weights and the reference's outputs."""

import json
import os

import pytest
import torch
from safetensors.torch import load_file

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def meta():
    with open(os.path.join(GOLD, "model_pins.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="module")
def vgold():
    return load_file(os.path.join(GOLD, "vae_model_pins.safetensors"))


@pytest.fixture(scope="module")
def dgold():
    return load_file(os.path.join(GOLD, "dit_model_pins.safetensors"))


def fingerprint(params):
    return float(sum(v.double().abs().sum().item() * (1 + (i % 7)) for i, (k, v) in enumerate(sorted(params.items()))))


def close(a, b, tol):
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a.double() - b.double()).abs().max().item()
    assert err <= tol, err
    return err


# ------------------------------------------------------------------------------------------------ G16: the VAE model
def test_oracle_vae_model_matches_reference_module(vgold, meta):
    """Encoder3D / Decoder3D / slicing_encode / slicing_decode of the reference (attn_video_vae.py:808-856,
    983-1035, 1246-1292) vs the oracle, on the small seeded model stored in the fixture."""
    from seedvr2_ref import vae as ovae
    m = meta["G16"]
    cfg = ovae.VAEConfig(block_out_channels=tuple(m["block_out_channels"]), norm_groups=m["norm_groups"],
                         scaling_factor=m["scaling_factor"])
    p = {k[len("G16.p."):]: v.float() for k, v in vgold.items() if k.startswith("G16.p.")}
    assert set(p) == set(ovae.param_shapes(cfg)) and all(tuple(p[k].shape) == s for k, s in ovae.param_shapes(cfg).items())
    x, mom, z, dec = (vgold[f"G16.{k}"] for k in ("x", "moments", "z", "dec"))
    with torch.no_grad():
        close(ovae.encoder(p, x, cfg), mom, 3e-5)
        close(ovae.encode_sliced(p, x, cfg, split=m["slicing"]["split_size"]), mom, 3e-5)
        close(ovae.encode(p, x, cfg), (z * cfg.scaling_factor).permute(1, 2, 3, 0), 3e-5)
        close(ovae.decoder(p, z, cfg), dec, 5e-5)
        close(ovae.decode_sliced(p, z, cfg, split=1), dec, 5e-5)
        close(ovae.decode(p, (z * cfg.scaling_factor).permute(1, 2, 3, 0), cfg), dec, 5e-5)


def test_vae_full_width_key_layout_is_the_reference_modules(pkg, meta):
    """State-dict keys / shapes of the full-width reference VAE == the oracle's and the product's parameter layout."""
    from seedvr2_ref import vae as ovae
    want = {k: tuple(v) for k, v in meta["G16.keys_fullwidth"].items()}
    assert {k: tuple(v) for k, v in ovae.param_shapes(ovae.VAEConfig()).items()} == want
    small = pkg.vae.VAEConfig(block_out_channels=(32, 64, 64, 64))
    sd = pkg.vae.synthetic_state_dict(small, "cpu", seed=0)
    osd = ovae.param_shapes(ovae.VAEConfig(block_out_channels=(32, 64, 64, 64)))
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(v) for k, v in osd.items()}


# ------------------------------------------------------------------------------------------------ G17: NaDiT.forward
@pytest.mark.parametrize("model", ["3b", "7b"])
def test_oracle_nadit_forward_matches_reference_module(dgold, meta, model):
    """The whole NaDiT.forward (dit_3b/nadit.py:190-248, dit_7b/nadit.py:153-190) with RoPE on."""
    from seedvr2_ref import dit as odit
    m = meta[f"G17.{model}"]
    c = m["cfg"]
    if model == "3b":
        cfg = odit.DiTConfig(vid_dim=c["vid_dim"], heads=c["heads"], head_dim=c["head_dim"], num_layers=c["num_layers"],
                             mm_layers=c["mm_layers"], txt_in_dim=c["txt_in_dim"])
    else:
        cfg = odit.DiTConfig(vid_dim=c["vid_dim"], heads=c["heads"], head_dim=c["head_dim"], num_layers=c["num_layers"],
                             mm_layers=c["num_layers"], txt_in_dim=c["txt_in_dim"], mlp_type="normal", out_norm=False,
                             last_vid_only=False)
    pre = f"G17.{model}.p."
    p = {k[len(pre):]: v.float() for k, v in dgold.items() if k.startswith(pre)}
    assert {k: tuple(v.shape) for k, v in p.items()} == {k: tuple(s) for k, s in odit.param_shapes(cfg).items()}
    vid, txt, out = (dgold[f"G17.{model}.{k}"] for k in ("vid", "txt", "out"))
    with torch.no_grad():
        mine = odit.forward(p, cfg, vid, txt, tuple(m["latent_thw"]), 1000.0, torch.float32, m["rope"])
        close(mine, out, 5e-5)
        wrong = odit.forward(p, cfg, vid, txt, tuple(m["latent_thw"]), 1000.0, torch.float32, "none")
    assert (wrong - out).abs().max() > 1e-3      # the vector is sensitive to the rotary embedding


def test_nadit_full_size_key_layout(meta):
    from seedvr2_ref import dit as odit
    for model, cfg in (("3b", odit.CFG_3B), ("7b", odit.CFG_7B)):
        shapes = odit.param_shapes(cfg)
        n = sum(int(torch.tensor(s).prod()) for s in shapes.values())
        assert len(shapes) == meta[f"G17.fullsize_{model}"]["keys"] and n == meta[f"G17.fullsize_{model}"]["params"]
    part = {k: tuple(v) for k, v in meta["G17.keys_3b"].items()}
    full = odit.param_shapes(odit.CFG_3B)
    assert all(tuple(full[k]) == v for k, v in part.items()) and len(part) > 60


# ------------------------------------------------------------------------------------------------ G16h / G17h
def test_hip_sized_reference_outputs_match_oracle(vgold, dgold, meta):
    """The fixtures the GPU tests compare the HIP path with (reference outputs at head_dim 128 / VAE 64-128-512-512,
    seeded weights regenerated here): the weights regenerate bit-identically and the oracle reproduces the outputs."""
    from seedvr2_ref import dit as odit, vae as ovae
    m = meta["G16h"]
    vcfg = ovae.VAEConfig(block_out_channels=tuple(m["block_out_channels"]))
    vp = {k: v.to(torch.bfloat16).float() for k, v in ovae.synthetic_params(vcfg, m["param_seed"]).items()}
    assert fingerprint(vp) == m["fingerprint"]
    with torch.no_grad():
        close(ovae.encoder(vp, vgold["G16h.x"], vcfg)[:16], vgold["G16h.z"], 5e-5)
        close(ovae.decode(vp, vgold["G16h.latent_in"].permute(1, 2, 3, 0), vcfg), vgold["G16h.dec"], 1e-4)
    for model in ("3b", "7b"):
        m = meta[f"G17h.{model}"]
        cfg = odit.DiTConfig(**m["cfg"])
        p = {k: v.to(torch.bfloat16).float() for k, v in odit.synthetic_params(cfg, m["param_seed"]).items()}
        assert fingerprint(p) == m["fingerprint"]
        noise, latent, txt, pred = (dgold[f"G17h.{model}.{k}"] for k in ("noise", "latent", "txt", "pred"))
        with torch.no_grad():
            x0 = odit.one_step_sr(p, cfg, latent, noise, txt, torch.float32, m["rope"])
        close(x0.reshape(-1, 16), noise.reshape(-1, 16) - pred, 1e-4)


# ------------------------------------------------------------------------------------------------ G10b: host helpers
def test_pad_video_temporal_table(pkg, meta):
    """pad_video_temporal (generation_utils.py:593-652) run as is: 23 lengths x 8 counts (append) and the prepend
    flavour -> the oracle, the product's batch planner and the node's prepend helper."""
    from seedvr2_ref import pipeline as opipe
    table = meta["G10b.pad_video_temporal"]
    n = 0
    for key, want in table.items():
        parts = key.split("|")
        if parts[0] == "prepend":
            t, count = int(parts[1]), int(parts[2])
            x = torch.arange(float(t)).view(t, 1, 1, 1)
            assert pkg.nodes.SeedVR2VideoUpscaler.prepend_reversed(x, count).view(-1).long().tolist() == want
        else:
            t, count = int(parts[0]), int(parts[1])
            v = torch.arange(float(t)).view(t, 1, 1, 1)
            assert opipe.pad_video_temporal(v, count).view(-1).long().tolist() == want
            if count == 0:      # the 4n+1 rule as the product's planner applies it to one batch
                b, _ = pkg.pipeline.plan_batches(t, t, 0)
                assert b[0].index == want
            else:               # uniform_batch_size padding: batch_size = t + count, then the 4n+1 rule
                want_u = opipe.pad_video_temporal(opipe.pad_video_temporal(v, count)).view(-1).long().tolist()
                b, _ = pkg.pipeline.plan_batches(t, t + count, 0, uniform=True)
                assert b[0].index == want_u
        n += 1
    assert n == 23 * 8 + 23 * 3


def test_blend_overlapping_frames_vectors(pkg, vgold):
    """blend_overlapping_frames (generation_utils.py:284-312) run as is, fp32 and bf16: oracle bit-exact; the product's
    weight table is the same bf16 window."""
    from seedvr2_ref import pipeline as opipe
    for ov in (1, 2, 3, 4, 5, 8):
        for tag in ("f32", "bf16"):
            a, b, out = (vgold[f"G10b.blend.{ov}.{tag}.{k}"] for k in ("a", "b", "out"))
            assert torch.equal(opipe.blend_overlapping_frames(a, b, ov), out)
        w = pkg.pipeline.blend_weights(ov).view(ov, 1, 1, 1)
        a, b, out = (vgold[f"G10b.blend.{ov}.bf16.{k}"] for k in ("a", "b", "out"))
        wb = w.to(torch.bfloat16)
        assert torch.equal(a * wb + b * (1.0 - wb), out)


def test_side_resize_two_pass_vectors(pkg, vgold, meta):
    """SideResize + clamp + DivisiblePad + Normalize (side_resize.py:36-75 incl. the second resize when max_size caps
    the long edge; divisible_crop.py:43-80) run as they are -> oracle.transform bit-exact; product geometry identical."""
    from seedvr2_ref import pipeline as opipe
    for i, c in enumerate(meta["G10c"]):
        x, y = vgold[f"G10c.{i}.x"], vgold[f"G10c.{i}.y"]
        assert torch.equal(opipe.transform(x, c["resolution"], c["max_resolution"]), y)
        _, h, w = c["in"]
        rh, rw, _, _, ph, pw = pkg.pipeline.target_dims(h, w, c["resolution"], c["max_resolution"])
        assert [rh, rw] == c["resized"] and [ph, pw] == c["padded"]
        capped = pkg.pipeline.uncapped_dims(h, w, c["resolution"]) != (rh, rw)
        assert capped == (c["max_resolution"] > 0 and max(pkg.pipeline.uncapped_dims(h, w, c["resolution"])) > c["max_resolution"])
