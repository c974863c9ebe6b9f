"""GPU parity tests of the spatial VAE tiling: the blend kernels reproduce, bit for bit, the vectors
of the reference's own tiled_encode / tiled_decode (bf16), and the tiled HIP VAE equals the oracle's tiling loop wrapped
around the same HIP tile codec."""

import json
import os

import pytest
import torch
from safetensors import safe_open

from seedvr2_ref import tiling as otile

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiling_pins.safetensors")


def bf(x):
    return x.to(torch.bfloat16)


def dev_bf16(x, gpu):
    return x.to(gpu).to(torch.bfloat16).contiguous()


def _cl(x):      # oracle layout [C,T,H,W] -> product layout [T,H,W,C]
    return x.permute(1, 2, 3, 0).contiguous()


def _cf(x):
    return x.permute(3, 0, 1, 2).contiguous()


@pytest.fixture(scope="module")
def model(pkg, gpu):
    cfg = pkg.vae.VAEConfig()
    return pkg.vae.VideoVAEHip(cfg, pkg.vae.synthetic_state_dict(cfg, gpu, seed=3), gpu)


def test_blend_kernels_reproduce_reference_vectors(pkg, gpu, model):
    """The product's tiling loop + svr2_tile_accumulate / svr2_tile_normalize around the SAME synthetic tile codec the
    reference methods were run with: every bf16 vector of G13 is reproduced exactly."""
    with safe_open(GOLD, "pt") as f:
        cases = json.loads(f.metadata()["cases"])
        t = {k: f.get_tensor(k) for k in f.keys()}
    enc = lambda tile: _cl(otile.synthetic_enc_codec(_cf(tile.cpu()))).to(gpu)
    dec = lambda tile: _cl(otile.synthetic_dec_codec(_cf(tile.cpu()))).to(gpu)
    for i in range(6):
        c = cases[f"enc.{i}"]
        got = model._tiled(_cl(t[f"G13.enc.bf16.{i}.x"]).to(gpu), enc, tuple(c["tile"]), tuple(c["overlap"]), 8, 1)
        assert torch.equal(_cf(got.cpu()), t[f"G13.enc.bf16.{i}.out"]), ("enc", i)
        c = cases[f"dec.{i}"]
        got = model._tiled(_cl(t[f"G13.dec.bf16.{i}.z"]).to(gpu), dec, tuple(c["tile"]), tuple(c["overlap"]), 1, 8)
        assert torch.equal(_cf(got.cpu()), t[f"G13.dec.bf16.{i}.out"]), ("dec", i)


@pytest.mark.parametrize("tile,ov", [((32, 32), (8, 8)), ((24, 40), (16, 8))])
def test_tiled_vae_encode_equals_oracle_loop(pkg, gpu, model, tile, ov):
    x = bf(torch.rand(5, 48, 64, 3, generator=torch.Generator().manual_seed(11)) * 2 - 1)
    got = model.encode(dev_bf16(x, gpu), tiled=True, tile_size=tile, tile_overlap=ov)
    codec = lambda v: _cf(model._encode_mean(_cl(v).to(gpu)).cpu())           # the HIP encoder as the oracle's tile codec
    want = otile.tiled_encode(_cf(x), codec, tile, ov)
    want = (_cl(want).float() * model.cfg.scaling_factor).to(torch.bfloat16)
    assert got.shape == (2, 6, 8, 16) and torch.equal(got.cpu(), want)
    whole = model.encode(dev_bf16(x, gpu))
    assert not torch.equal(got, whole)                                         # tiling is approximate by design ...
    rel = ((got.float() - whole.float()).pow(2).mean() / whole.float().pow(2).mean()).sqrt().item()
    print(f"tiled vs untiled encode, tile {tile}: rel-rms {rel:.3e}")
    assert model.encode(dev_bf16(x, gpu), tiled=True, tile_size=(64, 64), tile_overlap=ov).equal(whole)   # one tile: untiled


@pytest.mark.parametrize("tile,ov", [((32, 32), (8, 8)), ((24, 40), (13, 20))])
def test_tiled_vae_decode_equals_oracle_loop(pkg, gpu, model, tile, ov):
    z = bf(torch.randn(2, 6, 8, 16, generator=torch.Generator().manual_seed(12)))
    got = model.decode(dev_bf16(z, gpu), tiled=True, tile_size=tile, tile_overlap=ov)
    zs = (z.float() / model.cfg.scaling_factor).to(torch.bfloat16)
    codec = lambda v: _cf(model._decode_frames(_cl(v).to(gpu)).cpu())
    want = _cl(otile.tiled_decode(_cf(zs), codec, tile, ov))
    assert got.shape == (5, 48, 64, 3) and torch.equal(got.cpu(), want)
    assert model.decode(dev_bf16(z, gpu), tiled=True, tile_size=(64, 64), tile_overlap=ov).equal(model.decode(dev_bf16(z, gpu)))


def test_pipeline_and_runner_pass_the_tiling_flags(pkg, gpu, model):
    """Node / CLI settings reach the VAE: a tiled pipeline run equals the explicit tiled calls."""
    calls = []
    real_e, real_d = model.encode, model.decode

    class Spy:
        cfg, device = model.cfg, model.device

        def encode(self, x, **kw):
            calls.append(("e", kw))
            return real_e(x, **kw)

        def decode(self, z, **kw):
            calls.append(("d", kw))
            return real_d(z, **kw)

    class Dit:
        def one_step(self, noise, latent, txt):
            return latent

    pipe = pkg.pipeline.SeedVR2Pipeline(Spy(), Dit(), torch.zeros(58, 5120), gpu)
    vae_cfg, = pkg.nodes.SeedVR2LoadVAEModel.execute(encode_tiled=True, encode_tile_size=32, encode_tile_overlap=8,
                                                     decode_tiled=True, decode_tile_size=40, decode_tile_overlap=16)
    lr = torch.rand(1, 24, 32, 3, generator=torch.Generator().manual_seed(13))
    out = pkg.nodes.SeedVR2VideoUpscaler.run_device(lr, {}, vae_cfg, 1, 48, 0, 1, False, 0, 0, "none", pipeline=pipe)
    assert out.shape == (1, 48, 64, 3)
    assert calls[0] == ("e", dict(tiled=True, tile_size=(32, 32), tile_overlap=(8, 8)))
    assert calls[1] == ("d", dict(tiled=True, tile_size=(40, 40), tile_overlap=(16, 16)))
    r = pkg.infer.VideoDiffusionInfer(decode_tiled=True, decode_tile_size=(40, 40), decode_tile_overlap=(16, 16))
    r.vae = model
    z = bf(torch.randn(1, 6, 8, 16, generator=torch.Generator().manual_seed(14)))
    got = r.vae_decode([z])[0]
    assert torch.equal(got.permute(1, 2, 3, 0), model.decode(dev_bf16(z, gpu), tiled=True, tile_size=(40, 40), tile_overlap=(16, 16)))


def test_tile_kernels_reject_bad_arguments(pkg, gpu):
    hip = pkg.hip
    z = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=gpu)
    with pytest.raises(hip.HipError, match="leaves the"):
        hip.tile_accumulate(z(1, 4, 4, 2), z(1, 6, 6, 2), z(6, 6), 3, 3, 4, 4, z(4), z(4))
    with pytest.raises(hip.HipError):
        hip.tile_accumulate(z(1, 4, 4, 2), z(2, 6, 6, 2), z(6, 6), 0, 0, 4, 4, z(4), z(4))        # T mismatch
    with pytest.raises(hip.HipError):
        hip.tile_accumulate(z(1, 4, 4, 2), z(1, 6, 6, 2), z(6, 5), 0, 0, 4, 4, z(4), z(4))        # count shape
    with pytest.raises(hip.HipError):
        hip.tile_normalize(z(1, 6, 6, 2).float(), z(6, 6))
