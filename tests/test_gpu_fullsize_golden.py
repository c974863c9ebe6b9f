"""At-size parity inside the collected GPU suite: the HIP path at TRUE sizes against crops of the CPU oracle's outputs
that tests/fullsize_golden.py wrote (minutes of oracle time per case on the GPU box's host cores; stored under
tests/golden/fullsize/ as data: crops, per-frame means, what the HIP path measured when the golden was written).

  cfg2   BASELINE configs[1] at its true size, [8,540,960,3] -> 1080p, batch_size 9 (4n+1 padding), NaDiT-3B + VAE
  4k3b   one 4K frame through the assembled pipeline (every kernel at its 4K shape), NaDiT-3B
  4k7b   the same with NaDiT-7B (cfg4's model at cfg4's shapes)
  dit4k  3-layer full-width NaDiT (separate / shared / vid-only last; regular, shifted, regular windows) on cfg3's grid
         (6,135,240): the only grid with two temporal windows (810-token windows) [ref: dit_3b/window.py:28-83]
  vae4k_enc / vae4k_dec   5 frames of 4K through the VAE encoder (chunks 1 + 4) / decoder (one latent frame per chunk):
         causal carries across temporal chunks at 4K [ref: causal_inflation_lib.py:250-305]

Tolerance (floating point, stated here): on the crops, PSNR(hip, oracle) >= recorded - 0.3 dB, max-abs <= 1.25 x recorded,
relative rms error of latents <= 1.1 x recorded; per-frame channel means of the WHOLE output within 1.5 x recorded + 1e-4.
"Recorded" = what this same path measured against the full oracle output when the golden was written (<case>.json).
Reference path: src/core/generation_phases.py:342-521, 650-753, 904-1026.
"""

import pytest
import torch

import fullsize_cases as fc

pytestmark = pytest.mark.gpu


def _golden(case):
    gold, meta = fc.load_golden(case)
    assert gold is not None, f"tests/golden/fullsize/{case}.* missing (written by tests/fullsize_golden.py {case})"
    return gold, meta


def _check_weights(meta, **sds):
    for name, sd in sds.items():
        want = meta["weights"][name]
        got = fc.fingerprint(sd)
        for k, (s, a) in want.items():
            assert abs(got[k][0] - s) <= 1e-3 * max(abs(a), 1.0) and abs(got[k][1] - a) <= 1e-3 * max(abs(a), 1.0), \
                f"seeded {name} weights differ from the ones the golden was written with ({k})"


def _check(gold, meta, name, got, ref_name, latent_like=False):
    size = meta["crop_size"][ref_name]
    rec = meta["recorded"][name]
    assert list(got.shape) == meta["shape"][ref_name]
    assert torch.isfinite(got).all()
    boxes = [tuple(b) for b in gold[f"{ref_name}.boxes"].tolist()]
    g = fc.take_crops(got, boxes, size).float()
    r = gold[f"{ref_name}.crops"].float()
    maxabs = float((g - r).abs().max())
    means = float((got.float().mean(dim=(1, 2)) - gold[f"{ref_name}.means"]).abs().max())
    msg = f"{meta['case']}/{name}: "
    if latent_like:
        rel = fc.rel_rms(g, r)
        print(msg + f"rel-rms {rel:.4e} (recorded {rec['rel_rms_crops']:.4e}), max|d| {maxabs:.4f} ({rec['maxabs_crops']:.4f})")
        assert rel <= 1.1 * rec["rel_rms_crops"] + 1e-5, msg + f"rel-rms {rel} vs recorded {rec['rel_rms_crops']}"
    else:
        p = fc.psnr(g, r, rec.get("peak", 1.0))
        print(msg + f"PSNR {p:.2f} dB (recorded {rec['psnr_crops']:.2f}), max|d| {maxabs:.4f} ({rec['maxabs_crops']:.4f})")
        assert p >= rec["psnr_crops"] - 0.3, msg + f"PSNR {p} vs recorded {rec['psnr_crops']}"
    assert maxabs <= 1.25 * rec["maxabs_crops"] + 1e-6, msg + f"max-abs {maxabs} vs recorded {rec['maxabs_crops']}"
    assert means <= 1.5 * rec["means_maxabs"] + 1e-4, msg + f"frame means {means} vs recorded {rec['means_maxabs']}"


@pytest.mark.parametrize("case", ["cfg2", "4k3b", "4k7b"])
def test_pipeline_at_true_size_vs_oracle_crops(pkg, gpu, case):
    gold, meta = _golden(case)
    nfr, hin, win, res, bsz, model = fc.PIPELINE_CASES[case]
    vcfg = pkg.vae.VAEConfig()
    dcfg = pkg.dit.CFG_7B if model == "7b" else pkg.dit.CFG_3B
    vsd = pkg.vae.synthetic_state_dict(vcfg, gpu, seed=1)
    dsd = pkg.dit.synthetic_state_dict(dcfg, gpu, seed=2)
    _check_weights(meta, vae=vsd, dit=dsd)
    vae, dit = pkg.vae.VideoVAEHip(vcfg, vsd, gpu), pkg.dit.NaDiTHip(dcfg, dsd, gpu)
    del vsd, dsd
    pipe = pkg.pipeline.SeedVR2Pipeline(vae, dit, fc.text_embedding(), gpu)
    lr = fc.lr_clip(nfr, hin, win)
    got = pipe.generate(lr, resolution=res, batch_size=bsz, temporal_overlap=0, seed=42, color_correction="none").float().cpu()
    _check(gold, meta, "rgb", got, "rgb")
    # stage taps at true size: latent after the encoder, x0 after the one-step DiT
    batches, _ = pkg.pipeline.plan_batches(nfr, bsz, 0)
    vid = lr.to(gpu).to(torch.bfloat16).index_select(0, torch.tensor(batches[0].index, device=gpu))
    lat = vae.encode(pipe.transform(vid, res))
    x0 = dit.one_step(pipe.make_noise(lat.shape, 42).to(gpu), lat, pipe.text_pos)
    _check(gold, meta, "latent", lat.float().cpu(), "latent", latent_like=True)
    _check(gold, meta, "x0", x0.float().cpu(), "x0", latent_like=True)


def test_nadit_full_width_on_the_4k_grid_vs_oracle_crops(pkg, gpu):
    gold, meta = _golden("dit4k")
    cfg = pkg.dit.DiTConfig(num_layers=3, mm_layers=1)
    sd = pkg.dit.synthetic_state_dict(cfg, gpu, seed=4)
    _check_weights(meta, dit=sd)
    model = pkg.dit.NaDiTHip(cfg, sd, gpu)
    del sd
    noise, latent, txt = fc.dit4k_inputs()
    got = model.one_step(noise.to(gpu), latent.to(gpu), txt.to(gpu)).float().cpu()
    _check(gold, meta, "x0_vs_bf16", got, "x0_bf16", latent_like=True)
    _check(gold, meta, "x0_vs_fp32", got, "x0_fp32", latent_like=True)
    # and the HIP path is no farther from exact arithmetic than the reference's own bf16 CPU path is
    rec = meta["recorded"]
    size = meta["crop_size"]["x0_fp32"]
    boxes = [tuple(b) for b in gold["x0_fp32.boxes"].tolist()]
    e_hip = fc.rel_rms(fc.take_crops(got, boxes, size), gold["x0_fp32.crops"])
    e_b16 = fc.rel_rms(gold["x0_bf16.crops"], gold["x0_fp32.crops"])
    print(f"dit4k: rel-rms vs fp32 oracle: hip {e_hip:.4e}, bf16 oracle {e_b16:.4e}")
    assert e_hip <= e_b16 * 1.02, (e_hip, e_b16, rec)


def test_vae_encoder_4k_clip_vs_oracle_crops(pkg, gpu):
    gold, meta = _golden("vae4k_enc")
    nfr, hin, win, res = fc.VAE4K
    vcfg = pkg.vae.VAEConfig()
    vsd = pkg.vae.synthetic_state_dict(vcfg, gpu, seed=1)
    _check_weights(meta, vae=vsd)
    vae = pkg.vae.VideoVAEHip(vcfg, vsd, gpu)
    del vsd
    pipe = pkg.pipeline.SeedVR2Pipeline(vae, None, fc.text_embedding(), gpu)
    lat = vae.encode(pipe.transform(fc.lr_clip(nfr, hin, win).to(gpu).to(torch.bfloat16), res)).float().cpu()
    _check(gold, meta, "latent", lat, "latent", latent_like=True)


def test_vae_decoder_4k_clip_vs_oracle_crops(pkg, gpu):
    gold, meta = _golden("vae4k_dec")
    vcfg = pkg.vae.VAEConfig()
    vsd = pkg.vae.synthetic_state_dict(vcfg, gpu, seed=1)
    _check_weights(meta, vae=vsd)
    vae = pkg.vae.VideoVAEHip(vcfg, vsd, gpu)
    del vsd
    got = vae.decode(fc.vae4k_decode_latent().to(gpu)).float().cpu()
    _check(gold, meta, "dec", got, "dec")
