"""Helpers shared by the GPU parity tests and __graft_entry__.smoke() (checker side: uses the oracle)."""

from __future__ import annotations

import torch


def bf(x: torch.Tensor) -> torch.Tensor:
    """Round to bf16 and return as fp32 (what the device kernels see)."""
    return x.to(torch.bfloat16).float()


def assert_close(got: torch.Tensor, ref: torch.Tensor, name: str, rtol: float = 1.6e-2, atol_rel: float = 1.6e-2):
    """|got - ref| <= atol_rel * rms(ref) + rtol * |ref|  (bf16 output rounding is 2^-8 = 0.4 % per element)."""
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(got).all(), f"{name}: non-finite values"
    rms = ref.pow(2).mean().sqrt().item() + 1e-12
    err = (got - ref).abs()
    bound = atol_rel * rms + rtol * ref.abs()
    bad = err > bound
    if bad.any():
        i = int(torch.argmax(err - bound))
        raise AssertionError(
            f"{name}: {int(bad.sum())}/{bad.numel()} elements out of tolerance; worst |d|={err.flatten()[i]:.4g} "
            f"ref={ref.flatten()[i]:.4g} rms={rms:.4g} max|d|/rms={err.max().item() / rms:.4g}")
    return err.max().item() / rms


def psnr(a: torch.Tensor, b: torch.Tensor, peak: float = 1.0) -> float:
    mse = (a.float() - b.float()).pow(2).mean().item()
    return 99.0 if mse == 0 else 10.0 * torch.log10(torch.tensor(peak * peak / mse)).item()


def smoke_check(pkg):
    """One tiny end-to-end DiT step on cuda:0, compared with the oracle (fp32)."""
    from seedvr2_ref import dit as odit
    dev = torch.device("cuda:0")
    ocfg = odit.DiTConfig(vid_dim=256, heads=2, head_dim=128, num_layers=2, mm_layers=1, txt_in_dim=64,
                          mlp_multiple_of=64)
    p = odit.synthetic_params(ocfg, 1)
    p = {k: bf(v) for k, v in p.items()}
    cfg = pkg.dit.DiTConfig(vid_dim=256, heads=2, head_dim=128, num_layers=2, mm_layers=1, txt_in_dim=64,
                            mlp_multiple_of=64)
    model = pkg.dit.NaDiTHip(cfg, p, dev)
    g = torch.Generator().manual_seed(0)
    T, H, W = 2, 16, 24
    noise, latent = bf(torch.randn(T, H, W, 16, generator=g)), bf(torch.randn(T, H, W, 16, generator=g))
    txt = bf(torch.randn(58, 64, generator=g))
    got = model.one_step(noise.to(dev).bfloat16(), latent.to(dev).bfloat16(), txt.to(dev))
    torch.cuda.synchronize()
    ref = odit.one_step_sr(p, ocfg, latent, noise, txt)
    r = assert_close(got, ref, "smoke one_step", rtol=3e-2, atol_rel=3e-2)
    print(f"[smoke] DiT one-step vs oracle: max|d|/rms = {r:.4f}  OK")
