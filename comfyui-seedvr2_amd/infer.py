"""``VideoDiffusionInfer`` - the runner seam of the reference, backed by the HIP models.

Same methods, argument meaning and tensor layouts as the reference class
[ref: src/core/infer.py:33-374]: ``vae_encode(List[c t h w]) -> List[t h w c]``,
``get_condition``, ``inference(noises, conditions, texts_pos, texts_neg)``, ``vae_decode(List[t h w c]) ->
List[c t h w]``, ``configure_diffusion``.  The sampler is the one-step Euler endpoint that the reference's
phases force (steps=1, cfg scale 1 -> only the positive text branch runs)
[ref: src/core/generation_phases.py:595-598; common/diffusion/samplers/euler.py:36-66; utils.py:65-84].
"""

from __future__ import annotations

from typing import List, Optional

import torch

from . import hip


class VideoDiffusionInfer:
    def __init__(self, config=None, debug=None, encode_tiled: bool = False, encode_tile_size=(512, 512),
                 encode_tile_overlap=(64, 64), decode_tiled: bool = False, decode_tile_size=(512, 512),
                 decode_tile_overlap=(64, 64), tile_debug: str = "false"):
        """Same constructor arguments as the reference runner [ref: src/core/infer.py:37-52]."""
        self.config = config
        self.debug = debug
        self.vae = None   # VideoVAEHip
        self.dit = None   # NaDiTHip
        self.encode_tiled, self.encode_tile_size, self.encode_tile_overlap = encode_tiled, encode_tile_size, encode_tile_overlap
        self.decode_tiled, self.decode_tile_size, self.decode_tile_overlap = decode_tiled, decode_tile_size, decode_tile_overlap
        if str(tile_debug).lower() != "false":
            raise NotImplementedError("tile_debug overlays are a debugging aid of the reference UI, not built here")
        self.schedule_T = 1000.0
        self.steps = 1

    # ---- reference API -----------------------------------------------------------------------
    def get_condition(self, latent: torch.Tensor, latent_blur: torch.Tensor, task: str) -> torch.Tensor:
        """[ref: infer.py:54-78] only the 'sr' task is on the hot path: cond = [latent_blur, 1]."""
        if task != "sr":
            raise NotImplementedError(f"task {task!r}: the SeedVR2 upscaler only uses 'sr'")
        t, h, w, c = latent.shape
        cond = torch.ones(t, h, w, c + 1, device=latent.device, dtype=latent.dtype)
        cond[..., :-1] = latent_blur
        return cond

    def configure_diffusion(self, device=None, dtype=torch.float32):
        """[ref: infer.py:80-111] lerp schedule T=1000, uniform-trailing timesteps; steps forced to 1."""
        self.schedule_T, self.steps = 1000.0, 1

    @torch.no_grad()
    def vae_encode(self, samples: List[torch.Tensor]) -> List[torch.Tensor]:
        out = []
        for s in samples:  # c t h w in [-1, 1]
            x = s.permute(1, 2, 3, 0).to(self.vae.device).to(torch.bfloat16).contiguous()
            out.append(self.vae.encode(x, tiled=self.encode_tiled, tile_size=tuple(self.encode_tile_size),
                                       tile_overlap=tuple(self.encode_tile_overlap)))
        return out

    @torch.no_grad()
    def vae_decode(self, latents: List[torch.Tensor]) -> List[torch.Tensor]:
        out = []
        for z in latents:  # t h w c
            y = self.vae.decode(z.to(self.vae.device).to(torch.bfloat16).contiguous(), tiled=self.decode_tiled,
                                tile_size=tuple(self.decode_tile_size), tile_overlap=tuple(self.decode_tile_overlap))
            out.append(y.permute(3, 0, 1, 2))  # c t h w (view)
        return out

    @torch.no_grad()
    def inference(self, noises: List[torch.Tensor], conditions: List[torch.Tensor], texts_pos, texts_neg=None,
                  cfg_scale: Optional[float] = None) -> List[torch.Tensor]:
        assert len(noises) == len(conditions) == len(texts_pos)
        if cfg_scale not in (None, 1.0):
            raise NotImplementedError("classifier-free guidance != 1 is never used by the one-step upscaler")
        out = []
        for noise, cond, txt in zip(noises, conditions, texts_pos):
            c = noise.shape[-1]
            if cond.shape[-1] != c + 1:
                raise hip.HipError("condition must be [t,h,w,c+1] (latent + mask channel)")
            dev = self.dit.device
            latent = cond[..., :c].to(dev).to(torch.bfloat16).contiguous()
            out.append(self.dit.one_step(noise.to(dev).to(torch.bfloat16).contiguous(), latent, txt))
        return out
