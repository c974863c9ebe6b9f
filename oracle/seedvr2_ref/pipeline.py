"""End-to-end CPU restatement of the reference's generation phases for one clip (oracle).

Follows /root/reference/src/core/generation_phases.py:171-521, :538-801, :802-1047, :1049-1362 with
color_correction='none', and generation_utils.py:47-84,284-312,593-652.  These modules cannot be imported
here (they import torchvision / omegaconf / diffusers), so the arithmetic is restated by reading them;
the antialiased bicubic resize itself is torch's own op (F.interpolate(..., antialias=True), which is what
torchvision.transforms.functional.resize lowers to for tensors), computed in fp32 like torchvision does
for non-float32 inputs.
"""

from __future__ import annotations

from typing import List, Tuple

import torch
import torch.nn.functional as F

from . import dit as _dit
from . import vae as _vae


def target_dims(h, w, resolution, max_resolution=0):
    """side_resize.py:36-75 (torchvision resize(int): long = int(size*long/short)); generation_utils.py:134-136."""
    if w <= h:
        rw, rh = resolution, int(resolution * h / w)
    else:
        rh, rw = resolution, int(resolution * w / h)
    if max_resolution > 0 and max(rh, rw) > max_resolution:
        s = max_resolution / max(rh, rw)
        rh, rw = round(rh * s), round(rw * s)
    return rh, rw, (rh // 2) * 2, (rw // 2) * 2, -(-rh // 16) * 16, -(-rw // 16) * 16


def transform(frames_tchw_bf16: torch.Tensor, resolution: int, max_resolution: int = 0) -> torch.Tensor:
    """prepare_video_transforms (generation_utils.py:47-84): resize -> clamp -> pad16 -> (x-.5)/.5 -> CTHW.
    Input bf16 [T,3,H,W]; every step rounds to bf16 like the reference's bf16 tensors."""
    T, C, H, W = frames_tchw_bf16.shape
    rh, rw, _, _, ph, pw = target_dims(H, W, resolution, max_resolution)
    rh0, rw0 = target_dims(H, W, resolution, 0)[:2]
    # SideResize (side_resize.py:61-73): resize to the short-edge size; if the long edge then exceeds max_size, resize the
    # RESULT (a bf16 tensor; torchvision computes in fp32 and casts back) a second time to the capped size
    x = F.interpolate(frames_tchw_bf16.float(), size=(rh0, rw0), mode="bicubic", align_corners=False, antialias=True)
    x = x.to(torch.bfloat16)
    if (rh0, rw0) != (rh, rw):
        x = F.interpolate(x.float(), size=(rh, rw), mode="bicubic", align_corners=False, antialias=True).to(torch.bfloat16)
    x = torch.clamp(x, 0.0, 1.0)
    x = F.pad(x, (0, pw - rw, 0, ph - rh), value=0.0)
    x = (x - 0.5) / 0.5
    return x.permute(1, 0, 2, 3)


def pad_video_temporal(v: torch.Tensor, count: int = 0) -> torch.Tensor:
    """generation_utils.py:593-652 for temporal_dim=0, prepend=False."""
    t = v.shape[0]
    if count == 0:
        if t % 4 == 1:
            return v
        count = ((t - 1) // 4 + 1) * 4 + 1 - t
    if count <= 0:
        return v
    if count >= t:
        rep = count - t + 1
        last = v[-1:]
        rev = v[1:].flip(0) if t > 1 else last[:0]
        return torch.cat([v, rev, last.repeat(rep, *([1] * (v.dim() - 1)))], dim=0)
    return torch.cat([v, v[-count - 1:-1].flip(0)], dim=0)


def batch_ranges(total: int, batch_size: int, overlap: int) -> Tuple[List[Tuple[int, int]], int]:
    """generation_phases.py:267-282, 342-355."""
    step = batch_size - overlap if overlap > 0 else batch_size
    if step <= 0:
        step, overlap = batch_size, 0
    out = []
    for s in range(0, total, step):
        e = min(s + batch_size, total)
        if s > 0 and e - s <= overlap:
            break
        out.append((s, e))
    return out, overlap


def blend_overlapping_frames(prev_tail, cur_head, overlap):
    """generation_utils.py:284-312 (runs in the tensors' dtype)."""
    dt = prev_tail.dtype
    if overlap >= 3:
        t = torch.linspace(0.0, 1.0, steps=overlap, dtype=dt)
        u = ((t - 1.0 / 3.0) / (2.0 / 3.0 - 1.0 / 3.0)).clamp(0.0, 1.0)
        w_prev = 0.5 + 0.5 * torch.cos(torch.pi * u)
    else:
        w_prev = torch.linspace(1.0, 0.0, steps=overlap, dtype=dt)
    w_prev = w_prev.view(overlap, 1, 1, 1)
    return prev_tail * w_prev + cur_head * (1.0 - w_prev)


def make_noise(shape, seed):
    """generation_phases.py:659,676 - torch.manual_seed(seed); randn_like(latent) on the CPU, bf16."""
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    return torch.randn(tuple(shape), generator=g, dtype=torch.bfloat16, device="cpu")


def apply_input_noise(x: torch.Tensor, noise: torch.Tensor, scale: float) -> torch.Tensor:
    """generation_phases.py:409-425: x = transformed video (bf16), noise = randn_like(x)."""
    noise = noise * 0.05
    blend_factor = scale * 0.5
    return x * (1 - blend_factor) + (x + noise) * blend_factor


def timestep_transform(timesteps: torch.Tensor, latents_shapes: torch.Tensor, T: float = 1000.0) -> torch.Tensor:
    """core/infer.py:260-290 with `timesteps.transform: True` (configs_3b/main.yaml:80), vt = 4, vs = 8."""
    frames = (latents_shapes[:, 0] - 1) * 4 + 1
    heights = latents_shapes[:, 1] * 8
    widths = latents_shapes[:, 2] * 8

    def lin(x1, y1, x2, y2):
        m = (y2 - y1) / (x2 - x1)
        b = y1 - m * x1
        return lambda x: m * x + b

    img, vid = lin(256 * 256, 1.0, 1024 * 1024, 3.2), lin(256 * 256 * 37, 1.0, 1280 * 720 * 145, 5.0)
    shift = torch.where(frames > 1, vid(heights * widths * frames), img(heights * widths))
    timesteps = timesteps / T
    timesteps = shift * timesteps / (1 + (shift - 1) * timesteps)
    return timesteps * T


def latent_blur(latent: torch.Tensor, base_noise: torch.Tensor, second_noise: torch.Tensor, scale: float,
                T: float = 1000.0) -> torch.Tensor:
    """generation_phases.py:676-701: aug noise = base*0.1 + randn*0.05; t = 1000*scale through timestep_transform with
    the latent's shape[1:] -- the latent is channels-last [T_l,h,w,16] there, so the reference feeds (h, w, 16) where
    (frames, height, width) are expected; restated as it is -- then schedule.forward (schedules/base.py:82-87) and the
    cast into the bf16 condition tensor (infer.py:54-78)."""
    if scale == 0.0:
        return latent
    aug = base_noise * 0.1 + second_noise * 0.05
    t = blur_timestep(latent.shape, scale, latent.dtype, T)
    t = t.view(-1, *([1] * (latent.ndim - 1)))                       # expand_dims(t, x_0.ndim)
    return ((1 - (t / T)) * latent + (t / T) * aug).to(latent.dtype)


def blur_timestep(latent_shape, scale: float, dtype=torch.bfloat16, T: float = 1000.0) -> torch.Tensor:
    """The (shifted) timestep of latent_blur: fp32 [1]."""
    t = torch.tensor([1000.0], dtype=dtype) * scale
    return timestep_transform(t, torch.tensor(tuple(latent_shape[1:]))[None], T)


def generate(frames: torch.Tensor, vae_p, vae_cfg, dit_p, dit_cfg, text_pos, resolution: int, batch_size: int = 5,
             temporal_overlap: int = 0, seed: int = 42, dtype=torch.float32, uniform_batch_size: bool = False,
             taps: dict | None = None, color_correction="none", input_noise_scale: float = 0.0,
             latent_noise_scale: float = 0.0, rope_mode: str = "3b") -> torch.Tensor:
    """frames [N,H,W,3] float in [0,1] -> [N,true_h,true_w,3] in [0,1] (dtype bf16 storage between phases,
    model arithmetic in `dtype`).  rope_mode "3b" | "7b" | "none" selects the DiT's rotary embedding."""
    N, H, W, _ = frames.shape
    rh, rw, th, tw, ph, pw = target_dims(H, W, resolution)
    ranges, overlap = batch_ranges(N, batch_size, temporal_overlap)
    src = frames.to(torch.bfloat16)
    final = torch.empty(N, th, tw, 3, dtype=torch.bfloat16)
    write = 0
    written = []
    g_vae = torch.Generator(device="cpu").manual_seed(int(seed) + 1000000)     # generation_phases.py:325-326
    for bi, (s, e) in enumerate(ranges):
        vid = src[s:e]
        ori = e - s
        if uniform_batch_size and ori < batch_size:
            vid = pad_video_temporal(vid, batch_size - ori)
        vid = pad_video_temporal(vid)                                   # 4n+1
        x = transform(vid.permute(0, 3, 1, 2), resolution)             # [3,T,Hp,Wp] bf16
        x_in = x
        if input_noise_scale > 0:                                      # the colour reference below stays noise-free
            x_in = apply_input_noise(x, torch.randn(x.shape, generator=g_vae, dtype=torch.bfloat16), input_noise_scale)
        latent = _vae.encode(vae_p, x_in.float() if dtype == torch.float32 else x_in, vae_cfg, dtype).to(torch.bfloat16)
        g = torch.Generator(device="cpu").manual_seed(int(seed))        # set_seed(seed) per batch (:659)
        noise = torch.randn(tuple(latent.shape), generator=g, dtype=torch.bfloat16)
        second = torch.randn(tuple(latent.shape), generator=g, dtype=torch.bfloat16)   # always drawn (:679)
        cond_latent = latent_blur(latent, noise, second, latent_noise_scale)
        x0 = _dit.one_step_sr(dit_p, dit_cfg, cond_latent.to(dtype), noise.to(dtype), text_pos.to(dtype), dtype, rope_mode)
        x0 = x0.to(torch.bfloat16)
        dec = _vae.decode(vae_p, x0.to(dtype), vae_cfg, dtype).to(torch.bfloat16)   # [3,T,Hp,Wp]
        if taps is not None and bi == 0:
            taps.update(x=x, latent=latent, x0=x0, dec=dec)
        sample = dec.permute(1, 0, 2, 3)[:ori, :, :th, :tw].permute(0, 2, 3, 1)     # [T,H,W,3]
        skipped = 0
        if bi > 0 and overlap and overlap < sample.shape[0] and write >= overlap:
            final[write - overlap:write] = blend_overlapping_frames(final[write - overlap:write], sample[:overlap], overlap)
            sample = sample[overlap:]
            skipped = overlap
        final[write:write + sample.shape[0]] = sample
        # transformed input of the frames this batch wrote (generation_phases.py:1237-1262)
        xin = x.permute(1, 0, 2, 3)[:ori, :, :th, :tw].permute(0, 2, 3, 1)[skipped:skipped + sample.shape[0]]
        written.append((write, write + sample.shape[0], xin.contiguous()))
        write += sample.shape[0]
    final = final[:write]

    def phase4(method):
        out = final.clone()
        if method != "none":                                          # generation_phases.py:1290-1304
            from . import color as _color
            for ws, we, xin in written:
                out[ws:we] = _color.color_correct(out[ws:we].contiguous(), xin, method)
        return out.clamp(-1, 1).mul(0.5).add(0.5)                     # generation_phases.py:1337

    if isinstance(color_correction, (tuple, list)):   # several phase-4 flavours of ONE model pass (tests: the models run once)
        return {m: phase4(m) for m in color_correction}
    return phase4(color_correction)
