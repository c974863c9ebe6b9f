"""Per-frame-chunk pipeline: transform -> VAE encode -> one-step DiT -> VAE decode -> blend -> [0,1].

Host-side restatement of the reference's four generation phases
[ref: src/core/generation_phases.py:171-521 (encode_all_batches), :538-801 (upscale_all_batches),
 :802-1047 (decode_all_batches), :1049-1362 (postprocess_all_batches, color_correction='none')]
and their helpers [ref: src/core/generation_utils.py:47-84 (transforms), :284-312 (blend),
 :555-590 (batch params), :593-652 (temporal padding)] sized for 288 GB of HBM: every batch stays on the
device from LR frames to the blended output (no CPU offload, no BlockSwap, no tiling).
"""

from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import hip


# ----------------------------------------------------------------------------- geometry
def target_dims(h: int, w: int, resolution: int, max_resolution: int = 0) -> Tuple[int, int, int, int, int, int]:
    """(resized_h, resized_w, true_h, true_w, padded_h, padded_w).
    Shortest edge -> resolution with torchvision resize(int) semantics (long = int(res*long/short))
    [ref: data/image/transforms/side_resize.py:36-75]; true dims rounded down to even
    [ref: core/generation_utils.py:134-136]; processing dims padded up to x16 [ref: divisible_crop.py:43-80]."""
    short, long_ = (w, h) if w <= h else (h, w)
    new_short, new_long = resolution, int(resolution * long_ / short)
    rh, rw = (new_long, new_short) if w <= h else (new_short, new_long)
    if max_resolution > 0 and max(rh, rw) > max_resolution:
        s = max_resolution / max(rh, rw)
        rh, rw = round(rh * s), round(rw * s)
    th, tw = (rh // 2) * 2, (rw // 2) * 2
    ph, pw = (rh + 15) // 16 * 16, (rw + 15) // 16 * 16
    return rh, rw, th, tw, ph, pw


def uncapped_dims(h: int, w: int, resolution: int) -> Tuple[int, int]:
    """(rh, rw) of the first resize (shortest edge -> resolution), before any `max_resolution` cap."""
    rh, rw, *_ = target_dims(h, w, resolution, 0)
    return rh, rw


def _cubic(x: np.ndarray, a: float = -0.5) -> np.ndarray:
    x = np.abs(x).astype(np.float32)
    w1 = ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0
    w2 = (((x - 5.0) * x + 8.0) * x - 4.0) * a
    return np.where(x < 1.0, w1, np.where(x < 2.0, w2, 0.0)).astype(np.float32)


def aa_bicubic_taps(in_size: int, out_size: int):
    """Separable tap table of ATen's antialiased bicubic (_upsample_bicubic2d_aa, align_corners=False):
    support = 2*max(scale,1), weights = cubic(a=-0.5) of the (scaled) distance to the pixel centre,
    normalised to sum 1.  (torchvision's resize(..., BICUBIC, antialias=True) lowers to this op,
    [ref: data/image/transforms/side_resize.py:61].)  Returns (weights [out,K] f32, xmin [out], count [out])."""
    scale = in_size / out_size
    support = 2.0 * scale if scale >= 1.0 else 2.0
    invscale = 1.0 / scale if scale >= 1.0 else 1.0
    K = int(math.ceil(support)) * 2 + 1
    wts = np.zeros((out_size, K), dtype=np.float32)
    mins = np.zeros(out_size, dtype=np.int32)
    cnts = np.zeros(out_size, dtype=np.int32)
    for i in range(out_size):
        center = np.float32(scale) * np.float32(i + 0.5)
        xmin = max(int(center - np.float32(support) + np.float32(0.5)), 0)
        xsize = min(int(center + np.float32(support) + np.float32(0.5)), in_size) - xmin
        j = np.arange(xsize, dtype=np.float32)
        w = _cubic((j + np.float32(xmin) - center + np.float32(0.5)) * np.float32(invscale))
        tot = w.sum(dtype=np.float32)
        if tot != 0:
            w = w / tot
        wts[i, :xsize] = w
        mins[i], cnts[i] = xmin, xsize
    return wts, mins, cnts


# ----------------------------------------------------------------------------- temporal batching
@dataclass
class Batch:
    start: int       # first source frame
    end: int         # one past the last source frame
    ori_len: int     # frames that are real (kept after decode)
    index: List[int]  # source-frame index of every frame fed to the VAE (4n+1 / uniform padding applied)


def _reversed_tail_indices(t: int, count: int) -> List[int]:
    """Indices appended by pad_video_temporal(prepend=False) [ref: core/generation_utils.py:593-652]."""
    if count <= 0:
        return []
    if count >= t:
        rep = count - t + 1
        return list(range(t - 1, 0, -1)) + [t - 1] * rep if t > 1 else [t - 1] * rep
    return list(range(t - 2, t - 2 - count, -1))


def plan_batches(total: int, batch_size: int, overlap: int, uniform: bool = False) -> Tuple[List[Batch], int]:
    """Batch i = frames [i*step, i*step + batch_size), stop when the remainder <= overlap
    [ref: core/generation_phases.py:267-282, 342-400]; each batch padded to 4n+1 (and to batch_size when
    uniform) by appending reversed trailing frames [ref: :109-124].  Returns (batches, effective overlap)."""
    step = batch_size - overlap if overlap > 0 else batch_size
    if step <= 0:
        step, overlap = batch_size, 0
    out: List[Batch] = []
    for s in range(0, total, step):
        e = min(s + batch_size, total)
        if s > 0 and e - s <= overlap:
            break
        n = e - s
        idx = list(range(n))
        if uniform and n < batch_size:
            idx += [idx[j] for j in _reversed_tail_indices(len(idx), batch_size - n)]
        t = len(idx)
        if t % 4 != 1:
            cnt = ((t - 1) // 4 + 1) * 4 + 1 - t
            if cnt >= t:  # repeat-last + reversed order of pad_video_temporal's "exceeds length" branch
                rep = cnt - t + 1
                idx = idx + ([idx[j] for j in range(t - 1, 0, -1)] if t > 1 else []) + [idx[-1]] * rep
            else:
                idx = idx + [idx[j] for j in range(t - 2, t - 2 - cnt, -1)]
        out.append(Batch(s, e, n, [s + j for j in idx]))
    return out, overlap


def blend_weights(overlap: int) -> torch.Tensor:
    """w_prev per overlap frame, computed in bf16 like the reference (the blend runs on bf16 tensors)
    [ref: core/generation_utils.py:284-312]."""
    dt = torch.bfloat16
    if overlap >= 3:
        t = torch.linspace(0.0, 1.0, steps=overlap, dtype=dt)
        u = ((t - 1.0 / 3.0) / (2.0 / 3.0 - 1.0 / 3.0)).clamp(0.0, 1.0)
        w = 0.5 + 0.5 * torch.cos(torch.pi * u)
    else:
        w = torch.linspace(1.0, 0.0, steps=overlap, dtype=dt)
    return w.float()


# ----------------------------------------------------------------------------- the upscaler
class SeedVR2Pipeline:
    """Owns a VAE + DiT pair on one device and runs batches of LR frames through them."""

    def __init__(self, vae, dit, text_pos: torch.Tensor, device):
        self.vae, self.dit = vae, dit
        self.device = torch.device(device)
        self.text_pos = text_pos.to(self.device).to(torch.bfloat16).contiguous()
        self._taps = {}
        self.set_tiling()

    def set_tiling(self, encode_tiled: bool = False, encode_tile_size: int = 1024, encode_tile_overlap: int = 128,
                   decode_tiled: bool = False, decode_tile_size: int = 1024, decode_tile_overlap: int = 128) -> None:
        """Spatial VAE tiling flags of the reference's VAE loader / CLI (square tiles, sizes in pixels)
        [ref: src/core/infer.py:38-51, 160-173, 237-244; inference_cli.py:842-847]."""
        self.encode_tiling = dict(tiled=bool(encode_tiled), tile_size=(encode_tile_size,) * 2,
                                  tile_overlap=(encode_tile_overlap,) * 2)
        self.decode_tiling = dict(tiled=bool(decode_tiled), tile_size=(decode_tile_size,) * 2,
                                  tile_overlap=(decode_tile_overlap,) * 2)

    def _resize_taps(self, hin, win, rh, rw):
        key = (hin, win, rh, rw)
        if key not in self._taps:
            wy, ymin, ycnt = aa_bicubic_taps(hin, rh)
            wx, xmin, xcnt = aa_bicubic_taps(win, rw)
            up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
            bounds = (int((ymin + ycnt).max()), int((xmin + xcnt).max()), int(ycnt.max()), int(xcnt.max()))
            self._taps[key] = hip.ResizeTaps((up(a) for a in (wy, ymin, ycnt, wx, xmin, xcnt)), bounds)
        return self._taps[key]

    # one batch, device resident ------------------------------------------------------------
    @hip.on_device
    def transform(self, frames: torch.Tensor, resolution: int, max_resolution: int = 0) -> torch.Tensor:
        """frames bf16 [T,Hin,Win,3] in [0,1] -> bf16 [T,Hp,Wp,3] in [-1,1] (padding = -1)."""
        T, hin, win, _ = frames.shape
        rh, rw, _, _, ph, pw = target_dims(hin, win, resolution, max_resolution)
        rh0, rw0 = uncapped_dims(hin, win, resolution)
        if (rh0, rw0) != (rh, rw):
            # the cap triggered: the reference resizes to the short-edge size first and then resizes THAT (bf16) image to
            # the capped size [ref: data/image/transforms/side_resize.py:61-73] -- two cascaded antialiased passes
            frames = hip.resize_bicubic_aa(frames, self._resize_taps(hin, win, rh0, rw0))
            hin, win = rh0, rw0
        return hip.resize_pad_normalize(frames, self._resize_taps(hin, win, rh, rw), ph, pw)

    @staticmethod
    def make_noise(latent_shape, seed: int) -> torch.Tensor:
        """Noise is drawn by the CPU generator on a bf16 CPU tensor, exactly like the reference's CPU path:
        torch.manual_seed(seed); randn_like(latent)  [ref: core/generation_phases.py:659,676-679;
        common/seed.py:23-29].  (A second randn_like is drawn there for the unused aug noise.)"""
        g = torch.Generator(device="cpu").manual_seed(int(seed))
        return torch.randn(tuple(latent_shape), generator=g, dtype=torch.bfloat16, device="cpu")

    @staticmethod
    def blur_coefficients(latent_shape, scale: float, T: float = 1000.0) -> Tuple[float, float]:
        """(A, B) of x_t = A*latent + B*aug_noise for latent_noise_scale [ref: generation_phases.py:684-693; infer.py:260-290;
        schedules/lerp.py]: t = bf16(1000)*scale, resolution-dependent shift of `timestep_transform` fed -- as the reference
        does -- with the channels-last latent's shape[1:] = (h, w, 16) in the (frames, height, width) slots, fp32 from the
        shift on.  A handful of scalar torch ops on the host, with the reference's dtype promotions."""
        t = torch.tensor([1000.0], dtype=torch.bfloat16) * scale
        shape = torch.tensor(tuple(latent_shape[1:]))[None]
        frames, heights, widths = (shape[:, 0] - 1) * 4 + 1, shape[:, 1] * 8, shape[:, 2] * 8

        def lin(x1, y1, x2, y2):
            m = (y2 - y1) / (x2 - x1)
            b = y1 - m * x1
            return lambda x: m * x + b

        img, vid = lin(256 * 256, 1.0, 1024 * 1024, 3.2), lin(256 * 256 * 37, 1.0, 1280 * 720 * 145, 5.0)
        shift = torch.where(frames > 1, vid(heights * widths * frames), img(heights * widths))
        t = t / T
        t = shift * t / (1 + (shift - 1) * t)
        t = t * T
        return float(1 - (t / T)), float(t / T)

    @hip.on_device
    def upscale_batch(self, lr_frames: torch.Tensor, resolution: int, seed: int, max_resolution: int = 0,
                      noise: Optional[torch.Tensor] = None, input_noise_scale: float = 0.0,
                      latent_noise_scale: float = 0.0, vae_generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """lr_frames bf16 [T(4n+1),Hin,Win,3] on the device -> decoded bf16 [T,Hp,Wp,3] in ~[-1,1].
        input_noise_scale / latent_noise_scale: the reference's optional noise injections; the random numbers come from the
        CPU generators the reference's CPU path uses (seed + 1 000 000 once per clip for the input noise, `seed` per batch
        for base + aug noise), so results are reproducible against the oracle (a 4K batch draws 0.5 G numbers on the
        host: seconds -- these knobs are off by default)."""
        x = self.transform(lr_frames, resolution, max_resolution)
        if input_noise_scale > 0:
            g = vae_generator or torch.Generator(device="cpu").manual_seed(int(seed) + 1000000)
            T_, Hp, Wp, _ = x.shape
            nz = torch.randn((3, T_, Hp, Wp), generator=g, dtype=torch.bfloat16)       # randn_like([C,T,H,W]) of the reference
            nz = nz.permute(1, 2, 3, 0).contiguous().to(self.device)
            bf_ = input_noise_scale * 0.5
            x = hip.noise_mix(x, nz, 1 - bf_, bf_, 0)
            del nz
        # The noise has the latent's shape, which follows from x: draw it (CPU generator, like the reference's CPU path) and
        # start its upload BEFORE the encoder is enqueued.  Issued after it, the copy reached the GPU 0.76 ms after the
        # encoder's last kernel on a 512 x 512 image (the host draw of 65 k bf16 numbers alone takes 0.9 ms).  Page-locked
        # source + non_blocking: the host never waits for the stream.
        vcfg = getattr(self.vae, "cfg", None)
        latent = None
        if vcfg is not None:
            sdown, tdown = 2 ** (len(vcfg.block_out_channels) - 1), 2 ** vcfg.temporal_scale_num
            lshape = ((x.shape[0] - 1) // tdown + 1, x.shape[1] // sdown, x.shape[2] // sdown, vcfg.latent_channels)
        else:           # a codec that does not describe its strides: its latent tells the shape (noise drawn after the encode)
            latent = self.vae.encode(x, **self.encode_tiling)
            lshape = tuple(latent.shape)
        second = None
        if noise is None:
            g = torch.Generator(device="cpu").manual_seed(int(seed))
            noise = torch.randn(lshape, generator=g, dtype=torch.bfloat16, device="cpu", pin_memory=True)
            if latent_noise_scale > 0:
                second = torch.randn(lshape, generator=g, dtype=torch.bfloat16, device="cpu", pin_memory=True)
        elif latent_noise_scale > 0:
            raise hip.HipError("latent_noise_scale needs the pipeline's own noise (base and aug noise share a generator)")
        noise = noise.to(self.device, non_blocking=True).contiguous()
        if second is not None:
            second = second.to(self.device, non_blocking=True).contiguous()
        if latent is None:
            latent = self.vae.encode(x, **self.encode_tiling)
        del x
        if tuple(latent.shape) != tuple(noise.shape):
            raise hip.HipError(f"noise shape {tuple(noise.shape)} does not match the latent {tuple(latent.shape)}")
        cond = latent
        if latent_noise_scale > 0:
            aug = hip.noise_mix(noise, second, 0.1, 0.05, 1)
            a_, b_ = self.blur_coefficients(latent.shape, latent_noise_scale)
            cond = hip.noise_mix(latent, aug, a_, b_, 2)
            del aug
        x0 = self.dit.one_step(noise, cond, self.text_pos)
        del latent, noise, cond
        return self.vae.decode(x0, **self.decode_tiling)

    # whole clip ----------------------------------------------------------------------------
    @hip.on_device
    @torch.no_grad()
    def generate(self, frames: torch.Tensor, resolution: int, batch_size: int = 5, temporal_overlap: int = 0,
                 seed: int = 42, max_resolution: int = 0, uniform_batch_size: bool = False,
                 progress=None, color_correction: str = "none", input_noise_scale: float = 0.0,
                 latent_noise_scale: float = 0.0) -> torch.Tensor:
        """frames [N,Hin,Win,C] float in [0,1] (any device), C = 3 or 4 -> bf16 [N,true_h,true_w,C] in [0,1] on the device.
        color_correction: 'none' or one of hip.COLOR_METHODS (phase 4, against the transformed input of each batch
        [ref: src/core/generation_phases.py:1237-1304]).  RGBA: the colour channels take the normal path, the alpha plane
        is upscaled with the blended RGB as edge guide before colour correction [ref: generation_phases.py:403-469,
        1132-1190; alpha_upscaling.py] and passes through un-normalised [ref: :1325-1333]."""
        if color_correction != "none" and color_correction not in hip.COLOR_METHODS:
            raise hip.HipError(f"color_correction '{color_correction}' is not implemented "
                               f"(available: none, {', '.join(hip.COLOR_METHODS)})")
        N, hin, win, C = frames.shape
        if C not in (3, 4):
            raise hip.HipError(f"frames must be [N,H,W,3|4], got {tuple(frames.shape)}")
        rh, rw, th, tw, ph, pw = target_dims(hin, win, resolution, max_resolution)
        batches, overlap = plan_batches(N, batch_size, temporal_overlap, uniform_batch_size)
        src = frames.to(self.device).to(torch.bfloat16)
        alpha_src = None
        if C == 4:
            alpha_src = src[..., 3].contiguous()
            src = src[..., :3].contiguous()
        # every source frame is written exactly once  [ref: generation_phases.py:868 (pre-allocated final_video)]
        final = torch.empty(N, th, tw, 3, dtype=torch.bfloat16, device=self.device)
        w_prev = blend_weights(overlap).to(self.device) if overlap else None
        write = 0
        g_vae = torch.Generator(device="cpu").manual_seed(int(seed) + 1000000)   # [ref: generation_phases.py:325-326]
        written = []   # (first frame, end frame, batch, frames of the batch that went into the previous slice)
        for bi, b in enumerate(batches):
            idx = torch.tensor(b.index, device=self.device)
            dec = self.upscale_batch(src.index_select(0, idx).contiguous(), resolution, seed, max_resolution,
                                     input_noise_scale=input_noise_scale, latent_noise_scale=latent_noise_scale,
                                     vae_generator=g_vae)
            sample = hip.crop_to_unit(dec[: b.ori_len].contiguous(), th, tw, to_unit=False)  # trim temporal+spatial pad
            del dec
            skipped = 0
            if bi > 0 and overlap and overlap < sample.shape[0] and write >= overlap:
                tail = final[write - overlap:write]
                hip.blend_overlap(tail.clone(), sample[:overlap].contiguous(), w_prev, tail)
                sample = sample[overlap:]
                skipped = overlap
            final[write:write + sample.shape[0]] = sample
            written.append((write, write + sample.shape[0], bi, skipped))
            write += sample.shape[0]
            if progress:
                progress(bi + 1, len(batches))
        final = final[:write]
        alpha_out = None
        if alpha_src is not None:
            alpha_out = torch.empty(write, th, tw, dtype=torch.bfloat16, device=self.device)
            taps = self._resize_taps(hin, win, th, tw)
            for ws, we, bi, skipped in written:
                idx = batches[bi].index
                if len(idx) != we - ws:
                    # the reference pairs the alpha of the PADDED batch with the frames the batch contributed and fails in
                    # torch on the mismatch [ref: generation_phases.py:444,1157-1168; alpha_upscaling.py:82-88]
                    raise ValueError(f"RGBA: batch {bi} holds {len(idx)} frames (4n+1 / uniform padding, overlap) but "
                                     f"contributes {we - ws}; alpha upscaling needs equal counts (use batch sizes that "
                                     "divide the clip into 4n+1 batches without temporal overlap, or single images)")
                a_in = alpha_src.index_select(0, torch.tensor(idx, device=self.device)).contiguous()
                alpha_out[ws:we] = hip.alpha_upscale(a_in, final[ws:we].contiguous(), taps).to(torch.bfloat16)
        if color_correction != "none":
            # phase 4: every batch's slice of the blended video is matched to the transformed input of the same frames
            # (re-created on demand like the reference's _reconstruct_and_transform_batch, generation_phases.py:1237-1262)
            for ws, we, bi, skipped in written:
                b = batches[bi]
                idx = torch.tensor(b.index, device=self.device)
                x = self.transform(src.index_select(0, idx).contiguous(), resolution, max_resolution)
                x = hip.crop_to_unit(x[: b.ori_len].contiguous(), th, tw, to_unit=False)
                x = x[skipped:skipped + (we - ws)].contiguous()
                final[ws:we] = hip.color_correct(final[ws:we].contiguous(), x, color_correction)
        # phase 4 tail: clamp(-1,1)*0.5+0.5  [ref: generation_phases.py:1337]; alpha is attached as it is [ref: :1325-1333]
        rgb = hip.crop_to_unit(final, th, tw, to_unit=True)
        return rgb if alpha_out is None else torch.cat((rgb, alpha_out.unsqueeze(-1)), dim=-1)
