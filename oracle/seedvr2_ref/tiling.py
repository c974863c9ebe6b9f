"""CPU restatement of the reference's spatial VAE tiling (TEST INFRASTRUCTURE ONLY).

Follows `VideoAutoencoderKL.tiled_encode` / `tiled_decode`
[ref: src/models/video_vae_v3/modules/attn_video_vae.py:1294-1458, 1460-1622] around an arbitrary tile codec
(`encode_fn` / `decode_fn` stand for `slicing_encode` / `slicing_decode`).  Tensors are [C, T, H, W] (the reference's
[B, C, T, H, W] with B = 1 dropped) in the VAE dtype: every product, sum and the final division round to that dtype,
exactly as the in-place torch ops of the reference do.

Pinned bit for bit (fp32 and bf16) by oracle/pin_against_reference.py (G13) against the two reference methods themselves,
run with the same synthetic tile codec (tests/golden/tiling_pins.safetensors).
"""

from __future__ import annotations

from typing import Callable, List, Tuple

import torch


def tile_plan(total_h: int, total_w: int, tile: Tuple[int, int], overlap: Tuple[int, int]) -> List[Tuple[int, int, int, int]]:
    """Latent-space tiles (y, y_end, x, x_end) in the reference's visiting order; `tile` and `overlap` are already in
    latent units (tile >= 1, 0 <= overlap <= tile-1).  Tiles that lie fully inside the overlap of their predecessor are
    skipped (:1361-1364, :1524-1527)."""
    th, tw = tile
    oh, ow = overlap
    sh, sw = max(1, th - oh), max(1, tw - ow)
    out = []
    for y in range(0, total_h, sh):
        y_end = min(y + th, total_h)
        for x in range(0, total_w, sw):
            x_end = min(x + tw, total_w)
            if (y > 0 and (y_end - y) <= oh) or (x > 0 and (x_end - x) <= ow):
                continue
            out.append((y, y_end, x, x_end))
    return out


def ramp(steps: int, dtype) -> torch.Tensor:
    """0.5 - 0.5*cos(linspace(0,1,steps)*pi) evaluated op by op in `dtype` (:1346-1351, :1510-1515)."""
    t = torch.linspace(0, 1, steps=steps, dtype=dtype)
    return 0.5 - 0.5 * torch.cos(t * torch.pi)


def edge_weights(n: int, ov: int, r: torch.Tensor, fade_head: bool, fade_tail: bool, dtype) -> torch.Tensor:
    """Ones with the ramp on interior edges only (:1417-1431, :1579-1593)."""
    w = torch.ones((n,), dtype=dtype)
    if ov > 0:
        if fade_head:
            w[:ov] = r[:ov]
        if fade_tail:
            w[-ov:] = 1 - r[:ov]
    return w


def tiled_encode(x: torch.Tensor, encode_fn: Callable[[torch.Tensor], torch.Tensor], tile_size=(512, 512),
                 tile_overlap=(64, 64), scale: int = 8) -> torch.Tensor:
    """x [C,T,H,W] -> encoded [C',T',ceil(H/scale),ceil(W/scale)]; sizes in input pixels (:1294-1458)."""
    _, _, H, W = x.shape
    tile_h, tile_w = tile_size
    if H <= tile_h and W <= tile_w:
        return encode_fn(x)
    lt_h, lt_w = max(1, tile_h // scale), max(1, tile_w // scale)
    lo_h = max(0, min(tile_overlap[0] // scale, lt_h - 1))
    lo_w = max(0, min(tile_overlap[1] // scale, lt_w - 1))
    Hl, Wl = (H + scale - 1) // scale, (W + scale - 1) // scale
    result = count = None
    rh = rw = None
    for (y, y_end, xx, x_end) in tile_plan(Hl, Wl, (lt_h, lt_w), (lo_h, lo_w)):
        tile = x[:, :, y * scale:min(y_end * scale, H), xx * scale:min(x_end * scale, W)]
        enc = encode_fn(tile)
        if result is None:
            result = torch.zeros((enc.shape[0], enc.shape[1], Hl, Wl), dtype=enc.dtype)
            count = torch.zeros((1, 1, Hl, Wl), dtype=enc.dtype)
            rh = ramp(lo_h, enc.dtype) if lo_h > 0 else None
            rw = ramp(lo_w, enc.dtype) if lo_w > 0 else None
        eh = min(y_end - y, enc.shape[2], Hl - y)
        ew = min(x_end - xx, enc.shape[3], Wl - xx)
        enc = enc[:, : result.shape[1], :eh, :ew].clone()
        ov_h, ov_w = max(0, min(lo_h, eh - 1)), max(0, min(lo_w, ew - 1))
        wh = edge_weights(eh, ov_h, rh, y > 0, y_end < Hl, enc.dtype).view(1, 1, eh, 1)
        ww = edge_weights(ew, ov_w, rw, xx > 0, x_end < Wl, enc.dtype).view(1, 1, 1, ew)
        enc.mul_(wh).mul_(ww)
        result[:, : enc.shape[1], y:y + eh, xx:xx + ew] += enc
        count[:, :, y:y + eh, xx:xx + ew].addcmul_(wh, ww)
    result.div_(count.clamp(min=1e-6))
    return result


def tiled_decode(z: torch.Tensor, decode_fn: Callable[[torch.Tensor], torch.Tensor], tile_size=(512, 512),
                 tile_overlap=(64, 64), scale: int = 8) -> torch.Tensor:
    """z [C,T,h,w] -> decoded [C',T',h*scale,w*scale]; tile sizes in OUTPUT pixels, blend ramps `tile_overlap` output
    pixels long although tiles advance by (tile - overlap)//scale latent pixels (:1460-1622)."""
    _, _, H, W = z.shape
    tile_h, tile_w = tile_size
    ov_h_px, ov_w_px = tile_overlap
    lt_h, lt_w = max(1, tile_h // scale), max(1, tile_w // scale)
    if H <= lt_h and W <= lt_w:
        return decode_fn(z)
    lo_h = max(0, min(ov_h_px // scale, lt_h - 1))
    lo_w = max(0, min(ov_w_px // scale, lt_w - 1))
    result = count = None
    rh = rw = None
    for (y, y_end, xx, x_end) in tile_plan(H, W, (lt_h, lt_w), (lo_h, lo_w)):
        dec = decode_fn(z[:, :, y:y_end, xx:x_end]).clone()
        if result is None:
            result = torch.zeros((dec.shape[0], dec.shape[1], H * scale, W * scale), dtype=dec.dtype)
            count = torch.zeros((1, 1, H * scale, W * scale), dtype=dec.dtype)
            rh = ramp(ov_h_px, z.dtype) if ov_h_px > 0 else None
            rw = ramp(ov_w_px, z.dtype) if ov_w_px > 0 else None
        y0, y1, x0, x1 = y * scale, y_end * scale, xx * scale, x_end * scale
        h_out, w_out = y1 - y0, x1 - x0
        ov_h, ov_w = max(0, min(ov_h_px, h_out - 1)), max(0, min(ov_w_px, w_out - 1))
        wh = edge_weights(h_out, ov_h, rh, y > 0, y_end < H, dec.dtype).view(1, 1, h_out, 1)
        ww = edge_weights(w_out, ov_w, rw, xx > 0, x_end < W, dec.dtype).view(1, 1, 1, w_out)
        dec.mul_(wh).mul_(ww)
        result[:, : dec.shape[1], y0:y1, x0:x1] += dec
        count[:, :, y0:y1, x0:x1].addcmul_(wh, ww)
    result.div_(count.clamp(min=1e-6))
    return result


# --------------------------------------------------------------------------------------------------------------------
# Synthetic tile codecs used to pin the two functions above against the reference methods (pin_against_reference.py G13)
# and to re-check the committed vectors in tests/: cheap stand-ins for slicing_encode / slicing_decode whose output near
# a tile border depends on where the tile was cut, so that a wrong blend cannot go unnoticed.
def synthetic_enc_codec(x):      # [.., C, F, H, W] -> 4 channels, every 4th frame, 8x8 mean of a nonlinear channel mix
    lead = x.shape[:-4]
    c, f, h, w = x.shape[-4:]
    y = x.reshape(-1, c, f, h, w)[:, :, ::4].float()
    mix = torch.stack([y.sum(1), y[:, 0] - y[:, 1], torch.tanh(y[:, 2] * 2), y[:, 0] * y[:, 1]], dim=1)
    # the 3x3 smoothing sees the tile border (zero padding), so overlapping tiles disagree and the blend matters
    sm = torch.nn.functional.avg_pool2d(mix.flatten(0, 2)[:, None], 3, 1, 1)
    out = torch.nn.functional.avg_pool2d(sm, 8, 8, ceil_mode=True)
    return out.reshape(*lead, 4, y.shape[2], out.shape[-2], out.shape[-1]).to(x.dtype)

def synthetic_dec_codec(z):      # [.., C, F, h, w] -> 3 channels, 8x nearest + 5x5 smoothing
    lead = z.shape[:-4]
    c, f, h, w = z.shape[-4:]
    y = z.reshape(-1, c, f, h, w).float()
    mix = torch.stack([y[:, 0] + y[:, 1], torch.sin(y[:, 2] * 3), y[:, 3] - y[:, 0]], dim=1)
    up = torch.nn.functional.interpolate(mix.flatten(0, 2)[:, None], scale_factor=8, mode="nearest")
    sm = torch.nn.functional.avg_pool2d(up, 5, 1, 2)
    return sm.reshape(*lead, 3, f, h * 8, w * 8).to(z.dtype)

