"""ctypes binding of libseedvr2_hip.so (the C ABI declared in include/seedvr2_hip.h).

PyTorch-ROCm is used only as the owner of device memory and streams: every wrapper takes
torch tensors, checks shapes/dtypes on the host, and passes raw device pointers plus the
current HIP stream to the library.  There is NO fallback: if the shared library is missing
or the device is not gfx950 the import / first call raises.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION = 2     # SVR2_ABI_VERSION of include/seedvr2_hip.h this binding was written against (struct layouts below)
LIB_PATH = os.environ.get("SVR2_LIB") or os.path.join(_HERE, "libseedvr2_hip.so")  # SVR2_LIB: A/B builds

EPI_BIAS, EPI_GATE_RES, EPI_SWIGLU, EPI_PIXSHUF3D, EPI_GELU = 0, 1, 2, 3, 4


class HipError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [
        ("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int64),
        ("A", C.c_void_p), ("lda", C.c_int64), ("a_rows", C.c_void_p),
        ("W", C.c_void_p), ("C", C.c_void_p), ("ldc", C.c_int64),
        ("epilogue", C.c_int32),
        ("bias", C.c_void_p), ("gate", C.c_void_p), ("resid", C.c_void_p), ("ldr", C.c_int64),
        ("ps_frames", C.c_int32), ("ps_h", C.c_int32), ("ps_w", C.c_int32), ("ps_c", C.c_int32),
        ("ps_tz", C.c_int32), ("ps_drop_head", C.c_int32),
        ("zero_page", C.c_void_p),
    ]


class Conv3dDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("carry", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p),
        ("resid", C.c_void_p), ("y", C.c_void_p),
        ("T", C.c_int32), ("Hi", C.c_int32), ("Wi", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32),
        ("kt", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32), ("st", C.c_int32), ("sh", C.c_int32),
        ("sw", C.c_int32),
        ("npad", C.c_int32), ("pad_h0", C.c_int32), ("pad_w0", C.c_int32),
        ("To", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("zero_page", C.c_void_p),
        ("w_layout", C.c_int32),
        ("gn_partial", C.c_void_p),
        ("split_ws", C.c_void_p),
        ("split_ws_bytes", C.c_int64),
        ("carry_frames", C.c_int32),
    ]


class QkvPrepDesc(C.Structure):
    _fields_ = [
        ("qkv_vid", C.c_void_p), ("qkv_txt", C.c_void_p), ("src_row", C.c_void_p), ("pos", C.c_void_p),
        ("qn_vid", C.c_void_p), ("kn_vid", C.c_void_p), ("qn_txt", C.c_void_p), ("kn_txt", C.c_void_p),
        ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p),
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p),
        ("S", C.c_int64),
        ("H", C.c_int32), ("D", C.c_int32), ("nfreq", C.c_int32), ("max_pos", C.c_int32),
        ("rope_mode", C.c_int32), ("eps", C.c_float),
    ]


_lib = None
_zero_pages = {}
_INT64_RET = ("svr2_groupnorm_ws_floats", "svr2_adain_ws_bytes", "svr2_hist_match_ws_bytes", "svr2_hue_sat_match_ws_bytes",
              "svr2_jpeg_ws_bytes", "svr2_conv3d_split_ws_bytes")

# every symbol include/seedvr2_hip.h declares (tests check the .so exports all of them)
EXPORTS = [
    "svr2_abi_version", "svr2_last_error", "svr2_device_check", "svr2_set_option", "svr2_gemm", "svr2_conv3d", "svr2_im2col3d",
    "svr2_rmsnorm_mod", "svr2_qkv_prep", "svr2_window_attn", "svr2_window_attn_scatter", "svr2_txt_mean", "svr2_patch_in",
    "svr2_patch_out_x0", "svr2_groupnorm_ws_floats", "svr2_groupnorm_stats", "svr2_groupnorm_apply",
    "svr2_groupnorm_finalize", "svr2_conv3d_gn_tiles", "svr2_conv3d_split_ws_bytes",
    "svr2_attn_d512", "svr2_resize_pad_normalize", "svr2_resize_bicubic_aa", "svr2_blend_overlap", "svr2_crop_to_unit",
    "svr2_wavelet_level", "svr2_wavelet_merge", "svr2_adain_ws_bytes", "svr2_adain", "svr2_rgb_to_lab",
    "svr2_hist_match_ws_bytes", "svr2_hist_match", "svr2_lab_to_rgb",
    "svr2_rgb_to_hsv", "svr2_hue_sat_match_ws_bytes", "svr2_hue_sat_match", "svr2_hsv_to_rgb", "svr2_sat_blend",
    "svr2_u8_to_frames", "svr2_frames_to_u8", "svr2_frames_to_yuv", "svr2_yuv_to_frames",
    "svr2_tile_accumulate", "svr2_tile_normalize",
    "svr2_alpha_stats", "svr2_alpha_guide", "svr2_alpha_sobel", "svr2_alpha_edges", "svr2_resize_plane",
    "svr2_guided_ab", "svr2_guided_out", "svr2_alpha_refine",
    "svr2_jpeg_ws_bytes", "svr2_jpeg_encode", "svr2_noise_mix",
]


def lib() -> C.CDLL:
    """Load the shared library (once).  Raises if it was not built: there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipError(
                f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
                "The SeedVR2 MI355X path has no CPU fallback.")
        if os.environ.get("SVR2_LIB"):    # A/B builds (and tools/build_timing_variant.py): never silently
            import sys
            print(f"[seedvr2_amd] loading the HIP library from SVR2_LIB={LIB_PATH}", file=sys.stderr, flush=True)
        _lib = C.CDLL(LIB_PATH)
        _lib.svr2_abi_version.restype = C.c_int
        got = _lib.svr2_abi_version()
        if got != ABI_VERSION:     # descriptor layouts are version-bound: a stale library would read them wrongly, silently
            _lib = None
            raise HipError(f"{LIB_PATH} has ABI version {got}, this binding needs {ABI_VERSION}: rebuild it "
                           "(`python __graft_entry__.py`)")
        _lib.svr2_last_error.restype = C.c_char_p
        for name in EXPORTS:
            if name != "svr2_last_error" and hasattr(_lib, name):
                getattr(_lib, name).restype = C.c_int64 if name in _INT64_RET else C.c_int
    return _lib


def _check(rc: int, what: str):
    if rc != 0:
        raise HipError(f"{what} failed (code {rc}): {lib().svr2_last_error().decode()}")


def _stream() -> C.c_void_p:
    """The calling thread's current stream on its CURRENT device; `_p` rejects operands that live elsewhere, so a
    kernel can never be enqueued on one GPU's stream with another GPU's pointers."""
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if t.is_cuda and t.device.index != torch.cuda.current_device():
        raise HipError(f"operand lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}: "
                       "run the call under torch.cuda.device(tensor.device) (the pipeline / model entry points do)")
    return C.c_void_p(t.data_ptr())


def set_option(name: str, value: Optional[int]) -> None:
    """A/B switch of the library (svr2_set_option): value None returns it to the built-in default."""
    _check(lib().svr2_set_option(name.encode(), C.c_int(0 if value is None else int(value)), C.c_int(value is None)),
           "svr2_set_option")
    _SPLIT_WS.clear()       # SVR2_CONV_SPLIT decides which convs are split


_SPLIT_WS: dict = {}        # conv geometry -> scratch bytes of the split-K path (asked from the library once per geometry)


def on_device(fn):
    """Method decorator: run under torch.cuda.device(self.device), so launches use that device's stream."""
    import functools

    @functools.wraps(fn)
    def inner(self, *a, **k):
        dev = torch.device(self.device)
        if dev.type != "cuda" or dev.index is None or dev.index == torch.cuda.current_device():
            return fn(self, *a, **k)
        with torch.cuda.device(dev):
            return fn(self, *a, **k)
    return inner


def zero_page(device) -> torch.Tensor:
    dev = torch.device(device)
    key = (dev.type, dev.index)
    if key not in _zero_pages:
        _zero_pages[key] = torch.zeros(1024, dtype=torch.uint8, device=dev)
    return _zero_pages[key]


def device_check(index: int = 0) -> int:
    rc = lib().svr2_device_check(C.c_int(index))
    if rc <= 0:
        _check(rc, "svr2_device_check")
    return rc


def _bf16c(t: torch.Tensor, name: str):
    if t.dtype != torch.bfloat16 or not t.is_cuda:
        raise HipError(f"{name}: expected a bf16 device tensor, got {t.dtype} on {t.device}")


def _f32c(t: Optional[torch.Tensor], name: str, n: Optional[int] = None):
    if t is None:
        return
    if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
        raise HipError(f"{name}: expected a contiguous fp32 device tensor")
    if n is not None and t.numel() < n:
        raise HipError(f"{name}: expected at least {n} elements, got {t.numel()}")


def pad_weight(w: torch.Tensor, kpad: Optional[int] = None) -> torch.Tensor:
    """[N, K] -> bf16 [round_up(N,128), kpad or round_up(K,64)] zero padded (the layout svr2_gemm wants)."""
    n, k = w.shape
    kp = kpad if kpad is not None else (k + 63) // 64 * 64
    out = torch.zeros((n + 127) // 128 * 128, kp, dtype=torch.bfloat16, device=w.device)
    out[:n, :k] = w.to(torch.bfloat16)
    return out


def conv_weight(w: torch.Tensor) -> torch.Tensor:
    """Reference conv weight [Cout, Cin, kt, kh, kw] -> the K layout svr2_conv3d consumes:
    [Cout, kt, Cin/64, kh, kw, 64] flattened (Cin % 64 == 0), or [Cout, kt, kh, kw, Cin] for the small-Cin
    im2col path."""
    cout, cin, kt, kh, kw = w.shape
    if cin % 64:
        return w.permute(0, 2, 3, 4, 1).reshape(cout, kt * kh * kw * cin)
    return w.reshape(cout, cin // 64, 64, kt, kh, kw).permute(0, 3, 1, 4, 5, 2).reshape(cout, kt * kh * kw * cin)


WL_ROWS, WL_TILED_128x32, WL_CIN3 = 0, 1, 2


def conv_weight_cin3(w: torch.Tensor) -> torch.Tensor:
    """Reference conv_in weight [128, 3, 3, 3, 3] -> SVR2_WL_CIN3 (include/seedvr2_hip.h): bf16 [128, 128] with
    k = (kt*3 + ky)*12 + kx*4 + c; the pad channel c = 3 and k >= 108 are zero."""
    cout, cin, kt, kh, kw = w.shape
    if (cout, cin, kt, kh, kw) != (128, 3, 3, 3, 3):
        raise HipError(f"conv_weight_cin3: expected [128,3,3,3,3], got {tuple(w.shape)}")
    t = torch.zeros(cout, kt, kh, kw, 4, dtype=torch.bfloat16, device=w.device)
    t[..., :3] = w.to(torch.bfloat16).permute(0, 2, 3, 4, 1)                  # [cout, kt, ky, kx, c]
    out = torch.zeros(cout, 128, dtype=torch.bfloat16, device=w.device)
    out[:, :108] = t.reshape(cout, 108)
    return out


def conv_weight_tiled(w: torch.Tensor) -> torch.Tensor:
    """Reference conv weight [Cout, Cin, kt, 3, 3] -> SVR2_WL_TILED_128x32 (include/seedvr2_hip.h):
    bf16 [cout_pad/128, kt, Cin/32, kw, kh, 128, 32], each [128, 32] tile in LDS-image order (chunk slot s of row r
    holds channels 8*(s ^ f(r)), f(r) = {0,2,3,1}[(r>>2)&3]).  Returned as [cout_pad, kt*kh*kw*Cin]."""
    cout, cin, kt, kh, kw = w.shape
    if cin % 32:
        raise HipError("conv_weight_tiled: Cin % 32 != 0")
    cp = (cout + 127) // 128 * 128
    wp = torch.zeros(cp, cin, kt, kh, kw, dtype=torch.bfloat16, device=w.device)
    wp[:cout] = w.to(torch.bfloat16)
    # [nb, r, cs, s, e, kt, kh, kw] -> [nb, kt, cs, kw, kh, r, s, e]
    t = wp.reshape(cp // 128, 128, cin // 32, 4, 8, kt, kh, kw).permute(0, 5, 2, 7, 6, 1, 3, 4)
    r = torch.arange(128, device=w.device)
    f = torch.tensor([0, 2, 3, 1], device=w.device)[(r >> 2) & 3]
    src = torch.arange(4, device=w.device)[None, :] ^ f[:, None]                 # [128, 4]: slot s <- chunk s ^ f(r)
    idx = src[None, None, None, None, None, :, :, None].expand(cp // 128, kt, cin // 32, kw, kh, 128, 4, 8)
    return torch.gather(t, 6, idx).contiguous().reshape(cp, kt * kh * kw * cin)


def interleave_swiglu(w_gate: torch.Tensor, w_in: torch.Tensor) -> torch.Tensor:
    """Rows [8 gate | 8 in] per 16 (SVR2_EPI_SWIGLU).  w_*: [hidden, K]; hidden % 8 == 0."""
    h, k = w_gate.shape
    assert h % 8 == 0 and w_in.shape == w_gate.shape
    return torch.stack((w_gate.reshape(h // 8, 8, k), w_in.reshape(h // 8, 8, k)), dim=1).reshape(2 * h, k)


def gemm(a: torch.Tensor, w: torch.Tensor, n: int, out: Optional[torch.Tensor] = None, *, epilogue: int = EPI_BIAS,
         bias: Optional[torch.Tensor] = None, gate: Optional[torch.Tensor] = None,
         resid: Optional[torch.Tensor] = None, a_rows: Optional[torch.Tensor] = None, m: Optional[int] = None,
         pixshuf: Optional[tuple] = None) -> torch.Tensor:
    """out[M, N'] = epilogue(a[M, K] @ w[:n, K]^T).  `w` is a pad_weight()-style matrix."""
    _bf16c(a, "gemm.a"); _bf16c(w, "gemm.w")
    if a.dim() != 2 or w.dim() != 2 or a.stride(1) != 1 or not w.is_contiguous():
        raise HipError("gemm: a must be [M,K] with unit inner stride, w contiguous [n_pad,K]")
    K = w.shape[1]
    M = int(m if m is not None else (a_rows.numel() if a_rows is not None else a.shape[0]))
    if a.shape[1] != K or K % 64 or w.shape[0] % 128 or w.shape[0] < n:
        raise HipError(f"gemm: shape mismatch a{tuple(a.shape)} w{tuple(w.shape)} n={n}")
    if a_rows is None and a.shape[0] < M:
        raise HipError("gemm: a has fewer rows than M")
    if a_rows is not None and (a_rows.dtype != torch.int32 or not a_rows.is_cuda or a_rows.numel() < M):
        raise HipError("gemm: a_rows must be an int32 device tensor with >= M entries")
    n_out = n // 2 if epilogue == EPI_SWIGLU else n
    d = GemmDesc()
    if epilogue == EPI_PIXSHUF3D:
        frames, h_, w_, c_, tz, drop = pixshuf
        fo = frames * tz - (1 if drop else 0)
        if out is None:
            out = torch.empty(fo, 2 * h_, 2 * w_, c_, dtype=torch.bfloat16, device=a.device)
        if out.numel() != fo * 4 * h_ * w_ * c_ or not out.is_contiguous():
            raise HipError("gemm: pixel-shuffle output has the wrong size")
        d.ps_frames, d.ps_h, d.ps_w, d.ps_c, d.ps_tz, d.ps_drop_head = frames, h_, w_, c_, tz, int(bool(drop))
        ldc = c_
    else:
        if out is None:
            out = torch.empty(M, n_out, dtype=torch.bfloat16, device=a.device)
        _bf16c(out, "gemm.out")
        if out.dim() != 2 or out.shape[0] < M or out.shape[1] < n_out or out.stride(1) != 1:
            raise HipError("gemm: out too small")
        ldc = out.stride(0)
    _f32c(bias, "gemm.bias", n); _f32c(gate, "gemm.gate", n)
    if resid is not None:
        _bf16c(resid, "gemm.resid")
        if resid.dim() != 2 or resid.shape[0] < M or resid.shape[1] < n or resid.stride(1) != 1:
            raise HipError("gemm: resid too small")
    d.M, d.N, d.K = M, n, K
    d.A, d.lda, d.a_rows = a.data_ptr(), a.stride(0), (a_rows.data_ptr() if a_rows is not None else None)
    d.W, d.C, d.ldc, d.epilogue = w.data_ptr(), out.data_ptr(), ldc, epilogue
    d.bias = bias.data_ptr() if bias is not None else None
    d.gate = gate.data_ptr() if gate is not None else None
    d.resid = resid.data_ptr() if resid is not None else None
    d.ldr = resid.stride(0) if resid is not None else 0
    d.zero_page = zero_page(a.device).data_ptr()
    _check(lib().svr2_gemm(C.byref(d), _stream()), "svr2_gemm")
    return out


def conv_out_size(t, hi, wi, k, s, npad, pad_hw):
    """Output extent of the causal conv: pad_hw = (h0, h1, w0, w1) zero padding."""
    kt, kh, kw = k
    st, sh, sw = s
    to = (t + npad - kt) // st + 1
    ho = (hi + pad_hw[0] + pad_hw[1] - kh) // sh + 1
    wo = (wi + pad_hw[2] + pad_hw[3] - kw) // sw + 1
    return to, ho, wo


class GnPartial:
    """Per-tile per-channel (sum, sum of squares) of a conv output: fp32 [F, tiles, 2, C] (svr2_conv3d gn_partial)."""

    def __init__(self, buf: torch.Tensor, tiles: int):
        self.buf, self.tiles = buf, tiles


def _conv_desc(x, w, cout, k, s, npad, pad_hw, carry, bias, resid, y, w_layout=WL_ROWS, gn_partial=None):
    T, Hi, Wi, Cin = x.shape
    To, Ho, Wo = conv_out_size(T, Hi, Wi, k, s, npad, pad_hw)
    d = Conv3dDesc()
    d.x, d.carry, d.w = x.data_ptr(), (carry.data_ptr() if carry is not None else None), w.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.resid = resid.data_ptr() if resid is not None else None
    d.y = y.data_ptr() if y is not None else None
    d.T, d.Hi, d.Wi, d.Cin, d.Cout = T, Hi, Wi, Cin, cout
    d.kt, d.kh, d.kw = k
    d.st, d.sh, d.sw = s
    d.npad, d.pad_h0, d.pad_w0 = npad, pad_hw[0], pad_hw[2]
    d.To, d.Ho, d.Wo = To, Ho, Wo
    d.zero_page = zero_page(x.device).data_ptr()
    d.w_layout = w_layout
    d.gn_partial = gn_partial.data_ptr() if gn_partial is not None else None
    d.carry_frames = carry.shape[0] if carry is not None else 0
    return d, (To, Ho, Wo)


def conv3d(x: torch.Tensor, w: torch.Tensor, cout: int, k, s=(1, 1, 1), *, npad: int, pad_hw=(1, 1, 1, 1),
           carry: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
           resid: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
           w_layout: int = WL_ROWS, gn_stats: bool = False):
    """Causal NDHWC conv.  x [T,Hi,Wi,Cin] bf16; w = pad_weight(conv_weight(w5d)) ([cout_pad, kt*kh*kw*Cin]), or
    conv_weight_tiled(w5d) with w_layout = WL_TILED_128x32, or conv_weight_cin3(w5d) with WL_CIN3 (the fused Cin = 3
    conv_in).  gn_stats (tiled and Cin-3 layouts): also return the per-tile GroupNorm partial sums of the output,
    (out, GnPartial), for groupnorm_silu(..., partial=...)."""
    _bf16c(x, "conv3d.x"); _bf16c(w, "conv3d.w")
    if x.dim() != 4 or not x.is_contiguous() or not w.is_contiguous():
        raise HipError("conv3d: x must be contiguous [T,H,W,C]")
    T, Hi, Wi, Cin = x.shape
    if w_layout == WL_CIN3:
        if tuple(w.shape) != (128, 128) or Cin != 3 or cout != 128 or tuple(k) != (3, 3, 3) or resid is not None:
            raise HipError(f"conv3d: WL_CIN3 is the 3 -> 128 channel 3x3x3 conv without residual (Cin={Cin} cout={cout} k={k})")
    elif w.shape[1] != k[0] * k[1] * k[2] * Cin or w.shape[0] % 128 or w.shape[0] < cout:
        raise HipError(f"conv3d: weight shape {tuple(w.shape)} does not match k={k} Cin={Cin} cout={cout}")
    if carry is not None:
        _bf16c(carry, "conv3d.carry")
        if tuple(carry.shape[1:]) != (Hi, Wi, Cin) or not (1 <= carry.shape[0] <= npad) or not carry.is_contiguous():
            raise HipError(f"conv3d: carry must be [1..{npad},{Hi},{Wi},{Cin}] (fewer frames than npad: the first one also "
                           f"stands for the missing earlier ones), got {tuple(carry.shape)}")
    To, Ho, Wo = conv_out_size(T, Hi, Wi, k, s, npad, pad_hw)
    if To < 1 or Ho < 1 or Wo < 1:
        raise HipError("conv3d: empty output")
    if out is None:
        out = torch.empty(To, Ho, Wo, cout, dtype=torch.bfloat16, device=x.device)
    if tuple(out.shape) != (To, Ho, Wo, cout) or not out.is_contiguous():
        raise HipError("conv3d: out has the wrong shape")
    _f32c(bias, "conv3d.bias", cout)
    if resid is not None and (tuple(resid.shape) != tuple(out.shape) or not resid.is_contiguous()):
        raise HipError("conv3d: resid must match the output shape")
    part = None
    if gn_stats:
        if w_layout not in (WL_TILED_128x32, WL_CIN3):
            raise HipError("conv3d: gn_stats needs the tiled or the Cin-3 weight layout")
        lib().svr2_conv3d_gn_tiles.restype = C.c_int32
        tiles = int(lib().svr2_conv3d_gn_tiles(C.c_int32(Ho), C.c_int32(Wo), C.c_int32(cout)))
        part = GnPartial(torch.empty(To, tiles, 2, cout, dtype=torch.float32, device=x.device), tiles)
    d, _ = _conv_desc(x, w, cout, k, s, npad, pad_hw, carry, bias, resid, out, w_layout,
                      part.buf if part is not None else None)
    if w_layout == WL_TILED_128x32:      # scratch of the split-K path (small grids only; 0 bytes otherwise)
        key = (tuple(x.shape), cout, tuple(k), tuple(s), npad, tuple(pad_hw) if pad_hw is not None else None)
        nws = _SPLIT_WS.get(key)
        if nws is None:
            nws = _SPLIT_WS[key] = int(lib().svr2_conv3d_split_ws_bytes(C.byref(d)))
        if nws > 0:
            ws = torch.empty(nws, dtype=torch.uint8, device=x.device)
            d.split_ws, d.split_ws_bytes = _p(ws), nws
    _check(lib().svr2_conv3d(C.byref(d), _stream()), "svr2_conv3d")
    return (out, part) if gn_stats else out


def im2col3d(x: torch.Tensor, k, s, *, npad: int, pad_hw, carry: Optional[torch.Tensor], kpad: int) -> torch.Tensor:
    _bf16c(x, "im2col3d.x")
    if x.dim() != 4 or not x.is_contiguous():
        raise HipError("im2col3d: x must be contiguous [T,H,W,C]")
    if carry is not None and (tuple(carry.shape[1:]) != tuple(x.shape[1:]) or not (1 <= carry.shape[0] <= npad)
                              or not carry.is_contiguous()):
        raise HipError("im2col3d: bad carry shape")
    d, (To, Ho, Wo) = _conv_desc(x, x, 8, k, s, npad, pad_hw, carry, None, None, None)
    col = torch.empty(To * Ho * Wo, kpad, dtype=torch.bfloat16, device=x.device)
    _check(lib().svr2_im2col3d(C.byref(d), _p(col), C.c_int64(kpad), _stream()), "svr2_im2col3d")
    return col


def rmsnorm_mod(x: torch.Tensor, scale: torch.Tensor, shift: Optional[torch.Tensor], eps: float,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _bf16c(x, "rmsnorm_mod.x")
    if x.dim() != 2 or not x.is_contiguous():
        raise HipError("rmsnorm_mod: x must be contiguous [rows, dim]")
    rows, dim = x.shape
    _f32c(scale, "rmsnorm_mod.scale", dim); _f32c(shift, "rmsnorm_mod.shift", dim)
    if out is None:
        out = torch.empty_like(x)
    _check(lib().svr2_rmsnorm_mod(_p(x), _p(out), _p(scale), _p(shift), C.c_int64(rows), C.c_int32(dim),
                                  C.c_float(eps), _stream()), "svr2_rmsnorm_mod")
    return out


def qkv_prep(qkv_vid, qkv_txt, src_row, pos, qn_vid, kn_vid, qn_txt, kn_txt, rope_cos, rope_sin, heads: int,
             rope_mode: int, eps: float, out=None):
    _bf16c(qkv_vid, "qkv_prep.qkv_vid"); _bf16c(qkv_txt, "qkv_prep.qkv_txt")
    D = 128
    if qkv_vid.shape[1] != 3 * heads * D or qkv_txt.shape[1] != 3 * heads * D:
        raise HipError("qkv_prep: qkv must be [rows, 3*H*128]")
    if not (qkv_vid.is_contiguous() and qkv_txt.is_contiguous()):
        raise HipError("qkv_prep: qkv must be contiguous")
    S = src_row.numel()
    if src_row.dtype != torch.int32 or (pos is not None and (pos.dtype != torch.int32 or pos.numel() != 3 * S)):
        raise HipError("qkv_prep: src_row/pos must be int32, pos [S,3]")
    for t_, nm in ((qn_vid, "qn_vid"), (kn_vid, "kn_vid"), (qn_txt, "qn_txt"), (kn_txt, "kn_txt")):
        _f32c(t_, "qkv_prep." + nm, D)
    if out is None:
        out = torch.empty(3, S, heads, D, dtype=torch.bfloat16, device=qkv_vid.device)
    d = QkvPrepDesc()
    d.qkv_vid, d.qkv_txt, d.src_row, d.pos = qkv_vid.data_ptr(), qkv_txt.data_ptr(), src_row.data_ptr(), \
        (pos.data_ptr() if pos is not None else None)
    d.qn_vid, d.kn_vid, d.qn_txt, d.kn_txt = qn_vid.data_ptr(), kn_vid.data_ptr(), qn_txt.data_ptr(), kn_txt.data_ptr()
    nfreq, max_pos = 0, 0
    if rope_mode != 0:
        _f32c(rope_cos, "qkv_prep.rope_cos"); _f32c(rope_sin, "qkv_prep.rope_sin")
        d.rope_cos, d.rope_sin = rope_cos.data_ptr(), rope_sin.data_ptr()
        if rope_mode == 1:
            max_pos, nfreq = rope_cos.shape
        else:
            nfreq = rope_cos.shape[1] // 3
            if rope_cos.shape[0] != S:
                raise HipError("qkv_prep: per-position rope tables must have S rows")
    d.q, d.k, d.v = out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr()
    d.S, d.H, d.D, d.nfreq, d.max_pos, d.rope_mode, d.eps = S, heads, D, nfreq, max_pos, rope_mode, eps
    _check(lib().svr2_qkv_prep(C.byref(d), _stream()), "svr2_qkv_prep")
    return out


def window_attn(q, k, v, cu_seqlens, tile_seq, tile_q0, scale: float, out=None, out_rows=None):
    for t_, nm in ((q, "q"), (k, "k"), (v, "v")):
        _bf16c(t_, "window_attn." + nm)
        if t_.dim() != 3 or t_.shape[2] != 128 or not t_.is_contiguous() or t_.shape != q.shape:
            raise HipError("window_attn: q,k,v must be contiguous [S,H,128] of equal shape")
    for t_ in (cu_seqlens, tile_seq, tile_q0):
        if t_.dtype != torch.int32 or not t_.is_cuda:
            raise HipError("window_attn: index tensors must be int32 on the device")
    if tile_seq.numel() != tile_q0.numel():
        raise HipError("window_attn: tile list mismatch")
    if out is None:
        out = torch.empty_like(q)
    if out_rows is not None:   # result of position s goes to row out_rows[s] (a permutation of the S rows)
        if out_rows.dtype != torch.int32 or not out_rows.is_cuda or out_rows.numel() != q.shape[0]:
            raise HipError("window_attn: out_rows must be an int32 device tensor with one entry per position")
        _check(lib().svr2_window_attn_scatter(_p(q), _p(k), _p(v), _p(out), _p(out_rows), _p(cu_seqlens), _p(tile_seq),
                                              _p(tile_q0), C.c_int32(tile_seq.numel()), C.c_int32(q.shape[1]),
                                              C.c_int32(128), C.c_float(scale), _stream()), "svr2_window_attn_scatter")
        return out
    _check(lib().svr2_window_attn(_p(q), _p(k), _p(v), _p(out), _p(cu_seqlens), _p(tile_seq), _p(tile_q0),
                                  C.c_int32(tile_seq.numel()), C.c_int32(q.shape[1]), C.c_int32(128),
                                  C.c_float(scale), _stream()), "svr2_window_attn")
    return out


def txt_mean(attn: torch.Tensor, txt_row0: torch.Tensor, lt: int, out=None):
    _bf16c(attn, "txt_mean.attn")
    if attn.dim() != 2 or not attn.is_contiguous() or txt_row0.dtype != torch.int32:
        raise HipError("txt_mean: attn must be contiguous [S, dim], txt_row0 int32")
    dim = attn.shape[1]
    if out is None:
        out = torch.empty(lt, dim, dtype=torch.bfloat16, device=attn.device)
    _check(lib().svr2_txt_mean(_p(attn), _p(txt_row0), C.c_int32(txt_row0.numel()), C.c_int32(lt), C.c_int32(dim),
                               _p(out), _stream()), "svr2_txt_mean")
    return out


def patch_in(noise: torch.Tensor, latent: torch.Tensor, kpad: int) -> torch.Tensor:
    _bf16c(noise, "patch_in.noise"); _bf16c(latent, "patch_in.latent")
    if noise.shape != latent.shape or noise.dim() != 4 or not (noise.is_contiguous() and latent.is_contiguous()):
        raise HipError("patch_in: noise/latent must be contiguous [T,H,W,C] of equal shape")
    T, H, W, Cc = noise.shape
    rows = torch.empty(T * (H // 2) * (W // 2), kpad, dtype=torch.bfloat16, device=noise.device)
    _check(lib().svr2_patch_in(_p(noise), _p(latent), _p(rows), C.c_int32(T), C.c_int32(H), C.c_int32(W),
                               C.c_int32(Cc), C.c_int32(kpad), _stream()), "svr2_patch_in")
    return rows


def patch_out_x0(pred: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    _bf16c(pred, "patch_out_x0.pred"); _bf16c(noise, "patch_out_x0.noise")
    T, H, W, Cc = noise.shape
    if pred.dim() != 2 or pred.shape[0] < T * (H // 2) * (W // 2) or pred.shape[1] < 4 * Cc or pred.stride(1) != 1 \
            or not noise.is_contiguous():
        raise HipError("patch_out_x0: pred must be [L, >=4C]")
    x0 = torch.empty_like(noise)
    _check(lib().svr2_patch_out_x0(_p(pred), C.c_int64(pred.stride(0)), _p(noise), _p(x0), C.c_int32(T),
                                   C.c_int32(H), C.c_int32(W), C.c_int32(Cc), _stream()), "svr2_patch_out_x0")
    return x0


# ----------------------------------------------------------------------------------------- VAE kernels
def groupnorm_silu(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float,
                   silu: bool = True, out: Optional[torch.Tensor] = None, partial: Optional[GnPartial] = None,
                   carry_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Per-frame GroupNorm (+SiLU) of a channels-last activation x [F, H, W, C].
    partial: statistics the producing conv already reduced per tile (conv3d(..., gn_stats=True)) - skips the
    statistics pass over x.  carry_out [k, H, W, C]: also receives the last k output frames (the consumer conv's
    causal state)."""
    _bf16c(x, "groupnorm.x")
    if x.dim() != 4 or not x.is_contiguous():
        raise HipError("groupnorm: x must be contiguous [F,H,W,C]")
    Fr, H, W, Cc = x.shape
    _f32c(gamma, "groupnorm.gamma", Cc); _f32c(beta, "groupnorm.beta", Cc)
    P = H * W
    coef = torch.empty(Fr, Cc, 2, dtype=torch.float32, device=x.device)
    if partial is not None:
        if tuple(partial.buf.shape) != (Fr, partial.tiles, 2, Cc) or not partial.buf.is_contiguous():
            raise HipError("groupnorm: partial sums do not match x")
        _check(lib().svr2_groupnorm_finalize(_p(partial.buf), C.c_int32(partial.tiles), C.c_int32(Fr), C.c_int64(P),
                                             C.c_int32(Cc), C.c_int32(groups), C.c_float(eps), _p(gamma), _p(beta),
                                             _p(coef), _stream()), "svr2_groupnorm_finalize")
    else:
        nws = lib().svr2_groupnorm_ws_floats(C.c_int32(Fr), C.c_int64(P), C.c_int32(Cc))
        if nws <= 0:
            raise HipError(f"groupnorm: unsupported shape {tuple(x.shape)}")
        ws = torch.empty(nws, dtype=torch.float32, device=x.device)
        _check(lib().svr2_groupnorm_stats(_p(x), C.c_int32(Fr), C.c_int64(P), C.c_int32(Cc), C.c_int32(groups),
                                          C.c_float(eps), _p(gamma), _p(beta), _p(ws), _p(coef), _stream()),
               "svr2_groupnorm_stats")
    if out is None:
        out = torch.empty_like(x)
    nc = 0
    if carry_out is not None:
        _bf16c(carry_out, "groupnorm.carry_out")
        nc = carry_out.shape[0]
        if carry_out.dim() != 4 or tuple(carry_out.shape[1:]) != (H, W, Cc) or not carry_out.is_contiguous() or nc > Fr:
            raise HipError("groupnorm: carry_out must be contiguous [k <= F, H, W, C]")
    _check(lib().svr2_groupnorm_apply(_p(x), _p(coef), _p(out), C.c_int32(Fr), C.c_int64(P), C.c_int32(Cc),
                                      C.c_int32(1 if silu else 0), _p(carry_out) if carry_out is not None else None,
                                      C.c_int32(nc), _stream()), "svr2_groupnorm_apply")
    return out


def attn_d512(qkv: torch.Tensor, frames: int, tokens: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """qkv bf16 [frames*tokens, 1536] (q | k | v) -> out [frames*tokens, 512]."""
    _bf16c(qkv, "attn_d512.qkv")
    if qkv.dim() != 2 or qkv.shape[1] != 1536 or qkv.shape[0] != frames * tokens or not qkv.is_contiguous():
        raise HipError("attn_d512: qkv must be contiguous [frames*tokens, 1536]")
    if out is None:
        out = torch.empty(frames * tokens, 512, dtype=torch.bfloat16, device=qkv.device)
    base = qkv.data_ptr()
    _check(lib().svr2_attn_d512(C.c_void_p(base), C.c_void_p(base + 1024), C.c_void_p(base + 2048), C.c_int64(1536),
                                _p(out), C.c_int64(512), C.c_int32(frames), C.c_int32(tokens),
                                C.c_float(1.0 / (512 ** 0.5)), _stream()), "svr2_attn_d512")
    return out


class ResizeTaps(tuple):
    """(wy, ymin, ycnt, wx, xmin, xcnt) device tables plus `bounds` = (max(ymin+ycnt), max(xmin+xcnt), max(ycnt), max(xcnt))
    taken on the host when the tables were built, so the wrappers can check them against the input size without a device
    reduction (and the host synchronisation that reading one back costs on every call)."""

    def __new__(cls, tables, bounds):
        obj = super().__new__(cls, tables)
        obj.bounds = tuple(int(b) for b in bounds)
        return obj


_TAPS_SYNC = os.environ.get("SVR2_TAPS_SYNC") == "1"       # A/B switch, resolved once at import


def _taps_bounds(taps):
    b = getattr(taps, "bounds", None)
    if b is None or _TAPS_SYNC:   # plain tuple from an outside caller (or the A/B switch):
        # reduce on the device -- this synchronises the host with the stream
        _, ymin, ycnt, _, xmin, xcnt = taps
        b = (int((ymin + ycnt).max()), int((xmin + xcnt).max()), int(ycnt.max()), int(xcnt.max()))
    return b


def resize_pad_normalize(frames: torch.Tensor, taps, hp: int, wp: int) -> torch.Tensor:
    """frames bf16 [T,Hin,Win,3] in [0,1] -> bf16 [T,hp,wp,3] model input; taps from pipeline.resize_taps()."""
    _bf16c(frames, "resize.frames")
    if frames.dim() != 4 or frames.shape[3] != 3 or not frames.is_contiguous():
        raise HipError("resize: frames must be contiguous [T,H,W,3]")
    T, Hin, Win, _ = frames.shape
    wy, ymin, ycnt, wx, xmin, xcnt = taps
    Hout, KY = wy.shape
    Wout, KX = wx.shape
    ymax, xmax, kymax, kxmax = _taps_bounds(taps)
    if ymax > Hin or xmax > Win or kymax > KY or kxmax > KX:
        raise HipError("resize: tap table reaches outside the input")
    if hp < Hout or wp < Wout:
        raise HipError("resize: padded size smaller than the resized image")
    out = torch.empty(T, hp, wp, 3, dtype=torch.bfloat16, device=frames.device)
    _check(lib().svr2_resize_pad_normalize(_p(frames), _p(out), _p(wy), _p(ymin), _p(ycnt), _p(wx), _p(xmin), _p(xcnt),
                                           C.c_int32(T), C.c_int32(Hin), C.c_int32(Win), C.c_int32(Hout),
                                           C.c_int32(Wout), C.c_int32(hp), C.c_int32(wp), C.c_int32(KY), C.c_int32(KX),
                                           _stream()), "svr2_resize_pad_normalize")
    return out


def resize_bicubic_aa(frames: torch.Tensor, taps) -> torch.Tensor:
    """frames bf16 [T,Hin,Win,3] -> bf16 [T,Hout,Wout,3]: the antialiased bicubic resize alone (bf16-rounded)."""
    _bf16c(frames, "resize.frames")
    if frames.dim() != 4 or frames.shape[3] != 3 or not frames.is_contiguous():
        raise HipError("resize: frames must be contiguous [T,H,W,3]")
    T, Hin, Win, _ = frames.shape
    wy, ymin, ycnt, wx, xmin, xcnt = taps
    Hout, KY = wy.shape
    Wout, KX = wx.shape
    ymax, xmax, kymax, kxmax = _taps_bounds(taps)
    if ymax > Hin or xmax > Win or kymax > KY or kxmax > KX:
        raise HipError("resize: tap table reaches outside the input")
    out = torch.empty(T, Hout, Wout, 3, dtype=torch.bfloat16, device=frames.device)
    _check(lib().svr2_resize_bicubic_aa(_p(frames), _p(out), _p(wy), _p(ymin), _p(ycnt), _p(wx), _p(xmin), _p(xcnt),
                                        C.c_int32(T), C.c_int32(Hin), C.c_int32(Win), C.c_int32(Hout), C.c_int32(Wout),
                                        C.c_int32(KY), C.c_int32(KX), _stream()), "svr2_resize_bicubic_aa")
    return out


def blend_overlap(prev_tail: torch.Tensor, cur_head: torch.Tensor, w_prev: torch.Tensor, out: torch.Tensor):
    _bf16c(prev_tail, "blend.prev"); _bf16c(cur_head, "blend.cur"); _bf16c(out, "blend.out")
    ov = prev_tail.shape[0]
    if prev_tail.shape != cur_head.shape or out.shape != prev_tail.shape or w_prev.numel() != ov:
        raise HipError("blend: shape mismatch")
    if not (prev_tail.is_contiguous() and cur_head.is_contiguous() and out.is_contiguous()):
        raise HipError("blend: tensors must be contiguous")
    _f32c(w_prev, "blend.w_prev", ov)
    _check(lib().svr2_blend_overlap(_p(prev_tail), _p(cur_head), _p(out), _p(w_prev),
                                    C.c_int64(prev_tail[0].numel()), C.c_int32(ov), _stream()), "svr2_blend_overlap")
    return out


def crop_to_unit(x: torch.Tensor, h: int, w: int, to_unit: bool, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _bf16c(x, "crop.x")
    if x.dim() != 4 or x.shape[3] != 3 or not x.is_contiguous() or h > x.shape[1] or w > x.shape[2]:
        raise HipError("crop: x must be contiguous [T,Hp,Wp,3] with Hp >= h, Wp >= w")
    if out is None:
        out = torch.empty(x.shape[0], h, w, 3, dtype=torch.bfloat16, device=x.device)
    _check(lib().svr2_crop_to_unit(_p(x), _p(out), C.c_int32(x.shape[0]), C.c_int32(x.shape[1]), C.c_int32(x.shape[2]),
                                   C.c_int32(h), C.c_int32(w), C.c_int32(1 if to_unit else 0), _stream()),
           "svr2_crop_to_unit")
    return out


# ----------------------------------------------------------------------------------------------- VAE tiling (next-4)
def tile_accumulate(tile: torch.Tensor, result: torch.Tensor, count: torch.Tensor, y0: int, x0: int, eh: int, ew: int,
                    wh: torch.Tensor, ww: torch.Tensor) -> None:
    """result[:, y0:y0+eh, x0:x0+ew, :C] += tile[:, :eh, :ew, :C] * wh * ww (bf16 roundings of the reference), count too."""
    for t, n in ((tile, "tile"), (result, "result"), (count, "count"), (wh, "wh"), (ww, "ww")):
        _bf16c(t, f"tile_accumulate.{n}")
        if not t.is_contiguous():
            raise HipError(f"tile_accumulate.{n}: must be contiguous")
    if tile.dim() != 4 or result.dim() != 4 or tile.shape[0] != result.shape[0]:
        raise HipError(f"tile_accumulate: tile {tuple(tile.shape)} / result {tuple(result.shape)} must be [T,h,w,C] with equal T")
    T, H, W, Cc = result.shape
    if tuple(count.shape) != (H, W) or wh.numel() < eh or ww.numel() < ew:
        raise HipError("tile_accumulate: count must be [H,W] and the weights at least eh / ew long")
    _check(lib().svr2_tile_accumulate(_p(tile), C.c_int32(T), C.c_int32(tile.shape[1]), C.c_int32(tile.shape[2]),
                                      C.c_int32(tile.shape[3]), _p(result), _p(count), C.c_int32(H), C.c_int32(W),
                                      C.c_int32(Cc), C.c_int32(y0), C.c_int32(x0), C.c_int32(eh), C.c_int32(ew), _p(wh),
                                      _p(ww), _stream()), "svr2_tile_accumulate")


def tile_normalize(result: torch.Tensor, count: torch.Tensor) -> torch.Tensor:
    _bf16c(result, "tile_normalize.result")
    _bf16c(count, "tile_normalize.count")
    T, H, W, Cc = result.shape
    if not result.is_contiguous() or not count.is_contiguous() or tuple(count.shape) != (H, W):
        raise HipError("tile_normalize: result [T,H,W,C] and count [H,W] must be contiguous")
    _check(lib().svr2_tile_normalize(_p(result), _p(count), C.c_int32(T), C.c_int32(H), C.c_int32(W), C.c_int32(Cc),
                                     _stream()), "svr2_tile_normalize")
    return result


# ----------------------------------------------------------------------------------------------- JPEG (next-3)
_jpeg_tables = {}


def jpeg_encode(frames: torch.Tensor, quality: int = 90):
    """frames bf16 [T,H,W,3] in [0,1] -> (packed uint8 [T, cap] entropy-coded scans, lengths int32 [T]) on the device.
    Header / EOI: jpeg.frame_header(W, H, quality) + scan + b"\xff\xd9"."""
    from . import jpeg as _jpeg
    _frames3(frames, "jpeg_encode.frames")
    T, H, W, _ = frames.shape
    dev = frames.device
    key = (dev.index, int(quality))
    if key not in _jpeg_tables:
        div, huff = _jpeg.device_tables(quality)
        _jpeg_tables[key] = (torch.from_numpy(div).to(dev), torch.from_numpy(huff.view("int32").copy()).to(dev))
    div, huff = _jpeg_tables[key]
    cap = H * W * 3 // 2 + 4096
    ws_bytes = int(lib().svr2_jpeg_ws_bytes(C.c_int32(T), C.c_int32(H), C.c_int32(W)))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    packed = torch.empty(T, cap, dtype=torch.uint8, device=dev)
    lens = torch.empty(T, dtype=torch.int32, device=dev)
    _check(lib().svr2_jpeg_encode(_p(frames), C.c_int32(T), C.c_int32(H), C.c_int32(W), _p(div), _p(huff), _p(packed),
                                  C.c_int64(cap), _p(lens), _p(ws), C.c_int64(ws_bytes), _stream()), "svr2_jpeg_encode")
    return packed, lens


# ----------------------------------------------------------------------------------------------- RGBA alpha (next-4)
def alpha_guide_and_edges(rgb: torch.Tensor, norm_guide: bool, norm_edges: bool, want_edges: bool = True):
    """rgb bf16 [T,H,W,3] -> (grey guide fp32 [T,H,W], Sobel edge map fp32 [T,H,W] in [0,1] or None)
    [ref: alpha_upscaling.py:124-189 (detect_edges_batch), :217-218]."""
    _frames3(rgb, "alpha_guide_and_edges.rgb")
    T, H, W, _ = rgb.shape
    dev, st = rgb.device, _stream()
    gray = torch.empty(T, H, W, dtype=torch.uint8, device=dev)
    guide = torch.empty(T, H, W, dtype=torch.float32, device=dev)
    _check(lib().svr2_alpha_guide(_p(rgb), _p(gray), _p(guide), C.c_int64(T * H * W), C.c_int32(bool(norm_guide)),
                                  C.c_int32(bool(norm_edges)), st), "svr2_alpha_guide")
    if not want_edges:
        return guide, None
    mag = torch.empty(T, H, W, dtype=torch.float64, device=dev)
    fmax = torch.empty(T, dtype=torch.int64, device=dev)
    _check(lib().svr2_alpha_sobel(_p(gray), _p(mag), _p(fmax), C.c_int32(T), C.c_int32(H), C.c_int32(W), st), "svr2_alpha_sobel")
    edges = torch.empty(T, H, W, dtype=torch.float32, device=dev)
    _check(lib().svr2_alpha_edges(_p(mag), _p(fmax), _p(edges), C.c_int32(T), C.c_int64(H * W), st), "svr2_alpha_edges")
    return guide, edges


def alpha_upscale(alpha: torch.Tensor, rgb: torch.Tensor, taps) -> torch.Tensor:
    """edge_guided_alpha_upscale [ref: src/core/alpha_upscaling.py:293-430]: alpha bf16 [T,h,w] in [0,1] (the input's alpha
    plane), rgb bf16 [T,H,W,3] (the upscaled video, [-1,1]) -> fp32 [T,H,W] in [0,1].  `taps` = the six antialiased-bicubic
    tap tensors for (h,w) -> (H,W) (pipeline.aa_bicubic_taps, on the device).  Two data-dependent branches of the
    reference (binary mask?, rgb.min() < 0?) are read back from the device (8 + 4 bytes)."""
    _bf16c(alpha, "alpha_upscale.alpha")
    _frames3(rgb, "alpha_upscale.rgb")
    if alpha.dim() != 3 or not alpha.is_contiguous() or alpha.shape[0] != rgb.shape[0]:
        raise HipError(f"alpha_upscale: alpha {tuple(alpha.shape)} must be contiguous [T,h,w] with T = {rgb.shape[0]} "
                       "(the reference fails on a frame-count mismatch as well)")
    T, h, w = alpha.shape
    _, H, W, _ = rgb.shape
    dev, st = rgb.device, _stream()
    counts = torch.empty(2, dtype=torch.int32, device=dev)
    rmin = torch.empty(1, dtype=torch.float32, device=dev)
    ws = torch.empty(1024, dtype=torch.float32, device=dev)
    _check(lib().svr2_alpha_stats(_p(alpha), C.c_int64(alpha.numel()), _p(rgb), C.c_int64(rgb.numel()), _p(counts), _p(rmin),
                                  _p(ws), st), "svr2_alpha_stats")
    c0, c1 = counts.tolist()
    # binary_ratio = (near_zero + near_one) / numel in fp32, > 0.95   (:325-330)
    ratio = (torch.tensor(float(c0), dtype=torch.float32) + torch.tensor(float(c1), dtype=torch.float32)) / alpha.numel()
    binary = bool(ratio > 0.95)
    mn = float(rmin.item())
    norm_guide = mn < 0
    mn2 = (torch.tensor(mn, dtype=torch.float32) + 1) / 2 if norm_guide else torch.tensor(mn)
    norm_edges = bool(mn2 < 0)
    f32 = lambda: torch.empty(T, H, W, dtype=torch.float32, device=dev)
    guide, edges = alpha_guide_and_edges(rgb, norm_guide, norm_edges, want_edges=binary)
    up = f32()
    a32 = alpha.float()
    wy, ymin, ycnt, wx, xmin, xcnt = taps
    _check(lib().svr2_resize_plane(_p(a32), _p(up), _p(wy), _p(ymin), _p(ycnt), _p(wx), _p(xmin), _p(xcnt), C.c_int32(T),
                                   C.c_int32(h), C.c_int32(w), C.c_int32(H), C.c_int32(W), C.c_int32(wy.shape[1]),
                                   C.c_int32(wx.shape[1]), C.c_int32(1), st), "svr2_resize_plane")
    r = 2 if binary else 3
    a, b = f32(), f32()
    _check(lib().svr2_guided_ab(_p(guide), _p(up), _p(a), _p(b), C.c_int32(T), C.c_int32(H), C.c_int32(W), C.c_int32(r),
                                C.c_float(0.002), st), "svr2_guided_ab")
    refined = up   # reuse
    _check(lib().svr2_guided_out(_p(a), _p(b), _p(guide), _p(refined), C.c_int32(T), C.c_int32(H), C.c_int32(W), C.c_int32(r),
                                 st), "svr2_guided_out")
    out = a        # reuse
    _check(lib().svr2_alpha_refine(_p(refined), _p(edges), _p(out), C.c_int32(T), C.c_int32(H), C.c_int32(W),
                                   C.c_int32(1 if binary else 0), st), "svr2_alpha_refine")
    return out


# ----------------------------------------------------------------------------------------------- frame I/O (next-3)
PIX_SWAP_RB, PIX_BOTTOM_UP, PIX_ALPHA = 1, 2, 4


def _u8c(t: torch.Tensor, name: str, nbytes: int):
    if t.dtype != torch.uint8 or not t.is_cuda or not t.is_contiguous() or t.numel() < nbytes:
        raise HipError(f"{name}: expected a contiguous uint8 device tensor of at least {nbytes} bytes")


def u8_to_frames(raw: torch.Tensor, out: torch.Tensor, row_stride: int, frame_stride: int, flags: int = 0) -> torch.Tensor:
    """raw uint8 (packed rows) -> out bf16 [T,H,W,3] in [0,1] = bf16(fp16(u8 / 255))."""
    _frames3(out, "u8_to_frames.out", channels=4 if flags & PIX_ALPHA else 3)
    T, H, W, _ = out.shape
    _u8c(raw, "u8_to_frames.raw", T * frame_stride)
    _check(lib().svr2_u8_to_frames(_p(raw), _p(out), C.c_int32(T), C.c_int32(H), C.c_int32(W), C.c_int64(row_stride),
                                   C.c_int64(frame_stride), C.c_int32(flags), _stream()), "svr2_u8_to_frames")
    return out


def frames_to_u8(frames: torch.Tensor, out: torch.Tensor, row_stride: int, frame_stride: int, flags: int = 0) -> torch.Tensor:
    """frames bf16 [T,H,W,3] in [0,1] -> packed uint8 rows, trunc(x * 255)."""
    _frames3(frames, "frames_to_u8.frames", channels=4 if flags & PIX_ALPHA else 3)
    T, H, W, _ = frames.shape
    _u8c(out, "frames_to_u8.out", T * frame_stride)
    _check(lib().svr2_frames_to_u8(_p(frames), _p(out), C.c_int32(T), C.c_int32(H), C.c_int32(W), C.c_int64(row_stride),
                                   C.c_int64(frame_stride), C.c_int32(flags), _stream()), "svr2_frames_to_u8")
    return out


def frames_to_yuv(frames: torch.Tensor, out: torch.Tensor, sub: int, frame_stride: int) -> torch.Tensor:
    """frames bf16 [T,H,W,3] in [0,1] -> planar 8-bit BT.601 studio-range Y, U, V (sub = 1: 4:4:4, 2: 4:2:0)."""
    _frames3(frames, "frames_to_yuv.frames")
    T, H, W, _ = frames.shape
    _u8c(out, "frames_to_yuv.out", T * frame_stride)
    _check(lib().svr2_frames_to_yuv(_p(frames), _p(out), C.c_int32(T), C.c_int32(H), C.c_int32(W), C.c_int32(sub),
                                    C.c_int64(frame_stride), _stream()), "svr2_frames_to_yuv")
    return out


def yuv_to_frames(raw: torch.Tensor, out: torch.Tensor, sub: int, frame_stride: int) -> torch.Tensor:
    _frames3(out, "yuv_to_frames.out")
    T, H, W, _ = out.shape
    _u8c(raw, "yuv_to_frames.raw", T * frame_stride)
    _check(lib().svr2_yuv_to_frames(_p(raw), _p(out), C.c_int32(T), C.c_int32(H), C.c_int32(W), C.c_int32(sub),
                                    C.c_int64(frame_stride), _stream()), "svr2_yuv_to_frames")
    return out


def noise_mix(x: torch.Tensor, y: torch.Tensor, a: float, b: float, mode: int) -> torch.Tensor:
    """svr2_noise_mix on two bf16 device tensors of equal shape (see include/seedvr2_hip.h for the three modes)."""
    _bf16c(x, "noise_mix.x")
    _bf16c(y, "noise_mix.y")
    if x.shape != y.shape or not x.is_contiguous() or not y.is_contiguous():
        raise HipError("noise_mix: x and y must be contiguous and of equal shape")
    out = torch.empty_like(x)
    _check(lib().svr2_noise_mix(_p(x), _p(y), _p(out), C.c_int64(x.numel()), C.c_float(a), C.c_float(b), C.c_int32(mode),
                                _stream()), "svr2_noise_mix")
    return out


# ----------------------------------------------------------------------------------------------- colour correction
def _frames3(t: torch.Tensor, name: str, dtype=torch.bfloat16, channels: int = 3):
    if t.dtype != dtype or not t.is_cuda:
        raise HipError(f"{name}: expected a {dtype} device tensor, got {t.dtype} on {t.device}")
    if t.dim() != 4 or t.shape[3] != channels or not t.is_contiguous():
        raise HipError(f"{name}: expected contiguous [F,H,W,{channels}]")


def wavelet_decomposition(x: torch.Tensor, levels: int = 5, want_high: bool = True):
    """[F,H,W,3] bf16 (or fp32) -> (high, low) of the reference's dilated-Gaussian pyramid (high None if not wanted)."""
    f32 = x.dtype == torch.float32
    _frames3(x, "wavelet.x", x.dtype if f32 else torch.bfloat16)
    Fr, H, W, _ = x.shape
    high = torch.empty_like(x) if want_high else None
    img = x
    for i in range(levels):
        radius = min(2 ** i, max(1, min(H, W) // 8))
        low = torch.empty_like(x)
        _check(lib().svr2_wavelet_level(_p(img), _p(low), _p(high), C.c_int32(Fr), C.c_int32(H), C.c_int32(W),
                                        C.c_int32(radius), C.c_int32(1 if i == 0 else 0), C.c_int32(int(f32)), _stream()),
               "svr2_wavelet_level")
        img = low
    return high, img


def wavelet_reconstruction(content: torch.Tensor, style: torch.Tensor) -> torch.Tensor:
    if content.shape != style.shape or content.dtype != style.dtype:
        raise HipError("wavelet_reconstruction: content and style must have the same shape and dtype")
    high, _ = wavelet_decomposition(content, want_high=True)
    _, low = wavelet_decomposition(style, want_high=False)
    out = torch.empty_like(content)
    _check(lib().svr2_wavelet_merge(_p(high), _p(low), _p(out), C.c_int64(out.numel()),
                                    C.c_int32(int(content.dtype == torch.float32)), _stream()), "svr2_wavelet_merge")
    return out


def adain(content: torch.Tensor, style: torch.Tensor) -> torch.Tensor:
    _frames3(content, "adain.content"); _frames3(style, "adain.style")
    if content.shape != style.shape:
        raise HipError("adain: content and style must have the same shape")
    Fr, H, W, _ = content.shape
    ws = torch.empty(int(lib().svr2_adain_ws_bytes(C.c_int32(Fr), C.c_int64(H * W))), dtype=torch.uint8, device=content.device)
    out = torch.empty_like(content)
    _check(lib().svr2_adain(_p(content), _p(style), _p(out), C.c_int32(Fr), C.c_int64(H * W), _p(ws), _stream()), "svr2_adain")
    return out


def lab_color_transfer(content: torch.Tensor, style: torch.Tensor, luminance_weight: float = 0.8) -> torch.Tensor:
    """Wavelet reconstruction, then CIELAB histogram matching of a*, b* (and partly L*) over the whole batch."""
    base = wavelet_reconstruction(content, style)
    n = base.numel() // 3
    dev = base.device
    cl = torch.empty(3, n, dtype=torch.float32, device=dev)
    sl = torch.empty(3, n, dtype=torch.float32, device=dev)
    _check(lib().svr2_rgb_to_lab(_p(base), _p(cl), C.c_int64(n), _stream()), "svr2_rgb_to_lab")
    _check(lib().svr2_rgb_to_lab(_p(style), _p(sl), C.c_int64(n), _stream()), "svr2_rgb_to_lab")
    nws = int(lib().svr2_hist_match_ws_bytes(C.c_int64(n)))
    if nws <= 0:
        raise HipError("lab_color_transfer: batch too large for one histogram match")
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    matched = torch.empty(3, n, dtype=torch.float32, device=dev)
    chans = (0, 1, 2) if luminance_weight < 1.0 else (1, 2)
    for c in chans:
        _check(lib().svr2_hist_match(_p(cl[c]), _p(sl[c]), _p(matched[c]), C.c_int64(n), _p(ws), C.c_int64(nws), _stream()),
               "svr2_hist_match")
    out = torch.empty_like(base)
    _check(lib().svr2_lab_to_rgb(_p(cl[0]), _p(matched[0]) if luminance_weight < 1.0 else None, _p(matched[1]),
                                 _p(matched[2]), C.c_float(luminance_weight), _p(out), C.c_int64(n), _stream()),
           "svr2_lab_to_rgb")
    return out


_HUE_THRESHOLDS = None


def _hue_thresholds():
    """fp32(k * (1/12)) for k = 0..12 and fp32(1 - 1/12): the bin edges the reference compares fp32 hues against
    (Python doubles cast to the tensor dtype) [ref: color_fix.py:704-719]."""
    global _HUE_THRESHOLDS
    if _HUE_THRESHOLDS is None:
        import numpy as np
        bw = 1.0 / 12
        vals = [np.float32(k * bw) for k in range(12)] + [np.float32(12 * bw), np.float32(1.0 - bw)]
        _HUE_THRESHOLDS = (C.c_float * 14)(*[float(v) for v in vals])
    return _HUE_THRESHOLDS


def hsv_saturation_match(content: torch.Tensor, style: torch.Tensor, out_f32: bool = False) -> torch.Tensor:
    """Hue-conditional saturation histogram matching: content keeps its hue and value, its saturation takes the style's
    distribution per 30-degree hue bin."""
    _frames3(content, "hsv.content"); _frames3(style, "hsv.style")
    if content.shape != style.shape:
        raise HipError("hsv: content and style must have the same shape")
    n = content.numel() // 3
    dev = content.device
    hc = torch.empty(3, n, dtype=torch.float32, device=dev)
    hs = torch.empty(3, n, dtype=torch.float32, device=dev)
    _check(lib().svr2_rgb_to_hsv(_p(content), _p(hc), C.c_int64(n), _stream()), "svr2_rgb_to_hsv")
    _check(lib().svr2_rgb_to_hsv(_p(style), _p(hs), C.c_int64(n), _stream()), "svr2_rgb_to_hsv")
    nws = int(lib().svr2_hue_sat_match_ws_bytes(C.c_int64(n)))
    if nws <= 0:
        raise HipError("hsv: batch too large for one histogram match")
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    ms = torch.empty(n, dtype=torch.float32, device=dev)
    _check(lib().svr2_hue_sat_match(_p(hc), _p(hs), _p(ms), C.c_int64(n), _hue_thresholds(), C.c_int32(100), _p(ws),
                                    C.c_int64(nws), _stream()), "svr2_hue_sat_match")
    out = torch.empty(content.shape, dtype=torch.float32 if out_f32 else torch.bfloat16, device=dev)
    _check(lib().svr2_hsv_to_rgb(_p(hc[0]), _p(ms), _p(hc[2]), _p(out), C.c_int64(n), C.c_int32(int(out_f32)), _stream()),
           "svr2_hsv_to_rgb")
    return out


def wavelet_adaptive(content: torch.Tensor, style: torch.Tensor) -> torch.Tensor:
    """Wavelet base (in fp32 like the reference), HSV saturation correction blended into the oversaturated regions."""
    _frames3(content, "wavelet_adaptive.content"); _frames3(style, "wavelet_adaptive.style")
    wav = wavelet_reconstruction(content.float(), style.float())
    hsv = hsv_saturation_match(content, style, out_f32=True)
    out = torch.empty_like(content)
    _check(lib().svr2_sat_blend(_p(content), _p(style), _p(wav), _p(hsv), _p(out), C.c_int64(content.numel() // 3),
                                _stream()), "svr2_sat_blend")
    return out


COLOR_METHODS = {"lab": lab_color_transfer, "wavelet": wavelet_reconstruction, "adain": adain,
                 "hsv": hsv_saturation_match, "wavelet_adaptive": wavelet_adaptive}


def color_correct(sample: torch.Tensor, input_video: torch.Tensor, method: str) -> torch.Tensor:
    """Phase-4 colour correction of a decoded batch [T,H,W,3] in [-1,1] against the transformed input of the same frames
    [ref: src/core/generation_phases.py:1290-1304]."""
    if method not in COLOR_METHODS:
        raise HipError(f"unknown color_correction '{method}' (available: none, {', '.join(COLOR_METHODS)})")
    return COLOR_METHODS[method](sample, input_video)
