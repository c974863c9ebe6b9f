"""GPU parity tests of the individual HIP kernels, called through the C ABI (ctypes).

Reference = fp32 PyTorch on the CPU applied to the SAME bf16-rounded inputs (floating-point kernels),
or the oracle's functions (oracle/seedvr2_ref).  Tolerance: bf16 output rounding (2^-8 relative) plus
accumulation-order noise -> |d| <= 1.6e-2*rms(ref) + 1.6e-2*|ref| unless a test states otherwise.
"""

import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests_support import assert_close, bf

pytestmark = pytest.mark.gpu


def dev_bf16(x, gpu):
    return x.to(gpu).to(torch.bfloat16).contiguous()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return bf(torch.randn(*shape, generator=g) * scale)


# ----------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(300, 200, 192), (128, 128, 64), (1000, 2560, 2560), (58, 7680, 256), (129, 64, 6912)])
def test_gemm_bias(pkg, gpu, M, N, K):
    hip = pkg.hip
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3)
    out = hip.gemm(dev_bf16(a, gpu), hip.pad_weight(w.to(gpu)), N, bias=b.to(gpu))
    assert_close(out, a @ w.T + b, f"gemm {M}x{N}x{K}")


@pytest.mark.parametrize("M,N,K,epi", [(1024, 2560, 2560, "gate"), (58, 2560, 6912, "gate"), (1024, 7680, 2560, "bias"),
                                       (1000, 7680, 1024, "gelu"), (58, 13824, 2560, "swiglu")])
def test_tile_starved_gemm_geometries_bit_identical_to_the_two_deep_ring(pkg, gpu, M, N, K, epi, lib_option):
    """Tile-starved dense GEMMs (<= 256 tiles, K >= 1024) take a 3-deep LDS ring, or one 256-row workgroup per CU when
    128-row tiles would need a second round (gemm.hip launch()): same K order as the 2-deep 128-row geometry that
    SVR2_GEMM_BM=128 keeps, so the results are the same bits.  [ref: F.linear calls of mmattn.py:173,269, mlp.py:60-61]"""
    hip = pkg.hip
    a, w = dev_bf16(rnd(M, K, seed=1), gpu), hip.pad_weight(rnd(N, K, seed=2, scale=1 / math.sqrt(K)).to(gpu))
    bias, gate, res = rnd(N, seed=3).to(gpu), rnd(N, seed=4).to(gpu), dev_bf16(rnd(M, N, seed=5), gpu)

    def run():
        if epi == "gate":
            return hip.gemm(a, w, N, epilogue=hip.EPI_GATE_RES, bias=bias, gate=gate, resid=res, m=M)
        if epi == "gelu":
            return hip.gemm(a, w, N, epilogue=hip.EPI_GELU, bias=bias)
        if epi == "swiglu":
            return hip.gemm(a, w, N, epilogue=hip.EPI_SWIGLU)
        return hip.gemm(a, w, N, bias=bias)

    got = run()
    lib_option("SVR2_GEMM_BM", 128)
    want = run()
    assert torch.isfinite(got.float()).all() and got.float().abs().max() > 0
    assert torch.equal(got, want)


@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (1000, 512, 384), (777, 2560, 2560), (256, 256, 6912), (513, 768, 192)])
def test_gemm4_pingpong_bias(pkg, gpu, M, N, K, lib_option):
    """gemm4.hip (ping-pong 256x256x64 kernel, five LDS sub-stages): forced onto small shapes with SVR2_GEMM_V=3."""
    lib_option("SVR2_GEMM_V", 3)
    hip = pkg.hip
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3)
    out = hip.gemm(dev_bf16(a, gpu), hip.pad_weight(w.to(gpu)), N, bias=b.to(gpu))
    assert_close(out, a @ w.T + b, f"gemm4 {M}x{N}x{K}")


def test_gemm4_pingpong_epilogues(pkg, gpu, lib_option):
    lib_option("SVR2_GEMM_V", 3)
    hip = pkg.hip
    M, N, K, S = 700, 256, 256, 900
    a, w = rnd(S, K, seed=4), rnd(N, K, seed=5, scale=1 / 16)
    bias, gate, res = rnd(N, seed=6), rnd(N, seed=7), rnd(M, N, seed=8)
    rows = torch.randperm(S, generator=torch.Generator().manual_seed(9))[:M].to(torch.int32)
    out = hip.gemm(dev_bf16(a, gpu), hip.pad_weight(w.to(gpu)), N, epilogue=hip.EPI_GATE_RES, bias=bias.to(gpu),
                   gate=gate.to(gpu), resid=dev_bf16(res, gpu), a_rows=rows.to(gpu), m=M)
    assert_close(out, res + gate * (a[rows.long()] @ w.T + bias), "gemm4 gather+gate+res")
    Hd = 256
    wg, wi = rnd(Hd, K, seed=11, scale=0.1), rnd(Hd, K, seed=12, scale=0.1)
    out = hip.gemm(dev_bf16(a[:M], gpu), hip.pad_weight(hip.interleave_swiglu(wg, wi).to(gpu)), 2 * Hd,
                   epilogue=hip.EPI_SWIGLU)
    assert_close(out, F.silu(a[:M] @ wg.T) * (a[:M] @ wi.T), "gemm4 swiglu")
    b2 = rnd(N, seed=13)
    out = hip.gemm(dev_bf16(a[:M], gpu), hip.pad_weight(w.to(gpu)), N, epilogue=hip.EPI_GELU, bias=b2.to(gpu))
    assert_close(out, F.gelu(a[:M] @ w.T + b2, approximate="tanh"), "gemm4 gelu")


@pytest.mark.parametrize("M,N,K,ver", [(1500, 2560, 1024, 2), (700, 512, 256, 3), (3000, 256, 192, 2), (1024, 2560, 2560, 2)])
def test_gemm_gate_res_prefetched_epilogue_is_bit_identical(pkg, gpu, M, N, K, ver, lib_option):
    """gemm2 / gemm4's gated-residual epilogue requests every residual chunk of a thread before the first one is used
    (default); SVR2_EPI_PREF=0 loads them chunk by chunk: the same arithmetic, the same bits (ragged last row tile and the
    128-row kernel, which has its own epilogue, included).  [ref: modulation.py AdaSingle 'out' + residual, mmattn.py:269]"""
    hip = pkg.hip
    lib_option("SVR2_GEMM_V", ver)
    a, w = dev_bf16(rnd(M, K, seed=91), gpu), hip.pad_weight(rnd(N, K, seed=92, scale=1 / math.sqrt(K)).to(gpu))
    bias, gate, res = rnd(N, seed=93).to(gpu), rnd(N, seed=94).to(gpu), dev_bf16(rnd(M, N, seed=95), gpu)
    run = lambda: hip.gemm(a, w, N, epilogue=hip.EPI_GATE_RES, bias=bias, gate=gate, resid=res, m=M)
    got = run()
    lib_option("SVR2_EPI_PREF", 0)
    want = run()
    assert torch.equal(got, want)
    ref = res.float().cpu() + gate.cpu() * (a.float().cpu() @ w[:N].float().cpu().T + bias.cpu())
    assert_close(got, ref, f"gate_res {M}x{N}x{K}")


def test_gemm_gather_gate_res(pkg, gpu):
    hip = pkg.hip
    M, N, K, S = 700, 256, 256, 900
    a, w = rnd(S, K, seed=4), rnd(N, K, seed=5, scale=1 / 16)
    bias, gate, res = rnd(N, seed=6), rnd(N, seed=7), rnd(M, N, seed=8)
    rows = torch.randperm(S, generator=torch.Generator().manual_seed(9))[:M].to(torch.int32)
    out = hip.gemm(dev_bf16(a, gpu), hip.pad_weight(w.to(gpu)), N, epilogue=hip.EPI_GATE_RES, bias=bias.to(gpu),
                   gate=gate.to(gpu), resid=dev_bf16(res, gpu), a_rows=rows.to(gpu), m=M)
    ref = res + gate * (a[rows.long()] @ w.T + bias)
    assert_close(out, ref, "gemm gather+gate+res")


def test_gemm_swiglu(pkg, gpu):
    hip = pkg.hip
    M, Hd, K = 333, 192, 128
    a, wg, wi = rnd(M, K, seed=10), rnd(Hd, K, seed=11, scale=0.1), rnd(Hd, K, seed=12, scale=0.1)
    wcat = hip.pad_weight(hip.interleave_swiglu(wg, wi).to(gpu))
    out = hip.gemm(dev_bf16(a, gpu), wcat, 2 * Hd, epilogue=hip.EPI_SWIGLU)
    assert tuple(out.shape) == (M, Hd)
    assert_close(out, F.silu(a @ wg.T) * (a @ wi.T), "gemm swiglu")


@pytest.mark.parametrize("tz,drop", [(1, False), (2, False), (2, True)])
def test_gemm_pixshuf(pkg, gpu, tz, drop):
    """Upsample3D: 1x1x1 conv then 'b (x y z c) f h w -> b c (f z)(h x)(w y)' (+ remove_head)."""
    hip = pkg.hip
    Fr, Hh, Ww, Cc = 3, 6, 10, 64
    r = 4 * tz
    x, w, b = rnd(Fr, Hh, Ww, Cc, seed=13), rnd(Cc * r, Cc, seed=14, scale=1 / 8), rnd(Cc * r, seed=15)
    out = hip.gemm(dev_bf16(x.reshape(-1, Cc), gpu), hip.pad_weight(w.to(gpu)), Cc * r, epilogue=hip.EPI_PIXSHUF3D,
                   bias=b.to(gpu), pixshuf=(Fr, Hh, Ww, Cc, tz, drop))
    y = (x.reshape(-1, Cc) @ w.T + b).reshape(Fr, Hh, Ww, 2, 2, tz, Cc)  # f h w x y z c
    ref = y.permute(0, 5, 1, 3, 2, 4, 6).reshape(Fr * tz, Hh * 2, Ww * 2, Cc)
    if drop:
        ref = torch.cat((ref[:1], ref[2:]), dim=0)
    assert_close(out, ref, f"gemm pixshuf tz={tz} drop={drop}")


# ----------------------------------------------------------------------------------------------- conv3d
def conv_ref(x, w, b, k, s, npad, pad_hw, carry):
    """x [T,H,W,Cin] fp32, w [Cout,kt,kh,kw,Cin]; causal head = carry frames or replicas of frame 0."""
    head = carry if carry is not None else x[:1].expand(npad, -1, -1, -1)
    xp = torch.cat((head, x), dim=0) if npad > 0 else x
    xp = xp.permute(3, 0, 1, 2)[None]  # 1 C T H W
    xp = F.pad(xp, (pad_hw[2], pad_hw[3], pad_hw[0], pad_hw[1]))
    y = F.conv3d(xp, w.permute(0, 4, 1, 2, 3).contiguous(), b, stride=s)
    return y[0].permute(1, 2, 3, 0)


CONV_CASES = [
    # T, H, W, Cin, Cout, k, s, npad, pad_hw, with_carry, resid
    (3, 12, 20, 64, 128, (3, 3, 3), (1, 1, 1), 2, (1, 1, 1, 1), False, False),
    (4, 12, 20, 128, 64, (3, 3, 3), (1, 1, 1), 2, (1, 1, 1, 1), True, True),
    (5, 13, 21, 64, 64, (1, 3, 3), (1, 2, 2), 0, (0, 1, 0, 1), False, False),   # spatial-only downsample, pad right/bottom
    (5, 12, 20, 64, 64, (3, 3, 3), (2, 2, 2), 2, (0, 1, 0, 1), False, False),   # space+time downsample, first chunk
    (4, 12, 20, 64, 64, (3, 3, 3), (2, 2, 2), 1, (0, 1, 0, 1), True, False),    # ... later chunk with 1 carried frame
    (2, 9, 7, 128, 256, (1, 1, 1), (1, 1, 1), 0, (0, 0, 0, 0), False, False),   # 1x1x1 shortcut
    (1, 16, 16, 64, 16, (3, 3, 3), (1, 1, 1), 2, (1, 1, 1, 1), False, False),   # small Cout
    # Ho % 8 == 0 and Wo % 16 == 0: the 2-D patch tiling of the output is active
    (3, 16, 32, 64, 128, (3, 3, 3), (1, 1, 1), 2, (1, 1, 1, 1), True, True),
    (5, 16, 32, 64, 64, (3, 3, 3), (2, 2, 2), 2, (0, 1, 0, 1), False, False),
    (2, 24, 48, 128, 64, (1, 3, 3), (1, 1, 1), 0, (1, 1, 1, 1), False, True),
    (2, 8, 16, 64, 200, (1, 1, 1), (1, 1, 1), 0, (0, 0, 0, 0), False, False),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3d(pkg, gpu, case):
    hip = pkg.hip
    T, H, W, Cin, Cout, k, s, npad, pad_hw, with_carry, with_res = case
    x = rnd(T, H, W, Cin, seed=20)
    w = rnd(Cout, *k, Cin, seed=21, scale=1 / math.sqrt(Cin * k[0] * k[1] * k[2]))
    b = rnd(Cout, seed=22)
    carry = rnd(npad, H, W, Cin, seed=23) if with_carry else None
    ref = conv_ref(x, w, b, k, s, npad, pad_hw, carry)
    res = rnd(*ref.shape, seed=24) if with_res else None
    if with_res:
        ref = ref + res
    out = hip.conv3d(dev_bf16(x, gpu), hip.pad_weight(hip.conv_weight(w.permute(0, 4, 1, 2, 3)).to(gpu)), Cout, k, s, npad=npad,
                     pad_hw=pad_hw, carry=None if carry is None else dev_bf16(carry, gpu), bias=b.to(gpu),
                     resid=None if res is None else dev_bf16(res, gpu))
    assert_close(out, ref, f"conv3d {case}")


def test_conv3d_pointwise_shortcut_on_a_wide_grid(pkg, gpu, lib_option):
    """The 1x1x1 shortcut 256 -> 128 of the decoder's last up block on enough pixels for the 256-row streaming geometry
    (gemm2, >= 384 tiles: where a 4K frame runs it): against the fp32 conv and against the 128-row kernel (SVR2_GEMM_V=1).
    [ref: attn_video_vae.py ResnetBlock3D.conv_shortcut through causal_inflation_lib.py:84-113]"""
    hip = pkg.hip
    T, H, W, Cin, Cout = 2, 224, 225, 256, 128
    x = rnd(T, H, W, Cin, seed=60)
    w = rnd(Cout, 1, 1, 1, Cin, seed=61, scale=1 / 16)
    b = rnd(Cout, seed=62)
    ref = conv_ref(x, w, b, (1, 1, 1), (1, 1, 1), 0, (0, 0, 0, 0), None)
    wd = hip.pad_weight(hip.conv_weight(w.permute(0, 4, 1, 2, 3)).to(gpu))
    run = lambda: hip.conv3d(dev_bf16(x, gpu), wd, Cout, (1, 1, 1), npad=0, pad_hw=(0, 0, 0, 0), bias=b.to(gpu))
    out = run()
    assert_close(out, ref, "pointwise conv 256->128, wide grid")
    lib_option("SVR2_GEMM_V", 1)
    old = run()
    diff = (out.float() - old.float()).abs()
    assert (diff > 0).float().mean() < 0.02 and (diff <= out.float().abs() * 2 ** -7 + 1e-6).all()


HALO_CASES = [
    # T, H, W, Cin, Cout, kt, with_carry, resid
    (2, 16, 16, 64, 256, 3, False, False),
    (3, 20, 37, 128, 256, 3, True, True),      # ragged patches on both edges
    (2, 33, 18, 64, 512, 1, False, True),      # kt = 1 (spatial-only conv), two 256-column blocks
    (1, 7, 5, 192, 256, 3, True, False),       # image smaller than one patch
    (2, 16, 32, 32, 128, 3, False, False),
    (3, 20, 37, 128, 128, 3, True, True),
    (2, 33, 70, 96, 64, 1, False, True),
    (2, 18, 40, 128, 3, 3, True, False),       # conv_out: Cout = 3
    (1, 9, 11, 64, 200, 3, False, False),      # two 128-column blocks, the second ragged
    (2, 40, 72, 64, 384, 3, True, True),       # three 128-column blocks
    (2, 35, 150, 64, 3, 3, False, False),      # 16-channel tiles: ragged 16 x 64 patches
    (3, 16, 64, 128, 16, 3, True, True),
    (1, 20, 70, 32, 8, 1, False, False),
]


@pytest.mark.parametrize("wr", ["0", "3"])
@pytest.mark.parametrize("case", HALO_CASES)
def test_conv3d_halo_resident(pkg, gpu, case, wr, lib_option):
    """conv_halo.hip (halo kept in LDS, nine shifted reads, SVR2_WL_TILED_128x32 weights): both geometries, ragged
    edges, carry, resid; also against the tap-staged kernel on the same inputs."""
    hip = pkg.hip
    lib_option("SVR2_HALO_WR", int(wr))   # W fragments through LDS (0) or straight into registers (1)
    T, H, W, Cin, Cout, kt, with_carry, with_res = case
    k, s, npad, pad_hw = (kt, 3, 3), (1, 1, 1), kt - 1, (1, 1, 1, 1)
    x = rnd(T, H, W, Cin, seed=25)
    w = rnd(Cout, *k, Cin, seed=26, scale=1 / math.sqrt(Cin * 9 * kt))
    b = rnd(Cout, seed=27)
    carry = rnd(npad, H, W, Cin, seed=28) if (with_carry and npad) else None
    ref = conv_ref(x, w, b, k, s, npad, pad_hw, carry)
    res = rnd(*ref.shape, seed=29) if with_res else None
    if with_res:
        ref = ref + res
    w5 = w.permute(0, 4, 1, 2, 3)
    kw = dict(npad=npad, pad_hw=pad_hw, carry=None if carry is None else dev_bf16(carry, gpu), bias=b.to(gpu),
              resid=None if res is None else dev_bf16(res, gpu))
    out = hip.conv3d(dev_bf16(x, gpu), hip.conv_weight_tiled(w5.to(gpu)), Cout, k, s, w_layout=hip.WL_TILED_128x32, **kw)
    assert_close(out, ref, f"conv3d halo {case}")
    if wr:   # the fragment prefetch (default) only moves LDS reads: same bits as the kernel without it
        lib_option("SVR2_CONV_SPLIT", 0)
        a1 = hip.conv3d(dev_bf16(x, gpu), hip.conv_weight_tiled(w5.to(gpu)), Cout, k, s, w_layout=hip.WL_TILED_128x32, **kw)
        lib_option("SVR2_HALO_PREF", 0)
        a0 = hip.conv3d(dev_bf16(x, gpu), hip.conv_weight_tiled(w5.to(gpu)), Cout, k, s, w_layout=hip.WL_TILED_128x32, **kw)
        assert torch.equal(a0, a1)
    if Cin % 64 == 0 and (Cout % 8 == 0 or not with_res):
        out0 = hip.conv3d(dev_bf16(x, gpu), hip.pad_weight(hip.conv_weight(w5).to(gpu)), Cout, k, s, **kw)
        assert_close(out, out0.float().cpu(), f"conv3d halo vs tap-staged {case}")


@pytest.mark.parametrize("case", [
    (1, 64, 64, 512, 512, 3, False, True),     # the latent-resolution convs of a 512 x 512 image: 32 tiles -> 8 splits
    (2, 40, 56, 256, 256, 3, True, True),      # ragged patches, carried head, residual: 24 tiles, 24 slices -> 4 splits
    (1, 72, 80, 128, 128, 3, False, False),    # <32,128> geometry, 12 slices -> 2 splits
    (1, 48, 48, 384, 256, 1, False, True),     # kt = 1 (12 slices)
    (1, 135, 240, 256, 512, 3, True, True),    # 270 tiles: a little over one per CU -> halves pack tighter (2 splits)
])
def test_conv3d_split_k(pkg, gpu, case, lib_option):
    """Split-K path of the halo conv (small grids): against torch, against the unsplit kernel (fp32 summation order is
    the only difference: a few outputs move by one bf16 ulp), GroupNorm partial sums included; and the workspace query."""
    hip = pkg.hip
    T, H, W, Cin, Cout, kt, with_carry, with_res = case
    k, s, npad, pad_hw = (kt, 3, 3), (1, 1, 1), kt - 1, (1, 1, 1, 1)
    x = rnd(T, H, W, Cin, seed=125)
    w = rnd(Cout, *k, Cin, seed=126, scale=1 / math.sqrt(Cin * 9 * kt))
    b = rnd(Cout, seed=127)
    carry = rnd(npad, H, W, Cin, seed=128) if (with_carry and npad) else None
    ref = conv_ref(x, w, b, k, s, npad, pad_hw, carry)
    res = rnd(*ref.shape, seed=129) if with_res else None
    if with_res:
        ref = ref + res
    wt = hip.conv_weight_tiled(w.permute(0, 4, 1, 2, 3).to(gpu))
    kw = dict(npad=npad, pad_hw=pad_hw, carry=None if carry is None else dev_bf16(carry, gpu), bias=b.to(gpu),
              resid=None if res is None else dev_bf16(res, gpu), w_layout=hip.WL_TILED_128x32, gn_stats=True)
    outs = {}
    for split in (0, 1):
        lib_option("SVR2_CONV_SPLIT", split)
        d, _ = hip._conv_desc(dev_bf16(x, gpu), wt, Cout, k, s, npad, pad_hw, None, None, None, None, hip.WL_TILED_128x32, None)
        hip.lib().svr2_conv3d_split_ws_bytes.restype = hip.C.c_int64
        nws = int(hip.lib().svr2_conv3d_split_ws_bytes(hip.C.byref(d)))
        assert (nws > 0) == bool(split), (split, nws)
        out, part = hip.conv3d(dev_bf16(x, gpu), wt, Cout, k, s, **kw)
        torch.cuda.synchronize()
        assert_close(out, ref, f"conv3d split={split} {case}")
        outs[split] = (out.float().cpu(), part.buf.float().cpu())
    a, bb = outs[0][0], outs[1][0]
    diff = (a - bb).abs()
    assert (diff > 0).float().mean() < 0.02, (diff > 0).float().mean()
    # one bf16 ulp of the conv value (rounded before the residual is added) plus one of the rounded sum
    mag = a.abs() + (res.abs() if with_res else 0)
    assert (diff <= (mag + a.abs()) * 2 ** -7 + 1e-6).all()
    pa, pb = outs[0][1], outs[1][1]
    assert pa.shape == pb.shape and torch.allclose(pa, pb, rtol=2e-3, atol=2e-2)


@pytest.mark.parametrize("path", ["tap_staged", "tap_staged_wide", "halo", "halo_lds_w", "cin3", "im2col"])
def test_conv3d_short_carry_equals_materialised_replicas(pkg, gpu, path, lib_option):
    """svr2_conv3d_desc.carry_frames: a carry of ONE frame with npad = 2 (the state after a one-frame first chunk) must give
    the same bits as the materialised [f, f] the reference keeps [ref: causal_inflation_lib.py:266-278, 422-437] - on
    every kernel that reads a carry."""
    hip = pkg.hip
    k, s, npad, pad_hw = (3, 3, 3), (1, 1, 1), 2, (1, 1, 1, 1)
    Cin, Cout = (3, 128) if path in ("cin3", "im2col") else (64, 128)
    T, H, W = (3, 128, 260) if path == "tap_staged_wide" else (3, 20, 37)     # wide: >= 384 tiles of 256 rows -> gemm2
    x, c1 = dev_bf16(rnd(T, H, W, Cin, seed=70), gpu), dev_bf16(rnd(1, H, W, Cin, seed=71), gpu)
    c2 = c1.expand(2, -1, -1, -1).contiguous()
    w5 = rnd(Cout, Cin, *k, seed=72, scale=0.1).to(gpu)
    b = rnd(Cout, seed=73).to(gpu)
    if path == "im2col":
        run = lambda c: hip.im2col3d(x, k, s, npad=npad, pad_hw=pad_hw, carry=c, kpad=128)
    elif path == "cin3":
        run = lambda c: hip.conv3d(x, hip.conv_weight_cin3(w5), Cout, k, s, npad=npad, pad_hw=pad_hw, carry=c, bias=b,
                                   w_layout=hip.WL_CIN3)
    elif path.startswith("halo"):
        if path == "halo_lds_w":
            lib_option("SVR2_HALO_WR", 0)
        run = lambda c: hip.conv3d(x, hip.conv_weight_tiled(w5), Cout, k, s, npad=npad, pad_hw=pad_hw, carry=c, bias=b,
                                   w_layout=hip.WL_TILED_128x32)
    else:
        run = lambda c: hip.conv3d(x, hip.pad_weight(hip.conv_weight(w5)), Cout, k, s, npad=npad, pad_hw=pad_hw, carry=c,
                                   bias=b)
    full, short = run(c2), run(c1)
    assert torch.equal(full, short)
    assert not torch.equal(full, run(None))        # the carry is read at all (frame 0 of x differs from it)


CIN3_CASES = [
    # T, H, W, with_carry
    (1, 16, 32, False),        # exactly one patch
    (3, 37, 50, True),         # ragged on both edges, carried head
    (2, 7, 5, False),          # image smaller than a patch, replicated head
    (5, 48, 96, True),         # several patches per frame, more patches than ... frames in flight
    (4, 130, 70, False),
]


@pytest.mark.parametrize("case", CIN3_CASES)
def test_conv_in_fused_cin3(pkg, gpu, case):
    """conv_in.hip (Cin = 3 -> 128, halo in LDS, operands gathered from it, W in registers): against the fp32 conv of the
    same bf16 inputs, against the im2col + GEMM path it replaces, and its GroupNorm partial sums against the sums of its
    own rounded outputs.  [ref: causal_inflation_lib.py:250-305 through Encoder3D.conv_in]"""
    hip = pkg.hip
    T, H, W, with_carry = case
    k, s, npad, pad_hw, Cout = (3, 3, 3), (1, 1, 1), 2, (1, 1, 1, 1), 128
    x = rnd(T, H, W, 3, seed=40)
    w = rnd(Cout, *k, 3, seed=41, scale=0.2)                       # [Cout, kt, kh, kw, Cin]
    b = rnd(Cout, seed=42)
    carry = rnd(npad, H, W, 3, seed=43) if with_carry else None
    ref = conv_ref(x, w, b, k, s, npad, pad_hw, carry)
    xd, cd = dev_bf16(x, gpu), None if carry is None else dev_bf16(carry, gpu)
    w5 = w.permute(0, 4, 1, 2, 3).contiguous().to(gpu)            # reference layout [Cout, Cin, kt, kh, kw]
    out, part = hip.conv3d(xd, hip.conv_weight_cin3(w5), Cout, k, s, npad=npad, pad_hw=pad_hw, carry=cd, bias=b.to(gpu),
                           w_layout=hip.WL_CIN3, gn_stats=True)
    assert_close(out, ref, f"conv_in fused {case}")
    # the path it replaces: same products, fp32 sums in another order -> at most one bf16 ulp apart, rarely
    col = hip.im2col3d(xd, k, s, npad=npad, pad_hw=pad_hw, carry=cd, kpad=128)
    old = hip.gemm(col, hip.pad_weight(w.reshape(Cout, -1).to(gpu), 128), Cout, bias=b.to(gpu)).view(T, H, W, Cout)
    diff = (out.float() - old.float()).abs()
    assert (diff > 0).float().mean() < 0.02, (diff > 0).float().mean()
    assert (diff <= out.float().abs() * 2 ** -7 + 1e-6).all()
    # GroupNorm partials: per frame and channel, (sum, sum of squares) of the rounded outputs, folded over the patches
    o = out.float()
    got = part.buf.sum(dim=1)                                       # [T, 2, C]
    assert torch.allclose(got[:, 0], o.sum(dim=(1, 2)), rtol=1e-3, atol=1e-2 * math.sqrt(H * W))
    assert torch.allclose(got[:, 1], (o * o).sum(dim=(1, 2)), rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize("Cin,with_carry", [(3, False), (16, True)])
def test_im2col_gemm_small_cin(pkg, gpu, Cin, with_carry):
    hip = pkg.hip
    T, H, W, Cout, k, s, npad, pad_hw = 3, 10, 14, 128, (3, 3, 3), (1, 1, 1), 2, (1, 1, 1, 1)
    x = rnd(T, H, W, Cin, seed=30)
    w = rnd(Cout, *k, Cin, seed=31, scale=0.2)
    b = rnd(Cout, seed=32)
    carry = rnd(npad, H, W, Cin, seed=33) if with_carry else None
    kreal = 27 * Cin
    kpad = (kreal + 63) // 64 * 64
    col = hip.im2col3d(dev_bf16(x, gpu), k, s, npad=npad, pad_hw=pad_hw,
                       carry=None if carry is None else dev_bf16(carry, gpu), kpad=kpad)
    out = hip.gemm(col, hip.pad_weight(w.reshape(Cout, -1).to(gpu), kpad), Cout, bias=b.to(gpu))
    ref = conv_ref(x, w, b, k, s, npad, pad_hw, carry).reshape(-1, Cout)
    assert_close(out, ref, f"im2col+gemm Cin={Cin}")


# ----------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("rows,dim,with_shift", [(77, 2560, True), (5, 256, False), (58, 3072, True), (9, 8192, True)])
def test_rmsnorm_mod(pkg, gpu, rows, dim, with_shift):
    hip = pkg.hip
    x, sc, sh = rnd(rows, dim, seed=40, scale=3.0), rnd(dim, seed=41) + 1, rnd(dim, seed=42)
    out = hip.rmsnorm_mod(dev_bf16(x, gpu), sc.to(gpu), sh.to(gpu) if with_shift else None, 1e-5)
    ref = x / torch.sqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * sc + (sh if with_shift else 0)
    assert_close(out, ref, "rmsnorm_mod")


# ----------------------------------------------------------------------------------------------- attention
def attn_ref(q, k, v, lens):
    out, off = [], 0
    for n in lens:
        qs, ks, vs = (t[off:off + n].permute(1, 0, 2) for t in (q, k, v))
        pr = torch.softmax((qs @ ks.transpose(1, 2)) / math.sqrt(q.shape[-1]), dim=-1)
        out.append((pr @ vs).permute(1, 0, 2))
        off += n
    return torch.cat(out)


@pytest.mark.parametrize("lens,H", [([5, 71, 128, 1], 2), ([463, 868, 135, 64, 129], 3), ([2083], 1)])
def test_window_attn(pkg, gpu, lens, H):
    hip = pkg.hip
    S = sum(lens)
    q, k, v = rnd(S, H, 128, seed=50), rnd(S, H, 128, seed=51), rnd(S, H, 128, seed=52)
    # spike one key against one query so the running max jumps mid-sequence (rescale path)
    if lens[0] >= 463:
        k[300] = q[10] * 3.0
    cu = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    tseq, tq0 = [], []
    for i, n in enumerate(lens):
        for q0 in range(0, n, 128):
            tseq.append(i); tq0.append(q0)
    up = lambda a: torch.tensor(a, dtype=torch.int32, device=gpu)
    out = hip.window_attn(dev_bf16(q, gpu), dev_bf16(k, gpu), dev_bf16(v, gpu), up(cu), up(tseq), up(tq0),
                          1 / math.sqrt(128))
    assert_close(out, attn_ref(q, k, v, lens), f"window_attn {lens}", rtol=2e-2, atol_rel=2e-2)


def test_qkv_prep_is_bit_stable_beside_the_128_row_gemm_on_a_second_stream(pkg, gpu):
    """Regression for profiles/r02_concurrency: with hipcc's packed-f32 forms qkv_prep returned a wrong low half of one
    v_pk_fma_f32 of the rope rotation in 40 % of its calls whenever gemm_kernel<..,128> ran on the same CUs (a second
    stream of the same process is enough).  The build emits no packed-f32 instruction (checked on the shipped library by
    tests/test_cpu_host.py); here the condition itself: 3 000 calls beside the GEMM, every one equal to the first."""
    hip, Wn = pkg.hip, pkg.windows
    H, grid, Lt = 20, (2, 3, 4), 58
    g = torch.Generator().manual_seed(1)
    pl = Wn.plan(grid, Wn.REGULAR, Lt)
    L = grid[0] * grid[1] * grid[2]
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu)
    qkv_v = torch.randn(L, 3 * H * 128, generator=g).to(torch.bfloat16).to(gpu)
    qkv_t = torch.randn(Lt, 3 * H * 128, generator=g).to(torch.bfloat16).to(gpu)
    qn, kn = (1 + 0.1 * torch.randn(128, generator=g)).to(gpu), (1 + 0.1 * torch.randn(128, generator=g)).to(gpu)
    fr = 1.0 / (10000.0 ** (torch.arange(0, 42, 2)[:21].float() / 42))
    ang = torch.arange(1024, dtype=torch.float32)[:, None] * fr[None]
    rc, rs = ang.cos().contiguous().to(gpu), ang.sin().contiguous().to(gpu)
    src, pos = up(pl.src_row), up(pl.pos)
    ga = torch.randn(24, 2560, device=gpu).to(torch.bfloat16)
    gw = hip.pad_weight((torch.randn(7680, 2560, device=gpu) * 0.02).to(torch.bfloat16))
    gout = torch.empty(24, 7680, dtype=torch.bfloat16, device=gpu)
    side = torch.cuda.Stream()
    for mode, c, s_ in ((1, rc, rs), (2, torch.rand(pl.total, 30, generator=g).to(gpu), torch.rand(pl.total, 30, generator=g).to(gpu))):
        call = lambda: hip.qkv_prep(qkv_v, qkv_t, src, pos, qn, kn, qn, kn, c, s_, H, mode, 1e-5)
        ref = call().clone()
        torch.cuda.synchronize()
        bad = 0
        for i in range(1500):
            with torch.cuda.stream(side):
                for _ in range(4):
                    hip.gemm(ga, gw, 7680, gout)
            bad += int(not torch.equal(call(), ref))
        torch.cuda.synchronize()
        assert bad == 0, f"rope_mode {mode}: {bad} of 1500 qkv_prep calls differ beside the 128-row GEMM"


def test_window_attn_default_kernel_bit_identical_to_plain_kernel(pkg, gpu, lib_option):
    """flash_attn2_kernel (default) and the plain flash_attn_kernel of the same tile geometry (SVR2_ATTN_V=1) share the
    summation order: equal bits on ragged sequences incl. the rescale spike - the only reason the switch still exists."""
    hip = pkg.hip
    lens, H = [463, 868, 135, 64, 129, 1, 810 + 58], 3
    S = sum(lens)
    q, k, v = rnd(S, H, 128, seed=53), rnd(S, H, 128, seed=54), rnd(S, H, 128, seed=55)
    k[300] = q[10] * 3.0
    cu = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    tseq, tq0 = [], []
    for i, n in enumerate(lens):
        for q0 in range(0, n, 128):
            tseq.append(i); tq0.append(q0)
    up = lambda a: torch.tensor(a, dtype=torch.int32, device=gpu)
    args = (dev_bf16(q, gpu), dev_bf16(k, gpu), dev_bf16(v, gpu), up(cu), up(tseq), up(tq0), 1 / math.sqrt(128))
    out_default = hip.window_attn(*args)
    lib_option("SVR2_ATTN_V", 1)
    out_plain = hip.window_attn(*args)
    assert torch.equal(out_default, out_plain)


def test_qkv_prep_and_txt_mean(pkg, gpu):
    """Window gather + per-head RMSNorm + 3B mm-rope vs the oracle's pieces."""
    from seedvr2_ref import dit as odit, rope as orope, windows as owin
    hip, W = pkg.hip, pkg.windows
    H, D, Lt = 2, 128, 58
    grid = (3, 10, 12)
    L = grid[0] * grid[1] * grid[2]
    qkv_v, qkv_t = rnd(L, 3 * H * D, seed=60), rnd(Lt, 3 * H * D, seed=61)
    nq, nk, nqt, nkt = (rnd(D, seed=62 + i) * 0.1 + 1 for i in range(4))
    for method in (W.REGULAR, W.SHIFTED):
        pl = W.plan(grid, method, Lt)
        fr = orope.lang_freqs(D // 3)
        ang = torch.arange(256, dtype=torch.float32)[:, None] * fr[None]
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu)
        out = hip.qkv_prep(dev_bf16(qkv_v, gpu), dev_bf16(qkv_t, gpu), up(pl.src_row), up(pl.pos), nq.to(gpu),
                           nk.to(gpu), nqt.to(gpu), nkt.to(gpu), ang.cos().contiguous().to(gpu),
                           ang.sin().contiguous().to(gpu), H, 1, 1e-5).float().cpu()
        # oracle: per window
        boxes = owin.windows_for(method, grid)
        idx, lens, shapes = owin.window_token_index(grid, boxes)
        qv, kv, vv = (qkv_v.reshape(L, 3, H, D)[:, i] for i in range(3))
        qt, kt, vt = (qkv_t.reshape(Lt, 3, H, D)[:, i] for i in range(3))
        qv, kv = odit.rms_norm(qv, 1e-5, nq), odit.rms_norm(kv, 1e-5, nk)
        qt, kt = odit.rms_norm(qt, 1e-5, nqt), odit.rms_norm(kt, 1e-5, nkt)
        ref_q, ref_k, ref_v, off = [], [], [], 0
        for n, shp in zip(lens.tolist(), shapes.tolist()):
            sel = torch.from_numpy(idx[off:off + n]); off += n
            av, at = orope.mm_window_angles_3b(shp, Lt, D)
            ref_q += [orope.rotate_interleaved(qv[sel], av[:, None]), orope.rotate_interleaved(qt, at[:, None])]
            ref_k += [orope.rotate_interleaved(kv[sel], av[:, None]), orope.rotate_interleaved(kt, at[:, None])]
            ref_v += [vv[sel], vt]
        assert_close(out[0], torch.cat(ref_q), f"qkv_prep q {method}")
        assert_close(out[1], torch.cat(ref_k), f"qkv_prep k {method}")
        assert_close(out[2], torch.cat(ref_v), f"qkv_prep v {method}", rtol=0, atol_rel=0)
        # text mean over windows
        att = rnd(pl.total, 256, seed=70)
        tm = hip.txt_mean(dev_bf16(att, gpu), up(pl.txt_row0), Lt)
        ref_tm = torch.stack([att[r0:r0 + Lt] for r0 in pl.txt_row0.tolist()], dim=1).mean(1)
        assert_close(tm, ref_tm, "txt_mean")


def test_patch_in_out(pkg, gpu):
    from seedvr2_ref import dit as odit
    hip = pkg.hip
    T, H, W, Cc = 2, 6, 8, 16
    noise, latent = rnd(T, H, W, Cc, seed=80), rnd(T, H, W, Cc, seed=81)
    rows = hip.patch_in(dev_bf16(noise, gpu), dev_bf16(latent, gpu), 192).float().cpu()
    vid = torch.cat((noise, latent, torch.ones(T, H, W, 1)), dim=-1).reshape(-1, 33)
    ref = odit.patch_in(vid, (T, H, W), (1, 2, 2))
    assert torch.equal(rows[:, :132], ref) and rows[:, 132:].abs().max() == 0
    pred = rnd(T * (H // 2) * (W // 2), 64, seed=82)
    x0 = hip.patch_out_x0(dev_bf16(pred, gpu), dev_bf16(noise, gpu))
    ref0 = noise.reshape(-1, Cc) - odit.patch_out(pred, (T, H // 2, W // 2), (1, 2, 2), Cc)
    assert_close(x0.reshape(-1, Cc), ref0, "patch_out_x0", rtol=8e-3, atol_rel=8e-3)


# ----------------------------------------------------------------------------------------------- DiT
@pytest.mark.parametrize("grid_thw", [(2, 16, 24), (1, 32, 32), (3, 20, 36)])
def test_dit_one_step_vs_oracle(pkg, gpu, grid_thw):
    """Whole NaDiT one-step (3 layers: separate-weights layer, shared layer, last layer) vs the fp32 oracle.
    Tolerance: the bf16 oracle itself deviates from the fp32 oracle by err_bf16; require err_hip <= 2*err_bf16."""
    from seedvr2_ref import dit as odit
    kw = dict(vid_dim=256, heads=2, head_dim=128, num_layers=3, mm_layers=1, txt_in_dim=64, mlp_multiple_of=64)
    ocfg = odit.DiTConfig(**kw)
    p = {k: bf(v) for k, v in odit.synthetic_params(ocfg, 3).items()}
    model = pkg.dit.NaDiTHip(pkg.dit.DiTConfig(**kw), p, gpu)
    T, H, W = grid_thw
    noise, latent, txt = rnd(T, H, W, 16, seed=90), rnd(T, H, W, 16, seed=91), rnd(58, 64, seed=92)
    got = model.one_step(dev_bf16(noise, gpu), dev_bf16(latent, gpu), txt.to(gpu)).float().cpu()
    ref32 = odit.one_step_sr(p, ocfg, latent, noise, txt, torch.float32)
    ref16 = odit.one_step_sr(p, ocfg, latent, noise, txt, torch.bfloat16).float()
    rms = ref32.pow(2).mean().sqrt()
    e_hip = (got - ref32).pow(2).mean().sqrt() / rms
    e_b16 = (ref16 - ref32).pow(2).mean().sqrt() / rms
    print(f"DiT {grid_thw}: rel-rms err hip={e_hip:.4e} bf16-oracle={e_b16:.4e}")
    assert torch.isfinite(got).all()
    assert e_hip <= 2.0 * e_b16 + 1e-3, (e_hip, e_b16)
    assert_close(got, ref32, "dit one_step", rtol=5e-2, atol_rel=5e-2)


def test_gemm_gelu(pkg, gpu):
    hip = pkg.hip
    M, N, K = 200, 384, 128
    a, w, b = rnd(M, K, seed=95), rnd(N, K, seed=96, scale=0.2), rnd(N, seed=97)
    out = hip.gemm(dev_bf16(a, gpu), hip.pad_weight(w.to(gpu)), N, epilogue=hip.EPI_GELU, bias=b.to(gpu))
    assert_close(out, F.gelu(a @ w.T + b, approximate="tanh"), "gemm gelu")


def test_dit_graph_replay_equals_eager(pkg, gpu, monkeypatch):
    """Small grids replay a captured hipGraph of the forward pass: same bits as the eager launches, for fresh inputs too."""
    import time
    from seedvr2_ref import dit as odit
    kw = dict(vid_dim=256, heads=2, head_dim=128, num_layers=4, mm_layers=2, txt_in_dim=64, mlp_multiple_of=64)
    dcfg = odit.DiTConfig(**kw)
    dp = {k: v.to(torch.bfloat16).float() for k, v in odit.synthetic_params(dcfg, 5).items()}
    model = pkg.dit.NaDiTHip(pkg.dit.DiTConfig(**kw), dp, gpu)
    g = torch.Generator().manual_seed(6)
    txt = torch.randn(58, 64, generator=g).to(torch.bfloat16).to(gpu)
    outs = {}
    for mode in ("0", "1"):
        model.graph_mode = mode
        res = []
        for seed in (1, 2, 3):
            gg = torch.Generator().manual_seed(seed)
            noise = torch.randn(3, 8, 12, 16, generator=gg).to(torch.bfloat16).to(gpu)
            lat = torch.randn(3, 8, 12, 16, generator=gg).to(torch.bfloat16).to(gpu)
            res.append(model.one_step(noise, lat, txt).clone())
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(10):
            model.one_step(noise, lat, txt)
        torch.cuda.synchronize()
        outs[mode] = (res, (time.time() - t0) / 10)
    for a, b in zip(outs["0"][0], outs["1"][0]):
        assert torch.equal(a, b)
    assert not torch.equal(outs["1"][0][0], outs["1"][0][1])
    print(f"DiT forward (4 layers, 72 tokens): eager {outs['0'][1]*1e3:.2f} ms, graph replay {outs['1'][1]*1e3:.2f} ms")
    assert len(model._graphs) == 1


@pytest.mark.parametrize("flavour", ["3b", "7b"])
def test_dit_txt_lane_equals_single_stream(pkg, gpu, monkeypatch, flavour):
    """The text rows' launches on a side stream (dit.block txt_lane): same bits as the single-stream order, eager and
    under graph capture, for fresh inputs on every call (a lane that read a buffer too early would show here)."""
    from seedvr2_ref import dit as odit
    kw = dict(vid_dim=256, heads=2, head_dim=128, num_layers=5, mm_layers=2, txt_in_dim=64, mlp_multiple_of=64)
    if flavour == "7b":
        kw.update(mm_layers=5, mlp_type="normal", out_norm=False, last_vid_only=False)
    dcfg = odit.DiTConfig(**kw)
    dp = {k: v.to(torch.bfloat16).float() for k, v in odit.synthetic_params(dcfg, 8).items()}
    model = pkg.dit.NaDiTHip(pkg.dit.DiTConfig(rope_mode=1 if flavour == "3b" else 2, **kw), dp, gpu)
    g = torch.Generator().manual_seed(16)
    txt = torch.randn(58, 64, generator=g).to(torch.bfloat16).to(gpu)
    outs = {}
    for lane in ("0", "1"):
        for graph in ("0", "1"):
            model.txt_lane_mode, model.graph_mode = lane, graph
            if hasattr(model, "_graphs"):
                model._graphs.clear()
            res = []
            for seed in (1, 2, 3, 4):
                gg = torch.Generator().manual_seed(seed)
                noise = torch.randn(5, 16, 24, 16, generator=gg).to(torch.bfloat16).to(gpu)
                lat = torch.randn(5, 16, 24, 16, generator=gg).to(torch.bfloat16).to(gpu)
                res.append(model.one_step(noise, lat, txt).clone())
            torch.cuda.synchronize()
            outs[(lane, graph)] = res
    ref = outs[("0", "0")]
    assert not torch.equal(ref[0], ref[1])
    for key, res in outs.items():
        for a, b in zip(ref, res):
            assert torch.equal(a, b), key


@pytest.mark.parametrize("grid_thw", [(2, 16, 24), (3, 20, 36)])
def test_dit_7b_variant_vs_oracle(pkg, gpu, grid_thw):
    """NaDiT-7B deltas: GELU MLP with biases, separate vid/txt weights in every layer, video-only 'pixel'
    rope with bf16-quantised angles, no output norm / vid-only last layer."""
    from seedvr2_ref import dit as odit
    kw = dict(vid_dim=256, heads=2, head_dim=128, num_layers=2, mm_layers=2, txt_in_dim=64, mlp_type="normal",
              out_norm=False, last_vid_only=False)
    ocfg = odit.DiTConfig(**kw)
    p = {k: bf(v) for k, v in odit.synthetic_params(ocfg, 5).items()}
    model = pkg.dit.NaDiTHip(pkg.dit.DiTConfig(rope_mode=2, **kw), p, gpu)
    T, H, W = grid_thw
    noise, latent, txt = rnd(T, H, W, 16, seed=90), rnd(T, H, W, 16, seed=91), rnd(58, 64, seed=92)
    got = model.one_step(dev_bf16(noise, gpu), dev_bf16(latent, gpu), txt.to(gpu)).float().cpu()
    # the reference quantises the rope angles to the activations' dtype: compare against the bf16-angle oracle
    ref16 = odit.one_step_sr(p, ocfg, latent, noise, txt, torch.bfloat16, rope_mode="7b").float()
    ref32 = odit.one_step_sr(p, ocfg, latent, noise, txt, torch.float32, rope_mode="7b")
    rms = ref16.pow(2).mean().sqrt()
    e_hip = (got - ref16).pow(2).mean().sqrt() / rms
    e_32 = (ref32 - ref16).pow(2).mean().sqrt() / rms
    print(f"DiT-7B {grid_thw}: rel-rms hip-vs-bf16oracle={e_hip:.3e}  fp32oracle-vs-bf16oracle={e_32:.3e}")
    assert torch.isfinite(got).all()
    assert e_hip < 3e-2
