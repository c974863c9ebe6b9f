"""Full-size (BASELINE configs[2]: 720p->4K, batch of 21 frames) checks through size-independent properties,
bit-exact where the arithmetic allows: the oracle cannot run these shapes in seconds, but one-hot / constant
inputs make the expected result known in closed form, so every addressing path of the kernels (tile order,
XCD remap, 2-D patch tiles, causal head, window gather, text replication) is exercised at real sizes."""

import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

L4K, D3B = 194400, 2560          # tokens of a 4K batch of 21 frames (T_l = 6), NaDiT-3B width


def test_gemm_onehot_rows_fullsize(pkg, gpu):
    """A = one-hot rows -> C[m, :] = W[:, idx[m]] + bias exactly (bf16 values pass through fp32 accumulation
    unchanged), at the qkv GEMM's full shape M=194400, N=7680, K=2560, with the fused row gather."""
    hip = pkg.hip
    g = torch.Generator(device=gpu).manual_seed(0)
    M, N, K = L4K, 7680, D3B
    w = torch.randn(N, K, generator=g, device=gpu).to(torch.bfloat16)
    idx = torch.randint(0, K, (M,), generator=g, device=gpu)
    a = torch.zeros(M, K, dtype=torch.bfloat16, device=gpu)
    a[torch.arange(M, device=gpu), idx] = 1.0
    perm = torch.randperm(M, generator=g, device=gpu).to(torch.int32)
    out = hip.gemm(a, hip.pad_weight(w), N, a_rows=perm, m=M)
    rows = torch.randint(0, M, (4096,), generator=g, device=gpu)
    rows[:3] = torch.tensor([0, M - 1, M // 2], device=gpu)
    exp = w[:, idx[perm[rows].long()]].T
    assert torch.equal(out[rows], exp)
    assert torch.isfinite(out.float()).all()


@pytest.mark.parametrize("H,W,Cin,Cout,tiled", [
    (2160, 3840, 128, 128, False),     # tap-staged kernels (row weight layout)
    (2160, 3840, 128, 128, True),      # conv_halo_wr_kernel<32,128>: 512-pixel x 128-channel blocks
    (1080, 1920, 256, 256, True),      # conv_halo_wr_kernel<16,256>; 1080 = 67.5 patches: ragged last patch row
    (2160, 3840, 128, 3, True),        # conv_halo_kernel<64,16>: conv_out
])
def test_conv3d_delta_input_fullsize(pkg, gpu, H, W, Cin, Cout, tiled):
    """One-hot input pixels in a 4-frame chunk with a carried head: the output is the (flipped) filter placed around
    each pixel plus bias, everywhere else bias only - bit exact.  Pixels sit at the image border, at patch corners and
    in the carry, so zero padding, the 2-D patch tiles / halo loads and the causal head are all hit at full size; the
    GroupNorm partial sums of the halo kernels must fold to the sums of the stored output."""
    hip = pkg.hip
    T = 4
    g = torch.Generator(device=gpu).manual_seed(1)
    wt = torch.randn(Cout, 3, 3, 3, Cin, generator=g, device=gpu).to(torch.bfloat16)      # [Cout,kt,kh,kw,Cin]
    bias = torch.randn(Cout, generator=g, device=gpu)
    x = torch.zeros(T, H, W, Cin, dtype=torch.bfloat16, device=gpu)
    carry = torch.zeros(2, H, W, Cin, dtype=torch.bfloat16, device=gpu)
    hot = [("x", 1, 0, 0, 5), ("x", 3, H - 1, W - 1, Cin - 51), ("x", 2, H // 2 - 1, W // 2 - 1, 0), ("x", 0, 8, 16, Cin - 1),
           ("x", 2, 15, 31, 40), ("x", 3, 48, 96, 41), ("c", 1, 500, 700, 31)]
    for src, t, y, xx, c in hot:
        (x if src == "x" else carry)[t, y, xx, c] = 1.0
    w5 = wt.permute(0, 4, 1, 2, 3)
    stats = tiled and Cout % 8 == 0
    if tiled:
        res = hip.conv3d(x, hip.conv_weight_tiled(w5), Cout, (3, 3, 3), npad=2, carry=carry, bias=bias,
                         w_layout=hip.WL_TILED_128x32, gn_stats=stats)
        out, part = res if stats else (res, None)
    else:
        out = hip.conv3d(x, hip.pad_weight(hip.conv_weight(w5)), Cout, (3, 3, 3), npad=2, carry=carry, bias=bias)
    b16 = bias.to(torch.bfloat16)
    # padded frame index p: carry frames 0,1 ; chunk frame t -> p = t + 2 ; output t' reads p in [t', t'+2]
    touched = torch.zeros(T, H, W, dtype=torch.bool, device=gpu)
    for src, t, y, xx, c in hot:
        p = t + 2 if src == "x" else t
        for tp in range(max(p - 2, 0), min(p, T - 1) + 1):
            kt = p - tp
            for ky in range(3):
                for kx in range(3):
                    oy, ox = y - (ky - 1), xx - (kx - 1)
                    if 0 <= oy < H and 0 <= ox < W:
                        exp = (wt[:, kt, ky, kx, c].float() + bias).to(torch.bfloat16)
                        assert torch.equal(out[tp, oy, ox], exp), (src, t, y, xx, tp, oy, ox)
                        touched[tp, oy, ox] = True
    rest = out[~touched]
    assert torch.equal(rest, b16.expand_as(rest))
    if stats:
        of = out.float()
        sums = part.buf.double().sum(dim=1)
        want = torch.stack((of.sum(dim=(1, 2)), (of * of).sum(dim=(1, 2))), dim=1).double()
        assert ((sums - want).abs() <= 1e-4 * want.abs() + 1e-2).all()


def test_window_attention_softmax_rows_fullsize(pkg, gpu):
    """4K window set (regular + shifted), 20 heads: with V constant per (head, dim) inside every window the
    output must equal that constant (softmax rows sum to one) for every query, whatever q and k are; the text
    rows replicated into every window must come back identical from txt_mean."""
    hip, Wn = pkg.hip, pkg.windows
    H, D, Lt = 20, 128, 58
    grid = (6, 135, 240)
    g = torch.Generator(device=gpu).manual_seed(2)
    for method in (Wn.REGULAR, Wn.SHIFTED):
        pl = Wn.plan(grid, method, Lt)
        S = pl.total
        up = lambda arr: torch.from_numpy(np.ascontiguousarray(arr)).to(gpu)
        q = torch.randn(S, H, D, generator=g, device=gpu).to(torch.bfloat16)
        k = torch.randn(S, H, D, generator=g, device=gpu).to(torch.bfloat16)
        const = torch.randn(H, D, generator=g, device=gpu).to(torch.bfloat16)
        v = const.expand(S, H, D).contiguous()
        out = hip.window_attn(q, k, v, up(pl.cu_seqlens), up(pl.tile_seq), up(pl.tile_q0), 1 / math.sqrt(D))
        err = (out.float() - const.float()).abs().max().item()
        assert err <= 2 ** -6 * const.float().abs().max().item() + 1e-3, err   # P is bf16: rows sum to 1 +- 2^-8
        tm = hip.txt_mean(out.view(S, H * D), up(pl.txt_row0), Lt)
        assert (tm.float() - const.float().view(1, -1)).abs().max().item() <= 2 ** -6 * 4 + 1e-3
        # window partition covers every token exactly once and the fused gather inverts it
        assert (pl.src_row[pl.inv_rows] == np.arange(grid[0] * grid[1] * grid[2])).all()


def test_groupnorm_constant_and_scale_fullsize(pkg, gpu):
    """Per-frame GroupNorm at 4K (4 frames, 128 channels): normalised output has zero mean / unit variance per
    (frame, group) and is invariant to an input rescale (fp64-folded statistics)."""
    hip = pkg.hip
    g = torch.Generator(device=gpu).manual_seed(3)
    x = (torch.randn(2, 2160, 3840, 128, generator=g, device=gpu) * 3 + 1.5).to(torch.bfloat16)
    ones, zeros = torch.ones(128, device=gpu), torch.zeros(128, device=gpu)
    y = hip.groupnorm_silu(x, ones, zeros, 32, 1e-6, silu=False).float().view(2, -1, 32, 4)
    m, s = y.mean(dim=(1, 3)), y.var(dim=(1, 3), unbiased=False)
    assert m.abs().max() < 5e-3 and (s - 1).abs().max() < 1e-2
    y2 = hip.groupnorm_silu((x.float() * 4).to(torch.bfloat16), ones, zeros, 32, 1e-6, silu=False).float()
    assert (y2.view_as(y) - y).abs().max() < 2e-2
