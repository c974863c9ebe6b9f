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


# ----------------------------------------------------------------------------- cfg4 (NaDiT-7B at 4K) shapes
D7B, H7B, HID7B = 3072, 24, 12288      # [ref: configs_7b/main.yaml; src/models/dit_7b/mlp.py:27-43]


def _ulp_close(got, exp, what):
    """bf16 results that were rounded once from nearly the same fp32 value: at most one bf16 ulp apart."""
    d = (got.float() - exp.float()).abs()
    bound = exp.float().abs() * 2.0 ** -7 + 2.0 ** -10
    i = int((d - bound).argmax())
    assert (d <= bound).all(), f"{what}: worst |d| {float(d.flatten()[i]):.4g} at |exp| {float(exp.float().abs().flatten()[i]):.4g}"


def test_gemm_onehot_7b_mlp_in_gelu_fullsize(pkg, gpu):
    """proj_in of the 7B MLP at 4K: M = 194 400, K = 3072 -> N = 12 288 with the fused GELU(tanh): an output of 2.39 G
    elements (past 2^31).  One-hot A rows make the accumulator exactly W[n, idx[m]]; like the reference's op sequence
    (the Linear's bf16 output, then GELU on that bf16 tensor: two roundings) the kernel rounds acc + bias to bf16 before the
    activation, so the expected value is bf16(gelu_tanh(bf16(W[:, idx] + b))), within one bf16 ulp (tanh arithmetic);
    checked on 4 096 rows incl. the first and the very last one."""
    hip = pkg.hip
    g = torch.Generator(device=gpu).manual_seed(10)
    M, N, K = L4K, HID7B, D7B
    assert M * N > 2 ** 31
    w = torch.randn(N, K, generator=g, device=gpu).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device=gpu)
    idx = torch.randint(0, K, (M,), generator=g, device=gpu)
    a = torch.zeros(M, K, dtype=torch.bfloat16, device=gpu)
    a[torch.arange(M, device=gpu), idx] = 1.0
    out = hip.gemm(a, hip.pad_weight(w), N, epilogue=hip.EPI_GELU, bias=bias)
    assert tuple(out.shape) == (M, N)
    rows = torch.randint(0, M, (4096,), generator=g, device=gpu)
    rows[:3] = torch.tensor([0, M - 1, M // 2], device=gpu)
    lin = (w[:, idx[rows]].T.float() + bias).to(torch.bfloat16).float()
    exp = torch.nn.functional.gelu(lin, approximate="tanh").to(torch.bfloat16)
    _ulp_close(out[rows].flatten(), exp.flatten(), "7B proj_in + GELU at M = 194 400")
    # every row block was written (no tile skipped past the 2^31-element mark): row sums of a strided sample are finite and non-trivial
    s = out[::997].float().abs().sum(dim=1)
    assert torch.isfinite(s).all() and (s > 0).all()


def test_gemm_onehot_7b_mlp_out_gate_res_fullsize(pkg, gpu):
    """proj_out of the 7B MLP at 4K: K = 12 288 -> N = 3072 with `resid + gate * (acc + bias)` (AdaSingle 'out' + residual):
    the longest K loop of the path (192 K-steps).  One-hot rows: acc = W[n, idx[m]] exactly; the Linear's output is rounded
    to bf16 (like the reference's), gate and residual are applied to that in fp32 and rounded once."""
    hip = pkg.hip
    g = torch.Generator(device=gpu).manual_seed(11)
    M, N, K = L4K, D7B, HID7B
    w = torch.randn(N, K, generator=g, device=gpu).to(torch.bfloat16)
    bias, gate = torch.randn(N, generator=g, device=gpu), torch.randn(N, generator=g, device=gpu)
    resid = torch.randn(M, N, generator=g, device=gpu).to(torch.bfloat16)
    idx = torch.randint(0, K, (M,), generator=g, device=gpu)
    a = torch.zeros(M, K, dtype=torch.bfloat16, device=gpu)
    a[torch.arange(M, device=gpu), idx] = 1.0
    out = hip.gemm(a, hip.pad_weight(w), N, epilogue=hip.EPI_GATE_RES, bias=bias, gate=gate, resid=resid)
    rows = torch.randint(0, M, (4096,), generator=g, device=gpu)
    rows[:3] = torch.tensor([0, M - 1, M // 2], device=gpu)
    lin = (w[:, idx[rows]].T.float() + bias).to(torch.bfloat16).float()
    exp = (resid[rows].float() + gate * lin).to(torch.bfloat16)
    # fma or mul + add before the single final rounding: at most one bf16 ulp of the result
    _ulp_close(out[rows].flatten(), exp.flatten(), "7B proj_out + gate + residual at M = 194 400")


def test_window_attention_softmax_rows_7b_heads_fullsize(pkg, gpu):
    """The 4K window set with the 7B's 24 heads (softmax rows sum to one: constant V comes back)."""
    hip, Wn = pkg.hip, pkg.windows
    H, D, Lt = H7B, 128, 58
    grid = (6, 135, 240)
    g = torch.Generator(device=gpu).manual_seed(12)
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
        assert err <= 2 ** -6 * const.float().abs().max().item() + 1e-3, (method, err)
        del q, k, v, out


def test_qkv_prep_7b_rope_fullsize(pkg, gpu):
    """qkv_prep with 24 heads and the 7B 'pixel' rope (per-position tables, video rows only) on the 4K window set
    [ref: src/models/dit_7b/rope.py:73-124]: against the same call with identity tables (cos = 1, sin = 0), every rotated
    pair keeps its norm, the dims beyond the 3 x 10 rotary pairs and all text rows are bit-identical, v is the plain
    window gather of the input."""
    hip, Wn = pkg.hip, pkg.windows
    H, D, Lt = H7B, 128, 58
    grid = (6, 135, 240)
    L = grid[0] * grid[1] * grid[2]
    g = torch.Generator(device=gpu).manual_seed(13)
    cfg = pkg.dit.CFG_7B
    model = object.__new__(pkg.dit.NaDiTHip)            # only the table builder is used: no weights
    model.cfg, model.device, model._rope7 = cfg, gpu, {}
    qkv_v = torch.randn(L, 3 * H * D, generator=g, device=gpu).to(torch.bfloat16)
    qkv_t = torch.randn(Lt, 3 * H * D, generator=g, device=gpu).to(torch.bfloat16)
    ones = torch.ones(D, device=gpu)
    up = lambda arr: torch.from_numpy(np.ascontiguousarray(arr)).to(gpu)
    for method in (Wn.REGULAR, Wn.SHIFTED):
        pl = Wn.plan(grid, method, Lt)
        S = pl.total
        rc, rs = model._rope_tables_7b(grid, method, Lt)
        assert tuple(rc.shape) == (S, 30)
        src, pos = up(pl.src_row), up(pl.pos)
        rot = hip.qkv_prep(qkv_v, qkv_t, src, pos, ones, ones, ones, ones, rc, rs, H, 2, 1e-5)
        idn = hip.qkv_prep(qkv_v, qkv_t, src, pos, ones, ones, ones, ones, torch.ones_like(rc), torch.zeros_like(rs), H, 2, 1e-5)
        is_txt = src < 0
        assert int(is_txt.sum()) == pl.nwin * Lt
        assert torch.equal(rot[2], idn[2])                                  # v: no rope
        vid_rows = (~is_txt).nonzero().flatten()
        assert torch.equal(rot[2][vid_rows[:4096]].reshape(4096, -1), qkv_v[src[vid_rows[:4096]].long(), 2 * H * D:])
        for o in (0, 1):
            assert torch.equal(rot[o][is_txt], idn[o][is_txt])              # text rows are not rotated
            assert torch.equal(rot[o][:, :, 60:], idn[o][:, :, 60:])        # 3 axes x 10 freqs x 2 = 60 rotary dims
            sel = vid_rows[torch.randint(0, vid_rows.numel(), (20000,), generator=g, device=gpu)]
            sel[:2] = torch.stack((vid_rows[0], vid_rows[-1]))
            a = rot[o][sel][:, :, :60].float().reshape(-1, H, 30, 2)
            b = idn[o][sel][:, :, :60].float().reshape(-1, H, 30, 2)
            na, nb = a.pow(2).sum(-1).sqrt(), b.pow(2).sum(-1).sqrt()
            assert ((na - nb).abs() <= 2.0 ** -6 * nb + 1e-3).all()          # both sides carry one bf16 rounding per element
            assert (a - b).abs().max() > 0.1                                 # ... and the rotation is not the identity
        del rot, idn
