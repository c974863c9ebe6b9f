"""CPU tests (no GPU): oracle vs the committed golden vectors (generated from the importable reference
modules by oracle/pin_against_reference.py), product host logic vs the same vectors, and the C-ABI
library's symbol table.  No compute call into the library happens here."""

import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch
from safetensors.torch import load_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(GOLD, "dit_pins.json")) as fh:
        meta = json.load(fh)
    return meta, load_file(os.path.join(GOLD, "dit_pins.safetensors"))


def _key_grid(key):
    g, meth = key.split("|")
    return tuple(int(v) for v in g.split("x")), meth


def test_oracle_windows_match_reference_golden(gold):
    from seedvr2_ref import windows as owin
    meta, _ = gold
    assert len(meta["G1_windows"]) >= 20
    for key, boxes in meta["G1_windows"].items():
        grid, meth = _key_grid(key)
        assert [list(b) for b in owin.windows_for(meth, grid)] == boxes, key


def test_product_windows_match_reference_golden(pkg, gold):
    meta, _ = gold
    for key, boxes in meta["G1_windows"].items():
        grid, meth = _key_grid(key)
        assert [list(b) for b in pkg.windows.window_boxes(meth, grid)] == boxes, key


def test_product_window_plan_tables(pkg, gold):
    """src_row / inv_rows / cu_seqlens / tiles of the product plan vs reference-derived index goldens."""
    meta, _ = gold
    W = pkg.windows
    for key, g in meta["G2_index"].items():
        grid, meth = _key_grid(key)
        Lt = 58
        p = W.plan(grid, meth, Lt)
        vid = p.src_row[p.src_row >= 0]
        assert int(vid.astype(np.int64).sum()) == g["idx_sum"]
        lens = g["lens"]
        assert p.nwin == len(lens) and p.total == sum(lens) + Lt * len(lens)
        assert p.cu_seqlens.tolist() == np.concatenate(([0], np.cumsum(np.asarray(lens) + Lt))).tolist()
        # first window's video rows are the reference's window_idx head
        n0 = lens[0]
        assert p.src_row[:min(n0, 64)].tolist() == g["idx_head"][:min(n0, 64)]
        # text rows follow every window; inverse permutation round-trips
        assert p.src_row[n0:n0 + Lt].tolist() == [-1 - j for j in range(Lt)]
        L = grid[0] * grid[1] * grid[2]
        assert (p.src_row[p.inv_rows] == np.arange(L)).all()
        assert p.txt_row0.tolist() == [int(c) - Lt for c in p.cu_seqlens[1:]]
        # rope coordinates: video time offset by Lt, window-local space; text (j,j,j)
        assert p.pos[n0:n0 + Lt].tolist() == [[j, j, j] for j in range(Lt)]
        assert p.pos[0].tolist() == [Lt, 0, 0]
        # every 128-query tile is inside its sequence
        for s, q0 in zip(p.tile_seq.tolist(), p.tile_q0.tolist()):
            assert 0 <= q0 < p.cu_seqlens[s + 1] - p.cu_seqlens[s] and q0 % W.QTILE == 0
        ntiles = sum((n + Lt + W.QTILE - 1) // W.QTILE for n in lens)
        assert len(p.tile_seq) == ntiles


def test_oracle_attention_golden(gold):
    _, t = gold
    q, k, v, cu, ref = t["G3.q"], t["G3.k"], t["G3.v"], t["G3.cu"], t["G3.out"]
    out = []
    for a, b in zip(cu[:-1].tolist(), cu[1:].tolist()):
        qs, ks, vs = (x[a:b].permute(1, 0, 2) for x in (q, k, v))
        pr = torch.softmax((qs @ ks.transpose(1, 2)) / 128 ** 0.5, dim=-1)
        out.append((pr @ vs).permute(1, 0, 2))
    assert (torch.cat(out) - ref).abs().max() < 2e-6


def test_oracle_ada_patch_norm_golden(gold):
    from seedvr2_ref import dit as odit
    _, t = gold
    emb, hid = t["G4.emb"][0], t["G4.hid"]
    sh0, sc0, _ = odit.ada_slices(emb, 0)
    _, _, ga1 = odit.ada_slices(emb, 1)
    assert torch.allclose(hid * (sc0 + t["G4.attn_scale"]) + (sh0 + t["G4.attn_shift"]), t["G4.r_in"], atol=1e-6)
    assert torch.allclose(hid * (ga1 + t["G4.mlp_gate"]), t["G4.r_gate"], atol=1e-6)
    # the vid_out_ada cache-key collision: attn-layer slice + out_* parameters
    assert torch.allclose(hid * (sc0 + t["G4.out_scale"]) + (sh0 + t["G4.out_shift"]), t["G4.r_out"], atol=1e-6)
    y = torch.nn.functional.linear(odit.patch_in(t["G6.x"], (3, 4, 6), (1, 2, 2)), t["G6.win"], t["G6.bin"])
    assert torch.allclose(y, t["G6.y"], atol=1e-6)
    z = odit.patch_out(torch.nn.functional.linear(t["G6.y"], t["G6.wout"], t["G6.bout"]), (3, 2, 3), (1, 2, 2), 3)
    assert torch.allclose(z, t["G6.z"], atol=1e-6)
    assert torch.allclose(odit.rms_norm(t["G7.x"], 1e-5, t["G7.w"]), t["G7.y"], atol=1e-6)


@pytest.mark.parametrize("tag", ["sep", "shared_last"])
def test_oracle_transformer_block_golden(gold, tag):
    """A whole NaMMSRTransformerBlock (reference modules run as-is, RoPE off) vs the oracle's block()."""
    from seedvr2_ref import dit as odit
    meta, t = gold
    m = meta[f"G8.{tag}"]
    cfg = odit.DiTConfig(**m["cfg"])
    p = {k[len(f"G8.{tag}.p."):]: v for k, v in t.items() if k.startswith(f"G8.{tag}.p.")}
    xo, yo, _ = odit.block(p, cfg, 0, t[f"G8.{tag}.vid"], t[f"G8.{tag}.txt"], t[f"G8.{tag}.emb"][0],
                           tuple(m["grid"]), torch.float32, "none", method=m["method"])
    assert (xo - t[f"G8.{tag}.vid_out"]).abs().max() < 2e-5
    if not m["last"]:
        assert (yo - t[f"G8.{tag}.txt_out"]).abs().max() < 2e-5


def test_oracle_7b_transformer_block_golden(gold):
    """NaDiT-7B block (GELU MLP with biases, separate vid/txt weights everywhere, no vid-only last layer):
    reference modules of dit_7b run as-is (RoPE off) vs the oracle."""
    from seedvr2_ref import dit as odit
    meta, t = gold
    cfg = odit.DiTConfig(vid_dim=64, heads=2, head_dim=32, num_layers=2, mm_layers=2, mlp_type="normal",
                         out_norm=False, last_vid_only=False)
    p = {k[len("G11.p."):]: v for k, v in t.items() if k.startswith("G11.p.")}
    xo, yo, _ = odit.block(p, cfg, 0, t["G11.vid"], t["G11.txt"], t["G11.emb"][0], tuple(meta["G11"]["grid"]),
                           torch.float32, "none", method=meta["G11"]["method"])
    assert (xo - t["G11.vid_out"]).abs().max() < 2e-5 and (yo - t["G11.txt_out"]).abs().max() < 2e-5


def test_oracle_rope_properties():
    """RoPE arithmetic is 'parity unpinned' (rotary_embedding_torch absent): check its invariants."""
    from seedvr2_ref import rope
    fr = rope.lang_freqs(42)
    assert fr.numel() == 21 and abs(fr[0].item() - 1.0) < 1e-7 and abs(fr[1].item() - 10000 ** (-2 / 42)) < 1e-7
    av, at = rope.mm_window_angles_3b((2, 3, 4), 58)
    assert av.shape == (24, 126) and at.shape == (58, 126)
    assert torch.equal(at[:, :42], at[:, 42:84]) and torch.equal(at[5, 0::2], at[5, 1::2])
    assert torch.allclose(av[0, :42], (58 * fr).repeat_interleave(2))  # video time offset by text length
    assert av[0, 42:].abs().max() == 0
    x = torch.randn(24, 2, 128)
    y = rope.rotate_interleaved(x, av[:, None, :])
    assert torch.equal(y[..., 126:], x[..., 126:])  # last two dims pass through
    assert torch.allclose(y.pow(2).sum(-1), x.pow(2).sum(-1), rtol=1e-5)  # rotation preserves the norm
    # relative-position property on one axis: <R(a)q, R(b)k> depends only on a-b
    q, k = torch.randn(126), torch.randn(126)
    ang = lambda pos: rope.axis_angles(torch.tensor([pos]), fr).repeat(1, 3)[0]
    d1 = (rope.rotate_interleaved(q, ang(7)) * rope.rotate_interleaved(k, ang(3))).sum()
    d2 = (rope.rotate_interleaved(q, ang(14)) * rope.rotate_interleaved(k, ang(10))).sum()
    assert abs(d1 - d2) < 1e-3


def test_library_exports_every_declared_symbol(pkg):
    """include/seedvr2_hip.h <-> libseedvr2_hip.so <-> ctypes binding agree on the entry points."""
    hdr = open(os.path.join(ROOT, "include", "seedvr2_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(svr2_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations found"
    lib = ctypes.CDLL(pkg.hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert sorted(pkg.hip.EXPORTS) == declared
    lib.svr2_abi_version.restype = ctypes.c_int
    assert lib.svr2_abi_version() == 2
    # header, library and binding agree on the version the descriptor layouts belong to (hip.lib() refuses a mismatch)
    assert int(re.search(r"#define SVR2_ABI_VERSION (\d+)", hdr).group(1)) == pkg.hip.ABI_VERSION == 2


def test_argument_validation_without_gpu(pkg):
    """Bad descriptors are rejected on the host before any launch (safe without a GPU)."""
    hip = pkg.hip
    d = hip.GemmDesc()
    assert hip.lib().svr2_gemm(ctypes.byref(d), None) == -3  # SVR2_E_ARG: null pointers
    assert b"null" in hip.lib().svr2_last_error()
    buf = (ctypes.c_char * 4096)()
    base = (ctypes.addressof(buf) + 15) & ~15
    d.A = d.W = d.C = d.zero_page = base
    d.M, d.N, d.K, d.lda, d.ldc = 4, 8, 100, 128, 8
    assert hip.lib().svr2_gemm(ctypes.byref(d), None) == -1  # SVR2_E_SHAPE: K % 64
    c = hip.Conv3dDesc()
    c.x = c.w = c.y = c.zero_page = base
    c.Cin, c.Cout = 3, 128
    assert hip.lib().svr2_conv3d(ctypes.byref(c), None) == -1  # Cin % 64


def test_conv_weight_tiled_layout(pkg):
    """SVR2_WL_TILED_128x32 (include/seedvr2_hip.h): tile (nb, kt, cs, kx, ky), row r, slot s, element e holds
    w[nb*128 + r, cs*32 + 8*(s ^ f(r)) + e, kt, ky, kx], f(r) = {0,2,3,1}[(r>>2)&3]; rows >= Cout are zero."""
    g = torch.Generator().manual_seed(5)
    cout, cin, kt = 130, 64, 3
    w = torch.randn(cout, cin, kt, 3, 3, generator=g).to(torch.bfloat16)
    t = pkg.hip.conv_weight_tiled(w).reshape(2, kt, cin // 32, 3, 3, 128, 4, 8)
    f = [0, 2, 3, 1]
    rng = np.random.default_rng(0)
    for _ in range(300):
        nb, a, cs, kx, ky, r, s, e = (int(rng.integers(n)) for n in (2, kt, cin // 32, 3, 3, 128, 4, 8))
        n, c = nb * 128 + r, cs * 32 + 8 * (s ^ f[(r >> 2) & 3]) + e
        want = w[n, c, a, ky, kx].item() if n < cout else 0.0
        assert t[nb, a, cs, kx, ky, r, s, e].item() == want


def test_shipped_library_has_no_packed_f32_instructions(pkg):
    """The build flags that keep hipcc from emitting v_pk_(fma|mul|add)_f32 are a workaround for wrong results under kernel
    concurrency (profiles/r02_concurrency): a toolchain or flag change that brings the packed forms back must
    fail here, not in a timing-dependent GPU test.  Disassembles every gfx950 code object of the library that ships."""
    import __graft_entry__ as ge
    bad = ge.packed_f32_instructions(pkg.hip.LIB_PATH)
    assert bad == {}, bad


def test_packed_f32_canary_is_sensitive_and_the_flags_suppress_it():
    """The canary kernel is packed by hipcc's defaults (so the scan above can see the instruction it looks for) and is not
    under the product's flags; the device pass knows the feature (flag_canary raises otherwise)."""
    import __graft_entry__ as ge
    plain = [f for f in ge.DEVICE_FLAGS if f not in ("-fno-slp-vectorize", "-Xclang", "-target-feature", "-packed-fp32-ops")]
    assert ge.flag_canary(plain) > 0
    assert ge.flag_canary() == 0


def test_no_timing_variant_code_in_product_sources():
    """Libraries with wrong results (timing experiments) are patches under tools/timing_variants/, applied to a copy."""
    import glob
    import __graft_entry__ as ge
    for f in glob.glob(os.path.join(ge.PKG_DIR, "csrc", "*")):
        text = open(f).read()
        assert "_DBG" not in text.replace("_dbg.patch", ""), f
    for p in glob.glob(os.path.join(ge.ROOT, "tools", "timing_variants", "*.patch")):
        assert "WRONG results" in open(p).read(), p           # every variant says what it is


def test_timing_variant_patches_still_apply_to_the_product_sources():
    """tools/build_timing_variant.py applies each patch to a copy of csrc/: a kernel edit that breaks one shows up here."""
    import glob
    import shutil
    import subprocess
    import __graft_entry__ as ge
    if shutil.which("patch") is None:
        pytest.skip("no patch(1)")
    for p in sorted(glob.glob(os.path.join(ge.ROOT, "tools", "timing_variants", "*.patch"))):
        with open(p) as fh:
            target = os.path.join(ge.PKG_DIR, "csrc", os.path.basename(fh.readline().split()[1]))
        r = subprocess.run(["patch", "--dry-run", "-s", target, p], capture_output=True, text=True)
        assert r.returncode == 0, (p, r.stdout, r.stderr)


def test_conv_weight_cin3_packing(pkg):
    """SVR2_WL_CIN3 (include/seedvr2_hip.h): k = (kt*3 + ky)*12 + kx*4 + c, pad channel and k >= 108 zero."""
    hip = pkg.hip
    w = torch.randn(128, 3, 3, 3, 3)
    p = hip.conv_weight_cin3(w)
    assert p.shape == (128, 128) and p.dtype == torch.bfloat16
    wb = w.to(torch.bfloat16)
    for (n, c, kt, ky, kx) in [(0, 0, 0, 0, 0), (5, 2, 1, 2, 0), (127, 1, 2, 2, 2), (64, 0, 2, 0, 1)]:
        assert p[n, (kt * 3 + ky) * 12 + kx * 4 + c] == wb[n, c, kt, ky, kx]
    assert (p[:, 108:] == 0).all() and (p[:, 3:108:4] == 0).all()
    with pytest.raises(hip.HipError):
        hip.conv_weight_cin3(torch.randn(64, 3, 3, 3, 3))
