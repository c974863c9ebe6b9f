"""GPU parity tests of the streaming frame I/O: the device conversions against the oracle's numpy
restatement, bit for bit (integer / cast arithmetic), and the reader -> device -> writer pipeline on real container files."""

import numpy as np
import pytest
import torch

from seedvr2_ref import video_io as ovio

pytestmark = pytest.mark.gpu


def _frames(T, H, W, seed=0):
    return np.random.default_rng(seed).integers(0, 256, size=(T, H, W, 3), dtype=np.uint8)


def _all_unit_bf16(gpu):
    """Every bf16 value in [0, 1] (16 257 bit patterns), padded to a [1, H, W, 3] frame."""
    vals = torch.arange(0, 0x3F80 + 1, dtype=torch.int32).to(torch.int16).view(torch.bfloat16)
    W = 129
    rows = (vals.numel() + W * 3 - 1) // (W * 3)
    vals = torch.cat((vals, vals[: rows * W * 3 - vals.numel()]))
    return vals.view(1, rows, W, 3).to(gpu).contiguous()


@pytest.mark.parametrize("flags,stride_pad", [(0, 0), (1, 0), (2, 0), (3, 1), (3, 3)])
def test_u8_to_frames_matches_oracle(pkg, gpu, flags, stride_pad):
    hip, vio = pkg.hip, pkg.video_io
    T, H, W = 3, 10, 7
    rgb = _frames(T, H, W, seed=flags)
    rgb[0, 0, :, :] = np.arange(W * 3, dtype=np.uint8).reshape(W, 3)            # a row that shows channel / row order
    stride = W * 3 + stride_pad
    if flags & hip.PIX_SWAP_RB:
        raw = ovio.rgb_to_bgr_rows(rgb, bool(flags & hip.PIX_BOTTOM_UP), stride)
    else:
        rows = np.zeros((T, H, stride), np.uint8)
        rows[:, :, : W * 3] = (rgb[:, ::-1] if flags & hip.PIX_BOTTOM_UP else rgb).reshape(T, H, W * 3)
        raw = rows.reshape(T, -1)
    lay = vio.Layout("rgb", W, H, flags=flags, row_stride=stride)
    got = vio.frames_from_bytes(torch.from_numpy(raw).to(gpu), lay)
    want = ovio.u8_to_unit(rgb)
    assert got.dtype == torch.bfloat16 and torch.equal(got.cpu(), want)


def test_u8_chain_all_levels(pkg, gpu):
    vio = pkg.video_io
    u8 = np.arange(256, dtype=np.uint8).reshape(1, 16, 16, 1).repeat(3, axis=3)
    got = vio.frames_from_bytes(torch.from_numpy(u8.reshape(1, -1)).to(gpu), vio.Layout("rgb", 16, 16))
    assert torch.equal(got.cpu(), ovio.u8_to_unit(u8))


@pytest.mark.parametrize("flags,stride_pad", [(0, 0), (1, 0), (3, 1), (2, 3)])
def test_frames_to_u8_every_bf16_value(pkg, gpu, flags, stride_pad):
    """trunc(x * 255) for EVERY bf16 value in [0,1], in each channel/row order, with zeroed row padding."""
    hip, vio = pkg.hip, pkg.video_io
    x = _all_unit_bf16(gpu)
    _, H, W, _ = x.shape
    stride = W * 3 + stride_pad
    lay = vio.Layout("rgb", W, H, flags=flags, row_stride=stride)
    out = torch.full((1, lay.frame_bytes), 0xAA, dtype=torch.uint8, device=gpu)
    vio.frames_to_bytes(x, lay, out)
    u8 = ovio.unit_to_u8(x)
    if flags & hip.PIX_SWAP_RB:
        want = ovio.rgb_to_bgr_rows(u8, bool(flags & hip.PIX_BOTTOM_UP), stride)
    else:
        rows = np.zeros((1, H, stride), np.uint8)
        rows[:, :, : W * 3] = (u8[:, ::-1] if flags & hip.PIX_BOTTOM_UP else u8).reshape(1, H, W * 3)
        want = rows.reshape(1, -1)
    assert np.array_equal(out.cpu().numpy(), want)


def test_frames_to_u8_clamps_out_of_range(pkg, gpu):
    vio = pkg.video_io
    x = torch.tensor([-0.5, 0.0, 1.0, 1.5, float("inf"), -float("inf")], dtype=torch.bfloat16, device=gpu).view(1, 1, 2, 3)
    got = vio.frames_to_bytes(x.contiguous(), vio.Layout("rgb", 2, 1)).cpu().view(-1).tolist()
    assert got == [0, 0, 255, 255, 255, 0]


@pytest.mark.parametrize("sub,H,W", [(2, 12, 20), (1, 12, 20), (1, 7, 5), (2, 2, 2)])
def test_yuv_both_ways_match_oracle(pkg, gpu, sub, H, W):
    vio = pkg.video_io
    rgb = _frames(3, H, W, seed=sub + H)
    x = ovio.u8_to_unit(rgb).to(gpu)
    lay = vio.Layout("yuv", W, H, sub=sub)
    raw = vio.frames_to_bytes(x, lay)
    want = ovio.rgb_to_yuv(ovio.unit_to_u8(x.cpu()), sub)
    assert np.array_equal(raw.cpu().numpy(), want)
    back = vio.frames_from_bytes(raw, lay)
    assert torch.equal(back.cpu(), ovio.u8_to_unit(ovio.yuv_to_rgb(want, H, W, sub)))
    # extreme chroma exercises the clipping of the inverse matrix
    ext = np.zeros((1, lay.frame_bytes), np.uint8)
    ext[:, : H * W] = 235
    ext[:, H * W:] = 240
    got = vio.frames_from_bytes(torch.from_numpy(ext).to(gpu), lay)
    assert torch.equal(got.cpu(), ovio.u8_to_unit(ovio.yuv_to_rgb(ext, H, W, sub)))


def test_shape_errors_are_loud(pkg, gpu):
    hip, vio = pkg.hip, pkg.video_io
    x = torch.zeros(1, 5, 7, 3, dtype=torch.bfloat16, device=gpu)
    with pytest.raises(hip.HipError, match="multiple of the chroma"):
        hip.frames_to_yuv(x, torch.zeros(200, dtype=torch.uint8, device=gpu), 2, 35 + 2 * 12)
    with pytest.raises(hip.HipError):
        hip.frames_to_u8(x, torch.zeros(10, dtype=torch.uint8, device=gpu), 21, 105)         # buffer too small
    with pytest.raises(hip.HipError):
        hip.frames_to_u8(x.float(), torch.zeros(105, dtype=torch.uint8, device=gpu), 21, 105)  # not bf16
    with pytest.raises(vio.VideoIOError):
        vio.frames_to_bytes(x, vio.Layout("rgb", 8, 5))


def test_files_through_the_device_pipeline(pkg, gpu, tmp_path):
    """AVI (bottom-up BGR, padded rows) -> DeviceSource -> chunked 'upscale' with context frames -> DeviceSink -> y4m, and
    y4m -> ... -> png sequence / avi: what lands on disk equals the oracle's chain applied to the oracle's loop."""
    vio = pkg.video_io
    T, H, W = 11, 6, 10
    rgb = _frames(T, H, W, seed=9)
    src_path = str(tmp_path / "in.avi")
    w = vio.AVIWriter(src_path, W, H, 24.0)
    w.write_raw(ovio.rgb_to_bgr_rows(rgb, True, w.layout.row_stride))
    w.close()

    def up(x, prepend):       # 2x nearest 'upscaler' that mixes neighbouring frames so that context matters
        y = x.float()
        y = (y + torch.roll(y, 1, 0)) * 0.5
        return y.repeat_interleave(2, 1).repeat_interleave(2, 2).to(torch.bfloat16)

    want = torch.cat(ovio.stream_chunks(ovio.u8_to_unit(rgb), 4, 2, 0, lambda x, p: up(x, p).cpu()), dim=0)
    assert want.shape == (T, 2 * H, 2 * W, 3)

    out_path = str(tmp_path / "out.y4m")
    reader = vio.open_reader(src_path)
    source = vio.DeviceSource(reader, gpu)
    n = vio.stream_upscale(source, lambda ww, hh: vio.DeviceSink(vio.open_writer(out_path, ww, hh, reader.fps), gpu), up,
                           T, chunk_size=4, overlap=2)
    source.close()
    assert n == T
    r = vio.open_reader(out_path)
    assert (r.total_frames, r.width, r.height, r.fps) == (T, 2 * W, 2 * H, 24.0)
    got = np.empty((T, r.layout.frame_bytes), np.uint8)
    assert r.read_into(got, T) == T
    assert np.array_equal(got, ovio.rgb_to_yuv(ovio.unit_to_u8(want), 2))
    r.close()

    # second hop: y4m in, skip 2 frames, cap at 7, png sequence out
    seq = str(tmp_path / "seq")
    reader = vio.open_reader(out_path)
    reader.seek(2)
    source = vio.DeviceSource(reader, gpu)
    n = vio.stream_upscale(source, lambda ww, hh: vio.DeviceSink(vio.open_writer(seq, ww, hh, 24.0, base_name="f"), gpu),
                           lambda x, p: x, 7, chunk_size=3, overlap=0)
    source.close()
    assert n == 7
    back = vio.open_reader(seq)
    png = np.empty((7, back.layout.frame_bytes), np.uint8)
    assert back.read_into(png, 7) == 7
    dec = ovio.u8_to_unit(ovio.yuv_to_rgb(got[2:9], 2 * H, 2 * W, 2))
    assert np.array_equal(png.reshape(7, 2 * H, 2 * W, 3), ovio.unit_to_u8(dec))


def test_sink_surfaces_writer_errors(pkg, gpu, tmp_path):
    vio = pkg.video_io

    class Boom(vio.Writer):
        layout = vio.Layout("rgb", 4, 4)

        def write_raw(self, buf):
            raise IOError("disk full")

    sink = vio.DeviceSink(Boom(), gpu)
    sink.write(torch.zeros(2, 4, 4, 3, dtype=torch.bfloat16, device=gpu))
    with pytest.raises(IOError, match="disk full"):
        sink.close()


def test_full_size_quantisation_is_a_level_map(pkg, gpu):
    """4K, 17 frames: u8 -> frames -> u8 is the same 256-entry level map at every pixel (the map the oracle computes on
    the 256 levels), and the yuv output equals the oracle on a strip of the same frames."""
    vio = pkg.video_io
    T, H, W = 17, 2160, 3840
    g = torch.Generator(device=gpu).manual_seed(5)
    raw = torch.randint(0, 256, (T, H * W * 3), dtype=torch.uint8, device=gpu, generator=g)
    lay = vio.Layout("rgb", W, H)
    x = vio.frames_from_bytes(raw, lay)
    out = vio.frames_to_bytes(x, lay)
    levels = np.arange(256, dtype=np.uint8).reshape(1, 16, 16, 1).repeat(3, axis=3)
    lut = torch.from_numpy(ovio.unit_to_u8(ovio.u8_to_unit(levels))[0, :, :, 0].reshape(-1).copy()).to(gpu)
    for t in (0, 8, 16):
        assert torch.equal(out[t], lut[raw[t].long()])
    ylay = vio.Layout("yuv", W, H, sub=2)
    yuv = vio.frames_to_bytes(x, ylay)
    strip = x[3:4, 1000:1016].contiguous()
    want = ovio.rgb_to_yuv(ovio.unit_to_u8(strip.cpu()), 2)
    y = yuv[3, 1000 * W:1016 * W].cpu().numpy()
    assert np.array_equal(y, want[0, : 16 * W])
    u = yuv[3, H * W + 500 * (W // 2): H * W + 508 * (W // 2)].cpu().numpy()
    assert np.array_equal(u, want[0, 16 * W: 16 * W + 8 * (W // 2)])
