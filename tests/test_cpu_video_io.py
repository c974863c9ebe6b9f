"""CPU tests of the streaming frame I/O: container readers/writers at the byte level, the oracle's
conversion arithmetic, and the chunk-streaming loop against the oracle's restatement of the reference loop.  Nothing
here calls a kernel (the device conversions are covered by tests/test_gpu_video_io.py)."""

import os
import struct

import numpy as np
import pytest
import torch

from seedvr2_ref import video_io as ovio


def _frames(T, H, W, seed=0):
    return np.random.default_rng(seed).integers(0, 256, size=(T, H, W, 3), dtype=np.uint8)


def _read_all(reader):
    buf = np.empty((reader.total_frames, reader.layout.frame_bytes), dtype=np.uint8)
    assert reader.read_into(buf, reader.total_frames) == reader.total_frames
    return buf


# --------------------------------------------------------------------------------------------- oracle arithmetic
def test_oracle_u8_chain_is_the_reference_cast_sequence():
    u8 = np.arange(256, dtype=np.uint8).reshape(1, 16, 16, 1).repeat(3, axis=3)
    x = ovio.u8_to_unit(u8)
    assert x.dtype == torch.bfloat16 and float(x.min()) == 0.0 and float(x.max()) == 1.0
    # the fp16 stop is not a no-op: some levels round differently than fp32 -> bf16 directly
    direct = torch.from_numpy(u8.astype(np.float32) / 255.0).to(torch.bfloat16)
    assert (x.float() - direct.float()).abs().max() <= 2 ** -8
    # monotone, and the output quantisation (truncation) never moves a level by more than one step down
    back = ovio.unit_to_u8(x)
    d = u8.astype(np.int32) - back.astype(np.int32)
    assert d.min() >= -1 and d.max() <= 1
    assert np.all(np.diff(x.float().numpy()[0, :, :, 0].reshape(-1)) >= 0)


def test_oracle_unit_to_u8_truncates():
    x = torch.tensor([0.0, 0.999 / 255, 1.0 / 255, 254.99 / 255, 1.0], dtype=torch.float32).view(1, 1, 5, 1).repeat(1, 1, 1, 3)
    got = ovio.unit_to_u8(x)[0, 0, :, 0].tolist()
    assert got[0] == 0 and got[1] == 0 and got[4] == 255 and got[3] == 254


@pytest.mark.parametrize("sub", [1, 2])
def test_oracle_yuv_round_trip_error_bound(sub):
    rgb = _frames(2, 16, 24, seed=3)
    if sub == 2:   # smooth content: 4:2:0 is only near-lossless when chroma varies slowly
        rgb = np.repeat(np.repeat(_frames(2, 8, 12, seed=3), 2, axis=1), 2, axis=2)
    raw = ovio.rgb_to_yuv(rgb, sub)
    assert raw.shape == (2, 16 * 24 + 2 * (16 // sub) * (24 // sub))
    y = raw[:, :16 * 24]
    assert y.min() >= 16 and y.max() <= 235
    back = ovio.yuv_to_rgb(raw, 16, 24, sub)
    assert np.abs(back.astype(np.int32) - rgb.astype(np.int32)).max() <= 3      # studio-range 8-bit quantisation
    grey = np.full((1, 4, 4, 3), 128, dtype=np.uint8)
    g = ovio.rgb_to_yuv(grey, sub)
    assert np.all(g[:, 16:] == 128) and np.all(g[:, :16] == ((66 + 129 + 25) * 128 + 128 >> 8) + 16)


def test_oracle_bgr_rows_layout():
    rgb = _frames(2, 3, 5, seed=1)
    stride = (5 * 3 + 3) & ~3
    raw = ovio.rgb_to_bgr_rows(rgb, bottom_up=True, row_stride=stride)
    assert raw.shape == (2, 3 * stride)
    rows = raw.reshape(2, 3, stride)
    assert np.all(rows[:, :, 15:] == 0)
    assert tuple(rows[0, 2, 0:3]) == tuple(rgb[0, 0, 0, ::-1])                  # bottom-up: last stored row = top row
    assert np.array_equal(ovio.bgr_rows_to_rgb(raw, 3, 5, True, stride), rgb)


# --------------------------------------------------------------------------------------------- containers
def test_y4m_write_read_bytes(pkg, tmp_path):
    vio = pkg.video_io
    rgb = _frames(5, 8, 12, seed=2)
    raw = ovio.rgb_to_yuv(rgb, 2)
    p = str(tmp_path / "a.y4m")
    w = vio.Y4MWriter(p, 12, 8, 24.0)
    assert w.layout.kind == "yuv" and w.layout.sub == 2 and w.layout.frame_bytes == raw.shape[1]
    w.write_raw(raw[:2])
    w.write_raw(raw[2:])
    w.close()
    with open(p, "rb") as fh:
        assert fh.readline() == b"YUV4MPEG2 W12 H8 F24:1 Ip A1:1 C420jpeg\n"
    r = vio.open_reader(p)
    assert (r.total_frames, r.width, r.height, r.fps) == (5, 12, 8, 24.0)
    assert np.array_equal(_read_all(r), raw)
    r.seek(3)
    buf = np.empty((4, r.layout.frame_bytes), dtype=np.uint8)
    assert r.read_into(buf, 4) == 2 and np.array_equal(buf[:2], raw[3:])
    r.close()
    # odd sizes fall back to 4:4:4; NTSC rates keep their fraction
    w = vio.Y4MWriter(str(tmp_path / "b.y4m"), 7, 5, 30000 / 1001)
    assert w.layout.sub == 1
    w.close()
    with open(tmp_path / "b.y4m", "rb") as fh:
        assert fh.readline() == b"YUV4MPEG2 W7 H5 F30000:1001 Ip A1:1 C444\n"


def test_avi_write_read_bytes_and_structure(pkg, tmp_path):
    vio = pkg.video_io
    rgb = _frames(4, 6, 5, seed=4)                      # 5*3 = 15 bytes per row -> padded to 16
    p = str(tmp_path / "a.avi")
    w = vio.AVIWriter(p, 5, 6, 25.0)
    lay = w.layout
    assert lay.row_stride == 16 and lay.flags == pkg.hip.PIX_SWAP_RB | pkg.hip.PIX_BOTTOM_UP
    raw = ovio.rgb_to_bgr_rows(rgb, bottom_up=True, row_stride=16)
    w.write_raw(raw[:1])
    w.write_raw(raw[1:])
    w.close()
    data = open(p, "rb").read()
    assert data[:4] == b"RIFF" and data[8:12] == b"AVI " and struct.unpack("<I", data[4:8])[0] == len(data) - 8
    assert data.count(b"00db") == 8 and b"idx1" in data            # four chunks + four index entries
    r = vio.open_reader(p)
    assert (r.total_frames, r.width, r.height) == (4, 5, 6) and abs(r.fps - 25.0) < 1e-6
    assert r.layout.row_stride == 16 and r.layout.flags == lay.flags
    got = _read_all(r)
    assert np.array_equal(got, raw)
    assert np.array_equal(ovio.bgr_rows_to_rgb(got, 6, 5, True, 16), rgb)
    r.close()


def test_avi_mjpeg_frames_are_decoded(pkg, tmp_path):
    """An AVI whose stream is MJPG (what cameras and cv2's 'MJPG' fourcc write) is read through PIL."""
    import io
    from PIL import Image
    vio = pkg.video_io
    rgb = np.repeat(np.repeat(_frames(2, 4, 4, seed=5), 8, axis=1), 8, axis=2)      # 32 x 32 blocks: JPEG-friendly
    jpgs = []
    for f in rgb:
        b = io.BytesIO()
        Image.fromarray(f).save(b, format="JPEG", quality=95, subsampling=0)
        jpgs.append(b.getvalue())
    strf = struct.pack("<IiiHHIIiiII", 40, 32, 32, 1, 24, struct.unpack("<I", b"MJPG")[0], 0, 0, 0, 0, 0)
    strh = struct.pack("<4s4sIHHIIIIIIIIhhhh", b"vids", b"MJPG", 0, 0, 0, 0, 1, 30, 0, 2, 0, 0xFFFFFFFF, 0, 0, 0, 32, 32)
    strl = b"strl" + b"strh" + struct.pack("<I", len(strh)) + strh + b"strf" + struct.pack("<I", len(strf)) + strf
    avih = struct.pack("<14I", 33333, 0, 0, 0x10, 2, 0, 1, 0, 32, 32, 0, 0, 0, 0)
    hdrl = b"hdrl" + b"avih" + struct.pack("<I", len(avih)) + avih + b"LIST" + struct.pack("<I", len(strl)) + strl
    movi = b"movi" + b"".join(b"00dc" + struct.pack("<I", len(j)) + j + (b"\0" if len(j) & 1 else b"") for j in jpgs)
    body = b"AVI " + b"LIST" + struct.pack("<I", len(hdrl)) + hdrl + b"LIST" + struct.pack("<I", len(movi)) + movi
    p = str(tmp_path / "m.avi")
    open(p, "wb").write(b"RIFF" + struct.pack("<I", len(body)) + body)
    r = vio.open_reader(p)
    assert r.mjpg and r.total_frames == 2 and r.layout.flags == 0 and abs(r.fps - 30.0) < 1e-6
    got = _read_all(r).reshape(2, 32, 32, 3)
    assert np.abs(got.astype(np.int32) - rgb.astype(np.int32)).max() <= 12
    r.close()


def test_npy_png_and_image_writers(pkg, tmp_path):
    from PIL import Image
    vio = pkg.video_io
    rgb = _frames(3, 6, 8, seed=6)
    raw = rgb.reshape(3, -1)
    w = vio.open_writer(str(tmp_path / "o.npy"), 8, 6)
    w.write_raw(raw[:2])
    w.write_raw(raw[2:])
    w.close()
    assert np.array_equal(np.load(tmp_path / "o.npy", allow_pickle=False), rgb)
    r = vio.open_reader(str(tmp_path / "o.npy"))
    assert r.total_frames == 3 and np.array_equal(_read_all(r), raw)
    # float tensors in [0,1] (e.g. a ComfyUI IMAGE dump) are taken back to their 8-bit levels
    torch.save(torch.from_numpy(rgb.astype(np.float32) / 255.0), str(tmp_path / "f.pt"))
    assert np.array_equal(_read_all(vio.open_reader(str(tmp_path / "f.pt"))), raw)
    # png sequence: {base}_{index:06d}.png like the reference
    w = vio.open_writer(str(tmp_path / "seq"), 8, 6, base_name="clip")
    w.write_raw(raw)
    w.close()
    names = sorted(os.listdir(tmp_path / "seq"))
    assert names == ["clip_000000.png", "clip_000001.png", "clip_000002.png"]
    assert np.array_equal(np.asarray(Image.open(tmp_path / "seq" / names[1])), rgb[1])
    d = vio.open_reader(str(tmp_path / "seq"))
    assert d.total_frames == 3 and np.array_equal(_read_all(d), raw)
    w = vio.open_writer(str(tmp_path / "one.png"), 8, 6, single_image=True)
    w.write_raw(raw[:1])
    w.close()
    one = vio.open_reader(str(tmp_path / "one.png"))
    assert one.total_frames == 1 and not one.has_alpha and np.array_equal(_read_all(one), raw[:1])
    Image.fromarray(np.dstack([rgb[0], np.full((6, 8), 200, np.uint8)]), "RGBA").save(tmp_path / "a.png")
    assert vio.open_reader(str(tmp_path / "a.png")).has_alpha


def test_unavailable_backends_fail_loudly(pkg, tmp_path):
    vio = pkg.video_io
    (tmp_path / "x.mp4").write_bytes(b"\0" * 64)
    if not vio._have_cv2() and not __import__("shutil").which("ffmpeg"):
        with pytest.raises(vio.VideoIOError, match="cv2 or an ffmpeg"):
            vio.open_reader(str(tmp_path / "x.mp4"))
        with pytest.raises(vio.VideoIOError, match="cv2 or an ffmpeg"):
            vio.open_writer(str(tmp_path / "y.mp4"), 8, 8)
    with pytest.raises(FileNotFoundError):
        vio.open_reader(str(tmp_path / "missing.y4m"))
    (tmp_path / "bad.y4m").write_bytes(b"not a y4m\n")
    with pytest.raises(vio.VideoIOError):
        vio.open_reader(str(tmp_path / "bad.y4m"))
    (tmp_path / "bad.avi").write_bytes(b"RIFFxxxxWAVE")
    with pytest.raises(vio.VideoIOError):
        vio.AVIReader(str(tmp_path / "bad.avi"))


# --------------------------------------------------------------------------------------------- the streaming loop
class _FakeSource:
    def __init__(self, frames):
        self.frames, self.pos, self.calls = frames, 0, []

    def next_chunk(self, n, prefetch_next=0):
        self.calls.append((n, prefetch_next))
        out = self.frames[self.pos:self.pos + n]
        self.pos += out.shape[0]
        return out if out.shape[0] else None


class _FakeSink:
    def __init__(self):
        self.parts, self.frames_written, self.closed = [], 0, False

    def write(self, x):
        self.parts.append(x.clone())
        self.frames_written += x.shape[0]

    def close(self):
        self.closed = True


@pytest.mark.parametrize("total,chunk,overlap,prepend", [(11, 4, 2, 0), (11, 4, 0, 3), (9, 3, 3, 2), (7, 0, 2, 1),
                                                          (5, 8, 1, 0), (10, 5, 7, 0), (1, 1, 1, 0)])
def test_stream_upscale_matches_reference_loop(pkg, total, chunk, overlap, prepend):
    """Same frames go to the upscaler with the same prepend flag and the same outputs are kept as in the oracle's
    restatement of _stream_video_chunks (inference_cli.py:566-654)."""
    vio = pkg.video_io
    frames = torch.arange(total, dtype=torch.float32).view(total, 1, 1, 1).repeat(1, 2, 2, 3)
    seen_a, seen_b = [], []

    def proc(log):
        def f(x, p):
            log.append((x[:, 0, 0, 0].tolist(), p))
            return x * 2 + 1                  # one output frame per input frame
        return f

    want = ovio.stream_chunks(frames, chunk, overlap, prepend, proc(seen_a))
    src, sink = _FakeSource(frames), _FakeSink()
    n = vio.stream_upscale(src, lambda w, h: sink, proc(seen_b), total, chunk, overlap, prepend)
    assert seen_a == seen_b
    assert n == sum(w.shape[0] for w in want) == total and sink.closed
    assert len(sink.parts) == len(want) and all(torch.equal(a, b) for a, b in zip(sink.parts, want))
    # the next chunk is requested while the current one is still being processed
    assert all(nxt == min(chunk or total, total - sum(c[0] for c in src.calls[:i + 1])) for i, (_, nxt) in enumerate(src.calls))


def test_cli_output_paths_follow_reference_rules(pkg, tmp_path):
    import importlib
    cli = importlib.import_module(pkg.__name__ + ".inference_cli")
    vid = str(tmp_path / "in" / "clip.mp4")
    g = cli.generate_output_path
    assert g(vid, "mp4") == str(tmp_path / "in" / "clip_upscaled.mp4")
    assert g(vid, "png", input_type="video") == str(tmp_path / "in" / "clip_upscaled")
    assert g(str(tmp_path / "in" / "a.jpg"), "png", input_type="image") == str(tmp_path / "in" / "a_upscaled.png")
    assert g(vid, "mp4", from_directory=True) == str(tmp_path / "in_upscaled" / "clip.mp4")
    assert g(vid, "y4m", output_dir=str(tmp_path / "o")) == str(tmp_path / "o" / "clip.y4m")
    vio = pkg.video_io
    (tmp_path / "t.npy").write_bytes(b"")
    assert cli.get_input_type(str(tmp_path / "t.npy"), vio) == "video"
    assert cli.get_input_type(str(tmp_path), vio) == "directory"
    with pytest.raises(FileNotFoundError):
        cli.get_input_type(str(tmp_path / "nope.mp4"), vio)


def test_prepend_reversed_matches_pad_video_temporal(pkg):
    """[ref: src/core/generation_utils.py:593-652], prepend=True."""
    f = pkg.nodes.SeedVR2VideoUpscaler.prepend_reversed
    x = torch.arange(5.0).view(5, 1, 1, 1)
    ids = lambda t: t.view(-1).tolist()
    assert ids(f(x, 0)) == [0, 1, 2, 3, 4]
    assert ids(f(x, 2)) == [2, 1, 0, 1, 2, 3, 4]
    assert ids(f(x, 4)) == [4, 3, 2, 1, 0, 1, 2, 3, 4]
    assert ids(f(x, 5)) == [4, 4, 3, 2, 1, 0, 1, 2, 3, 4]            # count >= T: last frame repeated count-T+1 times
    assert ids(f(x, 7)) == [4, 4, 4, 4, 3, 2, 1, 0, 1, 2, 3, 4]
    assert ids(f(x[:1], 2)) == [0, 0, 0]
