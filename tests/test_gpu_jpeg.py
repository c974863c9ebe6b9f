"""GPU parity tests of the device JPEG encoder (csrc/jpeg.hip): bit-exact scans against the numpy oracle, files that an
independent decoder (PIL / libjpeg) reads, and the Motion-JPEG AVI writer behind the streaming sink."""

import io

import numpy as np
import pytest
import torch
from PIL import Image

from seedvr2_ref import jpeg as oj
from seedvr2_ref import video_io as ovio

pytestmark = pytest.mark.gpu


def _huff(J):
    return [(J.huffman_codes(*J.DC_LUMA), J.huffman_codes(*J.AC_LUMA)), (J.huffman_codes(*J.DC_CHROMA), J.huffman_codes(*J.AC_CHROMA))]


def _frames(T, h, w, seed=1, noise=3.0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    out = []
    for t in range(T):
        img = np.stack([(np.sin((xx + 3 * t) / 9.0) * 0.5 + 0.5) * 255, (np.cos(yy / 7.0) * 0.5 + 0.5) * 255, ((xx + yy + t) % 64) * 4], -1)
        out.append((img + rng.normal(0, noise, img.shape)).clip(0, 255).astype(np.uint8))
    return np.stack(out)


@pytest.mark.parametrize("T,h,w,q", [(2, 48, 64, 90), (1, 16, 16, 50), (3, 33, 47, 75), (1, 70, 100, 100), (1, 40, 24, 10)])
def test_scan_is_bit_exact(pkg, gpu, T, h, w, q):
    J = pkg.jpeg
    u8 = _frames(T, h, w, seed=q)
    x = ovio.u8_to_unit(u8).to(gpu)
    packed, lens = pkg.hip.jpeg_encode(x, q)
    lens = lens.cpu().tolist()
    div, _ = J.device_tables(q)
    back = ovio.unit_to_u8(x.cpu())                      # the encoder sees trunc(bf16 * 255), like every other writer
    for t in range(T):
        want = oj.encode_scan(back[t], div, J.ZIGZAG, _huff(J))
        got = packed[t, :lens[t]].cpu().numpy().tobytes()
        assert lens[t] == len(want) and got == want, (t, lens[t], len(want))
        data = J.frame_header(w, h, q) + got + b"\xff\xd9"
        dec = np.asarray(Image.open(io.BytesIO(data)).convert("RGB")).astype(np.float64)
        assert dec.shape == (h, w, 3)
        if q >= 50:
            assert 10 * np.log10(255 ** 2 / ((dec - back[t]) ** 2).mean()) > 26


def test_extreme_content(pkg, gpu):
    """Saturated checkerboards (largest coefficients, long codes), flat frames (EOB only) and white noise at q = 100
    (longest scans, many stuffed 0xFF bytes) against the oracle."""
    J = pkg.jpeg
    yy, xx = np.mgrid[0:32, 0:48]
    chk = (((yy + xx) & 1) * 255).astype(np.uint8)[..., None].repeat(3, -1)
    flat = np.full((32, 48, 3), 255, np.uint8)
    noise = np.random.default_rng(5).integers(0, 256, (32, 48, 3), dtype=np.uint8)
    u8 = np.stack([chk, flat, noise])
    x = ovio.u8_to_unit(u8).to(gpu)
    back = ovio.unit_to_u8(x.cpu())
    for q in (100, 60):
        packed, lens = pkg.hip.jpeg_encode(x, q)
        div, _ = J.device_tables(q)
        for t in range(3):
            want = oj.encode_scan(back[t], div, J.ZIGZAG, _huff(J))
            assert packed[t, :int(lens[t])].cpu().numpy().tobytes() == want
    assert int(lens[1]) < 40 and b"\xff\x00" in packed[2, :int(lens[2])].cpu().numpy().tobytes()


def test_mjpeg_sink_to_avi_and_back(pkg, gpu, tmp_path):
    vio, J = pkg.video_io, pkg.jpeg
    u8 = _frames(11, 40, 56, seed=3)
    x = ovio.u8_to_unit(u8).to(gpu)
    path = str(tmp_path / "o.avi")
    sink = vio.DeviceSink(vio.open_writer(path, 56, 40, 25.0, mjpeg_quality=92), gpu)
    sink.write(x[:9])          # more than one encode batch of eight
    sink.write(x[9:])
    sink.close()
    assert sink.frames_written == 11
    r = vio.open_reader(path)
    assert r.mjpg and (r.total_frames, r.width, r.height, r.fps) == (11, 56, 40, 25.0)
    buf = np.empty((11, r.layout.frame_bytes), np.uint8)
    assert r.read_into(buf, 11) == 11
    got = buf.reshape(11, 40, 56, 3).astype(np.float64)
    psnr = 10 * np.log10(255 ** 2 / ((got - u8) ** 2).mean())
    assert psnr > 26, psnr          # sawtooth + noise test pattern: libjpeg itself reaches 28 dB here
    # the bytes of frame 4 are header + oracle scan + EOI
    off, size = r.frames[4]
    raw = open(path, "rb").read()[off:off + size]
    div, _ = J.device_tables(92)
    want = J.frame_header(56, 40, 92) + oj.encode_scan(ovio.unit_to_u8(x[4:5].cpu())[0], div, J.ZIGZAG,
                                                       [(J.huffman_codes(*J.DC_LUMA), J.huffman_codes(*J.AC_LUMA)),
                                                        (J.huffman_codes(*J.DC_CHROMA), J.huffman_codes(*J.AC_CHROMA))]) + b"\xff\xd9"
    assert raw == want


def test_full_size_frames_decode(pkg, gpu):
    """4K: two frames encode, PIL decodes them at full size with high PSNR, and the first MCU row equals the oracle."""
    J = pkg.jpeg
    g = torch.Generator().manual_seed(9)
    low = 0.5 + 0.2 * torch.randn(2, 3, 135, 240, generator=g)
    rgb = torch.nn.functional.interpolate(low, size=(2160, 3840), mode="bicubic").clamp(0, 1)
    x = rgb.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(gpu)
    packed, lens = pkg.hip.jpeg_encode(x, 90)
    back = ovio.unit_to_u8(x.cpu())
    for t in range(2):
        n = int(lens[t])
        assert 0 < n < 2160 * 3840 * 3 // 10                     # smooth content: far below a tenth of the raw size
        data = J.frame_header(3840, 2160, 90) + packed[t, :n].cpu().numpy().tobytes() + b"\xff\xd9"
        dec = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        assert dec.shape == (2160, 3840, 3)
        assert 10 * np.log10(255 ** 2 / ((dec.astype(np.float64) - back[t]) ** 2).mean()) > 40
    div, _ = J.device_tables(90)
    row0 = oj.encode_row(oj.coefficients(back[0][:16], div, J.ZIGZAG)[0], _huff(J))
    assert packed[0, :len(row0)].cpu().numpy().tobytes() == row0
