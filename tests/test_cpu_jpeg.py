"""CPU tests of the Motion-JPEG path: the tables against libjpeg's defaults (read back from a PIL-written file), the
oracle's integer DCT against scipy, and the oracle encoder's files against an independent decoder (PIL / libjpeg)."""

import io
import struct

import numpy as np
import pytest
from PIL import Image

from seedvr2_ref import jpeg as oj


def _segments(data):
    i, out = 2, []
    while i < len(data):
        m, L = data[i + 1], struct.unpack(">H", data[i + 2:i + 4])[0]
        out.append((m, data[i + 4:i + 2 + L]))
        if m == 0xDA:
            break
        i += 2 + L
    return out


def _huff(J):
    return [(J.huffman_codes(*J.DC_LUMA), J.huffman_codes(*J.AC_LUMA)), (J.huffman_codes(*J.DC_CHROMA), J.huffman_codes(*J.AC_CHROMA))]


def _test_image(h, w, seed=1):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(np.sin(xx / 9.0) * 0.5 + 0.5) * 255, (np.cos(yy / 7.0) * 0.5 + 0.5) * 255, ((xx + yy) % 64) * 4], -1)
    return (img + rng.normal(0, 3, img.shape)).clip(0, 255).astype(np.uint8)


def test_tables_are_the_standard_ones(pkg):
    J = pkg.jpeg
    b = io.BytesIO()
    Image.fromarray(_test_image(32, 32)).save(b, "JPEG", quality=90, subsampling=2, optimize=False)
    dht, dqt = {}, {}
    for m, p in _segments(b.getvalue()):
        k = 0
        while m == 0xC4 and k < len(p):
            n = sum(p[k + 1:k + 17])
            dht[p[k]] = (list(p[k + 1:k + 17]), list(p[k + 17:k + 17 + n]))
            k += 17 + n
        while m == 0xDB and k < len(p):
            dqt[p[k]] = list(p[k + 1:k + 65])
            k += 65
    for tc, mine in ((0x00, J.DC_LUMA), (0x10, J.AC_LUMA), (0x01, J.DC_CHROMA), (0x11, J.AC_CHROMA)):
        assert dht[tc] == (list(mine[0]), list(mine[1]))
    ql, qc = J.quant_tables(90)
    assert dqt[0] == ql[J.ZIGZAG].tolist() and dqt[1] == qc[J.ZIGZAG].tolist()
    assert sorted(J.ZIGZAG.tolist()) == list(range(64))
    div, huff = J.device_tables(75)
    assert div.shape == (2, 64) and div[0, 0] == 8 * J.quant_tables(75)[0][0]
    assert huff.shape == (2, 268) and (huff[:, :12] >> 16).min() >= 2 and int((huff[0, 12:] != 0).sum()) == 162
    codes = J.huffman_codes(*J.AC_LUMA)
    assert codes[0x00] == (0b1010, 4) and codes[0xF0] == (0b11111111001, 11) and codes[0x01] == (0b00, 2)     # T.81 K.5


def test_oracle_integer_dct_matches_float_dct():
    from scipy.fft import dctn
    blk = np.random.default_rng(2).integers(-128, 128, size=(500, 8, 8))
    ref = dctn(blk.astype(np.float64), axes=(-2, -1), norm="ortho") * 8
    assert np.abs(oj.fdct(blk) - ref).max() < 1.6          # output is scaled by 8: well under a quarter quantisation step
    flat = np.full((1, 8, 8), 100)
    out = oj.fdct(flat)
    assert out[0, 0, 0] == 100 * 64 and np.abs(out).sum() == 100 * 64


@pytest.mark.parametrize("h,w,q", [(70, 100, 90), (16, 16, 50), (33, 47, 75), (48, 64, 100)])
def test_oracle_files_decode_with_libjpeg(pkg, h, w, q):
    J = pkg.jpeg
    img = _test_image(h, w)
    div, _ = J.device_tables(q)
    data = J.frame_header(w, h, q) + oj.encode_scan(img, div, J.ZIGZAG, _huff(J)) + b"\xff\xd9"
    dec = np.asarray(Image.open(io.BytesIO(data)).convert("RGB")).astype(np.float64)
    assert dec.shape == (h, w, 3)
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", quality=q, subsampling=2)
    pil = np.asarray(Image.open(io.BytesIO(b.getvalue())).convert("RGB")).astype(np.float64)
    psnr = lambda a: 10 * np.log10(255 ** 2 / ((a - img) ** 2).mean())
    assert psnr(dec) > psnr(pil) - 0.3, (psnr(dec), psnr(pil))       # same quality as libjpeg's own encoder at this setting
    assert 0.8 < len(data) / len(b.getvalue()) < 1.35                # and about the same size (libjpeg has no restart markers)
    # restart interval = one MCU row, markers cycle RST0..RST7
    scan = data[len(J.frame_header(w, h, q)):-2]
    marks = [scan[i + 1] for i in range(len(scan) - 1) if scan[i] == 0xFF and 0xD0 <= scan[i + 1] <= 0xD7]
    assert marks == [0xD0 + (r & 7) for r in range((h + 15) // 16 - 1)]


def test_mjpeg_avi_container(pkg, tmp_path):
    """The writer's container around pre-encoded frames (here: the oracle's) is read back by the AVI reader through PIL."""
    J, vio = pkg.jpeg, pkg.video_io
    imgs = [_test_image(32, 48, seed=s) for s in range(3)]
    div, _ = J.device_tables(90)
    scans = [np.frombuffer(oj.encode_scan(im, div, J.ZIGZAG, _huff(J)), dtype=np.uint8) for im in imgs]
    packed = np.zeros((3, max(len(s) for s in scans)), np.uint8)
    for i, s in enumerate(scans):
        packed[i, :len(s)] = s
    w = vio.open_writer(str(tmp_path / "m.avi"), 48, 32, 24.0, mjpeg_quality=90)
    assert isinstance(w, vio.MJPEGWriter) and w.device_encodes
    w.write_encoded(packed[:2], np.array([len(scans[0]), len(scans[1])]))
    w.write_encoded(packed[2:], np.array([len(scans[2])]))
    w.close()
    data = open(tmp_path / "m.avi", "rb").read()
    assert struct.unpack("<I", data[4:8])[0] == len(data) - 8
    r = vio.open_reader(str(tmp_path / "m.avi"))
    assert r.mjpg and (r.total_frames, r.width, r.height, r.fps) == (3, 48, 32, 24.0)
    buf = np.empty((3, r.layout.frame_bytes), np.uint8)
    assert r.read_into(buf, 3) == 3
    for i, im in enumerate(imgs):
        got = buf[i].reshape(32, 48, 3).astype(np.float64)
        assert 10 * np.log10(255 ** 2 / ((got - im) ** 2).mean()) > 27
    d = vio.open_writer(str(tmp_path / "jpgs"), 48, 32, 24.0, base_name="f", mjpeg_quality=90)
    d.write_encoded(packed[:1], np.array([len(scans[0])]))
    d.close()
    assert Image.open(tmp_path / "jpgs" / "f_000000.jpg").size == (48, 32)
    with pytest.raises(vio.VideoIOError):
        w2 = vio.open_writer(str(tmp_path / "bad.avi"), 48, 32, 24.0, mjpeg_quality=90)
        w2.write_encoded(packed[:1], np.array([-1]))
