"""Streaming frame I/O for the CLI: container bytes <-> device frames.

Replaces the reference's cv2 read / numpy float conversion / `mp4v` write chain
[ref: inference_cli.py:527-548 (_read_frames_from_cap), :566-654 (_stream_video_chunks), :666-749 (writers)]:

* the host only moves the container's own 8-bit pixels: a reader thread fills pinned buffers one chunk ahead, the
  upload is an async copy, and `svr2_u8_to_frames` / `svr2_yuv_to_frames` produce the bf16 [T,H,W,3] frames on the
  device with the reference's rounding chain (uint8 -> /255 fp32 -> fp16 -> bf16);
* finished frames are quantised on the device (`svr2_frames_to_u8` / `svr2_frames_to_yuv`: (x*255) truncated, the
  container's channel order, row order and padding), copied into pinned buffers and written by a writer thread while
  the next chunk is computed.

Containers handled natively (no codec library in this image): YUV4MPEG2 (.y4m, 4:2:0 / 4:4:4, any size, pipes into any
encoder), AVI with uncompressed BGR24 (.avi, < 4 GiB) or MJPEG frames (read only, through PIL), PNG/JPEG files and
directories (PIL), .npy / .pt tensors.  .mp4/.mov/.mkv/... go through cv2 when it is importable (the reference's own
backend, 'mp4v' on write) or through an `ffmpeg` binary on PATH (yuv4mpegpipe); with neither they fail loudly.
"""

from __future__ import annotations

import os
import queue
import shutil
import struct
import subprocess
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Iterator, List, Optional, Tuple

import numpy as np
import torch

from . import hip

VIDEO_EXTENSIONS = {'.mp4', '.avi', '.mov', '.mkv', '.webm', '.flv', '.wmv', '.m4v', '.y4m'}   # [ref: inference_cli.py:204]
IMAGE_EXTENSIONS = {'.png', '.jpg', '.jpeg', '.bmp', '.tiff', '.tif', '.webp'}                 # [ref: inference_cli.py:205]
TENSOR_EXTENSIONS = {'.npy', '.pt'}


class VideoIOError(RuntimeError):
    pass


class Layout:
    """How one frame lies in a container's byte stream."""

    def __init__(self, kind: str, width: int, height: int, sub: int = 1, flags: int = 0, row_stride: int = 0):
        self.kind, self.width, self.height, self.sub, self.flags = kind, width, height, sub, flags
        self.channels = 4 if (kind == "rgb" and flags & hip.PIX_ALPHA) else 3
        if kind == "rgb":
            self.row_stride = row_stride or width * self.channels
            self.frame_bytes = self.row_stride * height
        elif kind == "yuv":
            cw, ch = (width + sub - 1) // sub, (height + sub - 1) // sub
            self.row_stride = width
            self.frame_bytes = width * height + 2 * cw * ch
        else:
            raise ValueError(kind)

    def __repr__(self):
        return f"Layout({self.kind}, {self.width}x{self.height}, sub={self.sub}, flags={self.flags}, stride={self.row_stride})"


# =============================================================================================== device conversions
def frames_from_bytes(raw: torch.Tensor, lay: Layout) -> torch.Tensor:
    """raw: uint8 device tensor [T, frame_bytes] in the container layout -> bf16 [T,H,W,3] in [0,1]."""
    T = raw.shape[0]
    out = torch.empty(T, lay.height, lay.width, lay.channels, dtype=torch.bfloat16, device=raw.device)
    if lay.kind == "rgb":
        hip.u8_to_frames(raw, out, lay.row_stride, lay.frame_bytes, lay.flags)
    else:
        hip.yuv_to_frames(raw, out, lay.sub, lay.frame_bytes)
    return out


def frames_to_bytes(frames: torch.Tensor, lay: Layout, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16 device frames [T,H,W,3] in [0,1] -> uint8 device tensor [T, frame_bytes] in the container layout."""
    T = frames.shape[0]
    if tuple(frames.shape[1:]) != (lay.height, lay.width, lay.channels):
        raise VideoIOError(f"writer expects {lay.width}x{lay.height}x{lay.channels} frames, got {tuple(frames.shape)}"
                           + (" (this container has no alpha channel: write .png)" if frames.shape[-1] == 4 else ""))
    if out is None:
        out = torch.empty(T, lay.frame_bytes, dtype=torch.uint8, device=frames.device)
    if lay.kind == "rgb":
        hip.frames_to_u8(frames, out, lay.row_stride, lay.frame_bytes, lay.flags)
    else:
        hip.frames_to_yuv(frames, out, lay.sub, lay.frame_bytes)
    return out


# =============================================================================================== readers
class Reader:
    """fps, total_frames, layout; seek(frame); read_into(buf[T, frame_bytes] uint8, n) -> frames read."""
    fps: float = 30.0
    total_frames: int = 0
    layout: Layout

    def seek(self, frame: int) -> None:
        raise NotImplementedError

    def read_into(self, buf: np.ndarray, n: int) -> int:
        raise NotImplementedError

    def close(self) -> None:
        pass

    @property
    def width(self):
        return self.layout.width

    @property
    def height(self):
        return self.layout.height


class Y4MReader(Reader):
    _SUB = {"420jpeg": 2, "420mpeg2": 2, "420paldv": 2, "420": 2, "444": 1}

    def __init__(self, path_or_file):
        self.f = open(path_or_file, "rb") if isinstance(path_or_file, (str, os.PathLike)) else path_or_file
        self.seekable = isinstance(path_or_file, (str, os.PathLike))
        head = self.f.readline(256)
        if not head.startswith(b"YUV4MPEG2 ") or not head.endswith(b"\n"):
            raise VideoIOError("not a YUV4MPEG2 stream")
        w = h = 0
        cs, fps = "420jpeg", 30.0
        for tok in head[10:].split():
            k, v = chr(tok[0]), tok[1:].decode("ascii")
            if k == "W":
                w = int(v)
            elif k == "H":
                h = int(v)
            elif k == "F":
                num, den = v.split(":")
                fps = int(num) / max(int(den), 1)
            elif k == "C":
                cs = v
        if cs not in self._SUB:
            raise VideoIOError(f"y4m colourspace C{cs} is not supported (8-bit 4:2:0 and 4:4:4 only)")
        if w <= 0 or h <= 0:
            raise VideoIOError("y4m header lacks W/H")
        self.fps, self.layout = fps or 30.0, Layout("yuv", w, h, sub=self._SUB[cs])
        self._data0 = len(head)
        self._pos = 0
        self.total_frames = 0
        if self.seekable:
            first = self.f.readline(256)
            self._fh = len(first) if first.startswith(b"FRAME") else 6
            size = os.fstat(self.f.fileno()).st_size
            self.total_frames = max(size - self._data0, 0) // (self._fh + self.layout.frame_bytes)
            self.f.seek(self._data0)

    def seek(self, frame: int) -> None:
        if not self.seekable:
            raise VideoIOError("cannot seek in a y4m pipe")
        self.f.seek(self._data0 + frame * (self._fh + self.layout.frame_bytes))
        self._pos = frame

    def read_into(self, buf: np.ndarray, n: int) -> int:
        got = 0
        for i in range(n):
            line = self.f.readline(256)
            if not line:
                break
            if not line.startswith(b"FRAME"):
                raise VideoIOError("y4m: FRAME marker expected")
            if self.f.readinto(memoryview(buf[i])) != self.layout.frame_bytes:
                break
            got += 1
        self._pos += got
        return got

    def close(self):
        self.f.close()


def _riff_chunks(f, start: int, end: int) -> Iterator[Tuple[bytes, int, int]]:
    pos = start
    while pos + 8 <= end:
        f.seek(pos)
        hdr = f.read(8)
        if len(hdr) < 8:
            return
        cid, size = hdr[:4], struct.unpack("<I", hdr[4:])[0]
        yield cid, pos + 8, size
        pos += 8 + size + (size & 1)


class AVIReader(Reader):
    """AVI 1.0, stream 0 = video, uncompressed 24-bit DIB (BI_RGB) or MJPG frames."""

    def __init__(self, path: str):
        self.f = open(path, "rb")
        size = os.fstat(self.f.fileno()).st_size
        head = self.f.read(12)
        if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"AVI ":
            raise VideoIOError(f"{path}: not a RIFF AVI file")
        self.frames: List[Tuple[int, int]] = []
        self.fps, w, h, bits, comp = 30.0, 0, 0, 24, 0
        riff_end = min(size, 8 + struct.unpack("<I", head[4:8])[0])
        for cid, off, sz in _riff_chunks(self.f, 12, riff_end):
            if cid != b"LIST":
                continue
            self.f.seek(off)
            ltype = self.f.read(4)
            if ltype == b"hdrl":
                for c2, o2, s2 in _riff_chunks(self.f, off + 4, off + sz):
                    if c2 == b"avih":
                        self.f.seek(o2)
                        us = struct.unpack("<I", self.f.read(4))[0]
                        if us:
                            self.fps = 1e6 / us
                    elif c2 == b"LIST":
                        self.f.seek(o2)
                        if self.f.read(4) != b"strl" or w:
                            continue
                        vids = False
                        for c3, o3, s3 in _riff_chunks(self.f, o2 + 4, o2 + s2):
                            self.f.seek(o3)
                            if c3 == b"strh":
                                d = self.f.read(min(s3, 56))
                                vids = d[:4] == b"vids"
                                scale, rate = struct.unpack("<II", d[20:28])
                                if vids and scale and rate:
                                    self.fps = rate / scale
                            elif c3 == b"strf" and vids:
                                d = self.f.read(40)
                                _, w, h, _, bits, comp = struct.unpack("<IiiHHI", d[:20])
            elif ltype == b"movi":
                for c2, o2, s2 in _riff_chunks(self.f, off + 4, off + sz):
                    if c2 in (b"00db", b"00dc") and s2 > 0:
                        self.frames.append((o2, s2))
        if not w:
            raise VideoIOError(f"{path}: no video stream header")
        self.mjpg = comp == struct.unpack("<I", b"MJPG")[0]
        if not self.mjpg and not (comp == 0 and bits == 24):
            raise VideoIOError(f"{path}: AVI codec {struct.pack('<I', comp)!r}/{bits} bit is not supported here "
                               "(uncompressed BGR24 or MJPG; use cv2/ffmpeg for others)")
        self.total_frames = len(self.frames)
        if self.mjpg:
            self.layout = Layout("rgb", w, abs(h))
        else:  # DIB rows are padded to 4 bytes; positive height = bottom-up
            self.layout = Layout("rgb", w, abs(h), flags=hip.PIX_SWAP_RB | (hip.PIX_BOTTOM_UP if h > 0 else 0),
                                 row_stride=(w * 3 + 3) & ~3)
        self._pos = 0

    def seek(self, frame: int) -> None:
        self._pos = frame

    def read_into(self, buf: np.ndarray, n: int) -> int:
        got = 0
        while got < n and self._pos < self.total_frames:
            off, sz = self.frames[self._pos]
            self.f.seek(off)
            if self.mjpg:
                import io
                from PIL import Image
                img = np.asarray(Image.open(io.BytesIO(self.f.read(sz))).convert("RGB"), dtype=np.uint8)
                buf[got] = img.reshape(-1)
            else:
                if sz < self.layout.frame_bytes or self.f.readinto(memoryview(buf[got])) != self.layout.frame_bytes:
                    raise VideoIOError("AVI: short frame chunk")
            got += 1
            self._pos += 1
        return got

    def close(self):
        self.f.close()


class ArrayReader(Reader):
    """uint8 [N,H,W,3] array (np.load(mmap) of a .npy, or frames decoded from image files)."""

    def __init__(self, arr: np.ndarray, fps: float = 30.0):
        if arr.ndim != 4 or arr.shape[3] != 3 or arr.dtype != np.uint8:
            raise VideoIOError(f"expected a uint8 [N,H,W,3] array, got {arr.dtype} {arr.shape}")
        self.arr, self.fps, self.total_frames = arr, fps, arr.shape[0]
        self.layout = Layout("rgb", arr.shape[2], arr.shape[1])
        self._pos = 0

    def seek(self, frame: int) -> None:
        self._pos = frame

    def read_into(self, buf: np.ndarray, n: int) -> int:
        n = max(min(n, self.total_frames - self._pos), 0)
        buf[:n] = self.arr[self._pos:self._pos + n].reshape(n, -1)
        self._pos += n
        return n


class ImageFilesReader(Reader):
    """One image or a list of same-sized image files, decoded by PIL on read."""

    def __init__(self, paths: List[str], fps: float = 30.0):
        from PIL import Image
        self.paths, self.fps, self.total_frames = paths, fps, len(paths)
        with Image.open(paths[0]) as im:
            # alpha is preserved like cv2.IMREAD_UNCHANGED does [ref: inference_cli.py:254-266]
            self.has_alpha = im.mode in ("RGBA", "LA") or "transparency" in im.info
            self.layout = Layout("rgb", im.width, im.height, flags=hip.PIX_ALPHA if self.has_alpha else 0)
        self._pos = 0

    def seek(self, frame: int) -> None:
        self._pos = frame

    def read_into(self, buf: np.ndarray, n: int) -> int:
        from PIL import Image
        got = 0
        while got < n and self._pos < self.total_frames:
            with Image.open(self.paths[self._pos]) as im:
                a = np.asarray(im.convert("RGBA" if self.has_alpha else "RGB"), dtype=np.uint8)
            if a.shape[:2] != (self.layout.height, self.layout.width):
                raise VideoIOError(f"{self.paths[self._pos]}: size differs from the first frame")
            buf[got] = a.reshape(-1)
            got += 1
            self._pos += 1
        return got


class Cv2Reader(Reader):  # pragma: no cover  (cv2 is not in this image)
    def __init__(self, path: str):
        import cv2
        self.cv2, self.cap = cv2, cv2.VideoCapture(path)
        if not self.cap.isOpened():
            raise VideoIOError(f"Cannot open video file: {path}")
        self.fps = self.cap.get(cv2.CAP_PROP_FPS) or 30.0
        self.total_frames = int(self.cap.get(cv2.CAP_PROP_FRAME_COUNT))
        self.layout = Layout("rgb", int(self.cap.get(cv2.CAP_PROP_FRAME_WIDTH)), int(self.cap.get(cv2.CAP_PROP_FRAME_HEIGHT)),
                             flags=hip.PIX_SWAP_RB)   # cv2 hands out BGR; the swap happens on the device

    def seek(self, frame: int) -> None:
        self.cap.set(self.cv2.CAP_PROP_POS_FRAMES, frame)

    def read_into(self, buf: np.ndarray, n: int) -> int:
        got = 0
        for i in range(n):
            ok, fr = self.cap.read()
            if not ok:
                break
            buf[i] = fr.reshape(-1)
            got += 1
        return got

    def close(self):
        self.cap.release()


class FfmpegReader(Y4MReader):  # pragma: no cover  (no ffmpeg binary in this image)
    def __init__(self, path: str, start_frame: int = 0):
        self.proc = subprocess.Popen(["ffmpeg", "-v", "error", "-i", path, "-f", "yuv4mpegpipe", "-pix_fmt", "yuv420p", "-"],
                                     stdout=subprocess.PIPE)
        super().__init__(self.proc.stdout)

    def seek(self, frame: int) -> None:
        scratch = np.empty((1, self.layout.frame_bytes), dtype=np.uint8)
        for _ in range(frame - self._pos):
            if self.read_into(scratch, 1) != 1:
                break

    def close(self):
        self.proc.stdout.close()
        self.proc.wait()


def list_media_files(directory: str) -> List[str]:
    """[ref: inference_cli.py:212-229]"""
    ok = VIDEO_EXTENSIONS | IMAGE_EXTENSIONS
    return sorted(os.path.join(directory, f) for f in os.listdir(directory)
                  if os.path.isfile(os.path.join(directory, f)) and os.path.splitext(f)[1].lower() in ok)


def open_reader(path: str) -> Reader:
    if not os.path.exists(path):
        raise FileNotFoundError(f"Input path not found: {path}")
    if os.path.isdir(path):
        files = [p for p in list_media_files(path) if os.path.splitext(p)[1].lower() in IMAGE_EXTENSIONS]
        if not files:
            raise VideoIOError(f"{path}: no image files")
        return ImageFilesReader(files)
    ext = os.path.splitext(path)[1].lower()
    if ext == ".y4m":
        return Y4MReader(path)
    if ext == ".avi":
        try:
            return AVIReader(path)
        except VideoIOError:
            if not _have_cv2() and not shutil.which("ffmpeg"):
                raise
    if ext in IMAGE_EXTENSIONS:
        return ImageFilesReader([path])
    if ext == ".npy":
        return ArrayReader(_as_u8(np.load(path, mmap_mode="r", allow_pickle=False)))
    if ext == ".pt":
        return ArrayReader(_as_u8(torch.load(path, weights_only=True).numpy()))
    if ext in VIDEO_EXTENSIONS:
        if _have_cv2():
            return Cv2Reader(path)
        if shutil.which("ffmpeg"):
            return FfmpegReader(path)
        raise VideoIOError(f"{ext} needs cv2 or an ffmpeg binary, neither is present: convert to .y4m / uncompressed "
                           ".avi / an image directory")
    raise VideoIOError(f"unsupported input {ext!r}")


def _as_u8(a: np.ndarray) -> np.ndarray:
    """Tensor files hold [N,H,W,3] uint8, or floats in [0,1] (quantised with rounding: an exact inverse of /255)."""
    if a.dtype == np.uint8:
        return a
    return np.clip(np.rint(np.asarray(a, dtype=np.float32) * 255.0), 0, 255).astype(np.uint8)


def _have_cv2() -> bool:
    try:
        import cv2  # noqa: F401
        return True
    except Exception:
        return False


# =============================================================================================== writers
class Writer:
    """layout; write_raw(buf[T, frame_bytes] uint8 host array); close()."""
    layout: Layout

    def write_raw(self, buf: np.ndarray) -> None:
        raise NotImplementedError

    def close(self) -> None:
        pass


class Y4MWriter(Writer):
    def __init__(self, path_or_file, width: int, height: int, fps: float, sub: int = 2):
        if sub == 2 and (width % 2 or height % 2):
            sub = 1
        self.layout = Layout("yuv", width, height, sub=sub)
        if isinstance(path_or_file, (str, os.PathLike)):
            os.makedirs(os.path.dirname(os.path.abspath(path_or_file)), exist_ok=True)
            self.f = open(path_or_file, "wb")
        else:
            self.f = path_or_file
        num, den = _fps_fraction(fps)
        self.f.write(f"YUV4MPEG2 W{width} H{height} F{num}:{den} Ip A1:1 C{'420jpeg' if sub == 2 else '444'}\n".encode())

    def write_raw(self, buf: np.ndarray) -> None:
        for fr in buf:
            self.f.write(b"FRAME\n")
            self.f.write(memoryview(fr))

    def close(self):
        self.f.close()


def _fps_fraction(fps: float) -> Tuple[int, int]:
    for den in (1, 1001):
        num = round(fps * den)
        if abs(num / den - fps) < 1e-3:
            return int(num), den
    return int(round(fps * 1000)), 1000


class AVIWriter(Writer):
    """AVI 1.0 with uncompressed bottom-up BGR24 frames and an idx1 index (RIFF limit: 4 GiB)."""
    LIMIT = (1 << 32) - (1 << 20)

    def __init__(self, path: str, width: int, height: int, fps: float):
        self.layout = Layout("rgb", width, height, flags=hip.PIX_SWAP_RB | hip.PIX_BOTTOM_UP, row_stride=(width * 3 + 3) & ~3)
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        self.f, self.fps, self.n = open(path, "wb"), fps, 0
        self.f.write(self._headers(0))
        self._movi = self.f.tell() - 4      # position of the 'movi' fourcc
        self._index: List[int] = []

    def _headers(self, nframes: int) -> bytes:
        lay = self.layout
        num, den = _fps_fraction(self.fps)
        strf = struct.pack("<IiiHHIIiiII", 40, lay.width, lay.height, 1, 24, 0, lay.frame_bytes, 0, 0, 0, 0)
        strh = struct.pack("<4s4sIHHIIIIIIIIhhhh", b"vids", b"DIB ", 0, 0, 0, 0, den, num, 0, nframes, lay.frame_bytes,
                           0xFFFFFFFF, 0, 0, 0, lay.width, lay.height)
        strl = b"strl" + b"strh" + struct.pack("<I", len(strh)) + strh + b"strf" + struct.pack("<I", len(strf)) + strf
        avih = struct.pack("<IIIIIIIIII4I", int(round(1e6 * den / num)), int(lay.frame_bytes * num / den), 0, 0x10, nframes, 0, 1,
                           lay.frame_bytes, lay.width, lay.height, 0, 0, 0, 0)
        hdrl = b"hdrl" + b"avih" + struct.pack("<I", len(avih)) + avih + b"LIST" + struct.pack("<I", len(strl)) + strl
        movi_sz = 4 + nframes * (8 + lay.frame_bytes + (lay.frame_bytes & 1))
        idx_sz = 16 * nframes
        riff_sz = 4 + 8 + len(hdrl) + 8 + movi_sz + (8 + idx_sz if nframes else 0)
        return (b"RIFF" + struct.pack("<I", riff_sz) + b"AVI " + b"LIST" + struct.pack("<I", len(hdrl)) + hdrl +
                b"LIST" + struct.pack("<I", movi_sz) + b"movi")

    def write_raw(self, buf: np.ndarray) -> None:
        fb = self.layout.frame_bytes
        for fr in buf:
            if self.f.tell() + fb + 8 + 16 * (self.n + 1) + 8 > self.LIMIT:
                raise VideoIOError("AVI 1.0 is limited to 4 GiB: write .y4m (or a PNG directory) for longer clips")
            self._index.append(self.f.tell() - self._movi)
            self.f.write(b"00db" + struct.pack("<I", fb))
            self.f.write(memoryview(fr))
            if fb & 1:
                self.f.write(b"\0")
            self.n += 1

    def close(self):
        fb = self.layout.frame_bytes
        if self.n:
            self.f.write(b"idx1" + struct.pack("<I", 16 * self.n))
            for off in self._index:
                self.f.write(struct.pack("<4sIII", b"00db", 0x10, off, fb))
        self.f.seek(0)
        self.f.write(self._headers(self.n))
        self.f.close()


class MJPEGWriter(Writer):
    """Motion-JPEG: AVI ('MJPG' stream, one JPEG per '00dc' chunk, idx1 index), or a directory of .jpg files.  The frames
    are encoded ON THE DEVICE (hip.jpeg_encode -> csrc/jpeg.hip); this class only adds the constant header and EOI and
    writes bytes.  ~10-20x smaller than raw frames, so AVI's 4 GiB limit holds thousands of 4K frames."""
    device_encodes = True
    LIMIT = (1 << 32) - (1 << 20)

    def __init__(self, path: str, width: int, height: int, fps: float, quality: int = 90, base_name: str = "frame"):
        from . import jpeg as _jpeg
        self.layout = Layout("rgb", width, height)
        self.quality, self.fps, self.n = int(quality), fps, 0
        self.header = _jpeg.frame_header(width, height, self.quality)
        self.avi = os.path.splitext(path)[1].lower() == ".avi"
        self.path, self.base = path, base_name
        self._index: List[Tuple[int, int]] = []
        if self.avi:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            self.f = open(path, "wb")
            self.f.write(self._headers(0, 0, 0))
            self._movi = self.f.tell() - 4
        else:
            os.makedirs(path, exist_ok=True)

    def device_encode(self, frames: torch.Tensor):
        return hip.jpeg_encode(frames, self.quality)

    def _headers(self, nframes: int, movi_bytes: int, max_chunk: int) -> bytes:
        lay = self.layout
        num, den = _fps_fraction(self.fps)
        mjpg = struct.unpack("<I", b"MJPG")[0]
        strf = struct.pack("<IiiHHIIiiII", 40, lay.width, lay.height, 1, 24, mjpg, lay.width * lay.height * 3, 0, 0, 0, 0)
        strh = struct.pack("<4s4sIHHIIIIIIIIhhhh", b"vids", b"MJPG", 0, 0, 0, 0, den, num, 0, nframes, max_chunk, 0xFFFFFFFF, 0, 0,
                           0, lay.width, lay.height)
        strl = b"strl" + b"strh" + struct.pack("<I", len(strh)) + strh + b"strf" + struct.pack("<I", len(strf)) + strf
        avih = struct.pack("<IIIIIIIIII4I", int(round(1e6 * den / num)), int(max_chunk * num / den), 0, 0x10, nframes, 0, 1,
                           max_chunk, lay.width, lay.height, 0, 0, 0, 0)
        hdrl = b"hdrl" + b"avih" + struct.pack("<I", len(avih)) + avih + b"LIST" + struct.pack("<I", len(strl)) + strl
        riff_sz = 4 + 8 + len(hdrl) + 8 + 4 + movi_bytes + (8 + 16 * nframes if nframes else 0)
        return (b"RIFF" + struct.pack("<I", riff_sz) + b"AVI " + b"LIST" + struct.pack("<I", len(hdrl)) + hdrl +
                b"LIST" + struct.pack("<I", 4 + movi_bytes) + b"movi")

    def write_encoded(self, packed: np.ndarray, lens: np.ndarray) -> None:
        for row, n in zip(packed, lens):
            if n < 0:
                raise VideoIOError("JPEG: a frame did not fit its buffer (noise-like content at very high quality)")
            data = self.header + row[:n].tobytes() + b"\xff\xd9"
            if self.avi:
                if self.f.tell() + len(data) + 8 + 16 * (self.n + 1) + 8 > self.LIMIT:
                    raise VideoIOError("AVI 1.0 is limited to 4 GiB: write .y4m or a directory for longer clips")
                self._index.append((self.f.tell() - self._movi, len(data)))
                self.f.write(b"00dc" + struct.pack("<I", len(data)) + data + (b"\0" if len(data) & 1 else b""))
            else:
                with open(os.path.join(self.path, f"{self.base}_{self.n:06d}.jpg"), "wb") as fh:
                    fh.write(data)
            self.n += 1

    def close(self):
        if not self.avi:
            return
        movi_bytes = self.f.tell() - self._movi - 4
        if self.n:
            self.f.write(b"idx1" + struct.pack("<I", 16 * self.n))
            for off, size in self._index:
                self.f.write(struct.pack("<4sIII", b"00dc", 0x10, off, size))
        self.f.seek(0)
        self.f.write(self._headers(self.n, movi_bytes, max((s for _, s in self._index), default=0)))
        self.f.close()


class PngSequenceWriter(Writer):
    """{base}_{index:06d}.png files [ref: inference_cli.py:709-749]; PNG compression runs on a thread pool."""

    def __init__(self, directory: str, base_name: str, width: int, height: int, start_index: int = 0, workers: int = 8,
                 alpha: bool = False):
        self.layout = Layout("rgb", width, height, flags=hip.PIX_ALPHA if alpha else 0)
        os.makedirs(directory, exist_ok=True)
        self.dir, self.base, self.idx = directory, base_name, start_index
        self.pool, self.pending = ThreadPoolExecutor(max_workers=workers), []

    def _save(self, arr: np.ndarray, path: str):
        from PIL import Image
        Image.fromarray(arr).save(path, compress_level=3)

    def write_raw(self, buf: np.ndarray) -> None:
        lay = self.layout
        for fr in buf:
            path = os.path.join(self.dir, f"{self.base}_{self.idx:06d}.png")
            self.pending.append(self.pool.submit(self._save, fr.reshape(lay.height, lay.width, lay.channels).copy(), path))
            self.idx += 1
        while len(self.pending) > 32:
            self.pending.pop(0).result()

    def close(self):
        for p in self.pending:
            p.result()
        self.pool.shutdown()


class ImageWriter(Writer):
    """A single image file (PIL picks the codec from the extension)."""

    def __init__(self, path: str, width: int, height: int, alpha: bool = False):
        self.layout, self.path = Layout("rgb", width, height, flags=hip.PIX_ALPHA if alpha else 0), path
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)

    def write_raw(self, buf: np.ndarray) -> None:
        from PIL import Image
        Image.fromarray(buf[0].reshape(self.layout.height, self.layout.width, self.layout.channels)).save(self.path)


class NpyWriter(Writer):
    """uint8 [N,H,W,3] .npy written incrementally (header patched on close)."""

    def __init__(self, path: str, width: int, height: int):
        self.layout, self.path, self.n = Layout("rgb", width, height), path, 0
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        self.f = open(path, "wb")
        self._header()

    def _header(self):
        d = f"{{'descr': '|u1', 'fortran_order': False, 'shape': ({self.n}, {self.layout.height}, {self.layout.width}, 3), }}"
        d = d.ljust(117) + "\n"                       # fixed 128-byte header: magic(6)+ver(2)+len(2)+118
        self.f.seek(0)
        self.f.write(b"\x93NUMPY\x01\x00" + struct.pack("<H", len(d)) + d.encode("latin1"))

    def write_raw(self, buf: np.ndarray) -> None:
        self.f.seek(0, 2)
        self.f.write(memoryview(np.ascontiguousarray(buf)))
        self.n += buf.shape[0]

    def close(self):
        self._header()
        self.f.close()


class Cv2Writer(Writer):  # pragma: no cover
    def __init__(self, path: str, width: int, height: int, fps: float):
        import cv2
        self.layout = Layout("rgb", width, height, flags=hip.PIX_SWAP_RB)
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        self.w = cv2.VideoWriter(path, cv2.VideoWriter_fourcc(*"mp4v"), fps, (width, height))   # [ref: inference_cli.py:697-700]
        if not self.w.isOpened():
            raise VideoIOError(f"Cannot create video writer for: {path}")

    def write_raw(self, buf: np.ndarray) -> None:
        for fr in buf:
            self.w.write(fr.reshape(self.layout.height, self.layout.width, 3))

    def close(self):
        self.w.release()


class FfmpegWriter(Y4MWriter):  # pragma: no cover
    def __init__(self, path: str, width: int, height: int, fps: float):
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        self.proc = subprocess.Popen(["ffmpeg", "-v", "error", "-y", "-f", "yuv4mpegpipe", "-i", "-", "-c:v", "mpeg4", "-q:v", "2",
                                      path], stdin=subprocess.PIPE)
        super().__init__(self.proc.stdin, width, height, fps)

    def close(self):
        self.proc.stdin.close()
        if self.proc.wait() != 0:
            raise VideoIOError("ffmpeg failed")


def open_writer(path: str, width: int, height: int, fps: float = 30.0, base_name: str = "frame", single_image: bool = False,
                alpha: bool = False, mjpeg_quality: int = 0) -> Writer:
    """mjpeg_quality > 0: .avi / directory outputs become Motion-JPEG encoded on the device (else raw BGR24 / PNG)."""
    ext = os.path.splitext(path)[1].lower()
    if mjpeg_quality > 0 and not alpha and not single_image and ext in (".avi", ""):
        return MJPEGWriter(path, width, height, fps, mjpeg_quality, base_name)
    if single_image and ext in IMAGE_EXTENSIONS:
        return ImageWriter(path, width, height, alpha)
    if ext == "":
        return PngSequenceWriter(path, base_name, width, height, alpha=alpha)
    if alpha:
        raise VideoIOError(f"{ext} has no alpha channel: write RGBA results as .png / a PNG directory")
    if ext == ".y4m":
        return Y4MWriter(path, width, height, fps)
    if ext == ".avi" and not _have_cv2():
        return AVIWriter(path, width, height, fps)
    if ext == ".npy":
        return NpyWriter(path, width, height)
    if ext in VIDEO_EXTENSIONS:
        if _have_cv2():
            return Cv2Writer(path, width, height, fps)
        if shutil.which("ffmpeg"):
            return FfmpegWriter(path, width, height, fps)
        raise VideoIOError(f"{ext} output needs cv2 or an ffmpeg binary, neither is present: use .y4m, .avi (uncompressed), "
                           "--output_format png or .npy")
    raise VideoIOError(f"unsupported output {ext!r}")


# =============================================================================================== async device <-> host
class DeviceSource:
    """Reads chunks one ahead on a thread into pinned memory; `next_chunk(n)` returns bf16 device frames."""

    def __init__(self, reader: Reader, device, pinned: bool = True):
        self.reader, self.device, self.lay = reader, torch.device(device), reader.layout
        self.pool = ThreadPoolExecutor(max_workers=1)
        self.pinned = pinned and self.device.type == "cuda"
        self._future = None
        self._bufs: List[torch.Tensor] = []

    def _buf(self, n: int) -> torch.Tensor:
        for i, b in enumerate(self._bufs):
            if b.shape[0] >= n:
                return self._bufs.pop(i)
        return torch.empty(n, self.lay.frame_bytes, dtype=torch.uint8, pin_memory=self.pinned)

    def _read(self, n: int):
        buf = self._buf(n)
        got = self.reader.read_into(buf.numpy(), n)
        return buf, got

    def prefetch(self, n: int) -> None:
        if self._future is None and n > 0:
            self._future = self.pool.submit(self._read, n)

    def next_chunk(self, n: int, prefetch_next: int = 0) -> Optional[torch.Tensor]:
        if self._future is None:
            self.prefetch(n)
        buf, got = self._future.result()
        self._future = None
        if prefetch_next > 0:
            self.prefetch(prefetch_next)
        if got == 0:
            self._bufs.append(buf)
            return None
        raw = buf[:got].to(self.device, non_blocking=True)
        frames = frames_from_bytes(raw, self.lay)
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()   # the pinned buffer is free again
        self._bufs.append(buf)
        return frames

    def close(self):
        if self._future is not None:
            self._future.result()
        self.pool.shutdown()
        self.reader.close()


class DeviceSink:
    """Quantises finished frames on the device and hands the container bytes to a writer thread."""

    def __init__(self, writer: Writer, device, depth: int = 2):
        self.writer, self.device, self.lay = writer, torch.device(device), writer.layout
        self.q: "queue.Queue" = queue.Queue(maxsize=depth)
        self.free: "queue.Queue" = queue.Queue()
        self.depth, self.allocated, self.frames_written = depth + 1, 0, 0
        self.err: Optional[BaseException] = None
        self.thread = threading.Thread(target=self._run, name="svr2-writer", daemon=True)
        self.thread.start()

    def _run(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            host, n, ev = item[:3]
            try:
                if self.err is None:
                    if ev is not None:
                        ev.synchronize()
                    if len(item) == 5:
                        self.writer.write_encoded(host[:n, :item[4]].numpy(), item[3])
                    else:
                        self.writer.write_raw(host[:n].numpy())
            except BaseException as e:  # surfaced by the next write()/close() on the caller's thread
                self.err = e
            finally:
                self.free.put(host)

    def _host(self, n: int, width: int = 0) -> torch.Tensor:
        width = width or self.lay.frame_bytes
        while True:
            if self.allocated < self.depth and self.free.empty():
                self.allocated += 1
                return torch.empty(n, width, dtype=torch.uint8, pin_memory=self.device.type == "cuda")
            b = self.free.get()
            if b.shape[0] >= n and b.shape[1] >= width:
                return b
            self.allocated -= 1          # too small: drop it and allocate a bigger one

    def write(self, frames: torch.Tensor) -> None:
        """frames: bf16 [T,H,W,3] in [0,1] on the device."""
        if self.err is not None:
            raise self.err
        if frames.shape[0] == 0:
            return
        if getattr(self.writer, "device_encodes", False):
            for i in range(0, frames.shape[0], 8):       # bounded scratch: eight frames per encode
                part = frames[i:i + 8].contiguous()
                packed, lens = self.writer.device_encode(part)
                lens_h = lens.cpu()                         # tiny read-back: how many bytes to fetch
                need = max(int(lens_h.max()), 1)
                host = self._host(part.shape[0], need)
                host[:part.shape[0], :need].copy_(packed[:, :need], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                self.q.put((host, part.shape[0], ev, lens_h.numpy(), need))
            self.frames_written += frames.shape[0]
            return
        raw = frames_to_bytes(frames.contiguous(), self.lay)
        host = self._host(frames.shape[0])
        host[:frames.shape[0]].copy_(raw, non_blocking=True)
        ev = None
        if self.device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
        self.q.put((host, frames.shape[0], ev))
        self.frames_written += frames.shape[0]

    def close(self) -> None:
        self.q.put(None)
        self.thread.join()
        self.writer.close()
        if self.err is not None:
            raise self.err


# =============================================================================================== the streaming loop
def stream_upscale(source: DeviceSource, sink_for: Callable[[int, int], DeviceSink], run: Callable[[torch.Tensor, int], torch.Tensor],
                   frames_to_process: int, chunk_size: int = 0, overlap: int = 0, prepend_frames: int = 0,
                   log: Optional[Callable[[str], None]] = None) -> int:
    """The reference's chunk streaming [ref: inference_cli.py:566-654, 440-495]: read `chunk_size` new frames (all of them
    when 0), prepend the last `overlap` RAW frames of the previous chunk as context, upscale, drop the context frames from
    the result, append to the output; `prepend_frames` applies to the first chunk only.  `run(frames, prepend) -> bf16
    [T',H',W',3]` on the device; `sink_for(width, height)` opens the output once the first result's size is known.
    Returns the number of frames written."""
    chunk = chunk_size if chunk_size > 0 else frames_to_process
    total_chunks = (frames_to_process + chunk - 1) // chunk
    read, idx, tail, sink = 0, 0, None, None
    try:
        while read < frames_to_process:
            n = min(chunk, frames_to_process - read)
            nxt = min(chunk, frames_to_process - read - n)
            new = source.next_chunk(n, prefetch_next=nxt)
            if new is None:
                break
            read += new.shape[0]
            idx += 1
            ctx = 0
            frames = new
            if tail is not None and overlap > 0:
                ctx = min(overlap, tail.shape[0])
                frames = torch.cat((tail[-ctx:], new), dim=0)
            if log and chunk < frames_to_process:
                log(f"Chunk {idx}/{total_chunks}: {new.shape[0]} new + {ctx} context frames")
            out = run(frames, prepend_frames if idx == 1 else 0)
            if ctx:
                out = out[ctx:]
            tail = new[-overlap:].clone() if overlap > 0 else None
            del frames, new
            if sink is None:
                sink = sink_for(out.shape[2], out.shape[1])
            sink.write(out)
            del out
    finally:
        if sink is not None:
            sink.close()
    return sink.frames_written if sink is not None else 0
