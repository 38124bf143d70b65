"""Input side of the boundary: folder listing, pairing and grayscale decode.

Mirrors PIVDataset / ToTensor (PIVbackend.py:103-144) and natural_keys
(PlotterFunctions.py:27-37).  The reference decodes with
cv2.imdecode(np.fromfile(path), IMREAD_GRAYSCALE); OpenCV is not a dependency here, so
8/24/32-bit uncompressed BMP (what PIV cameras write, and the reference's test_images
format) is decoded by hand with OpenCV's BGR->gray fixed-point weights, and other
formats are opened with Pillow and converted with OpenCV's rules (16-bit >> 8, fixed-point BGR2GRAY).
"""
from __future__ import annotations

import os
import re
import struct

import numpy as np
import torch


def atoi(text):
    return int(text) if text.isdigit() else text


def natural_keys(text):
    """Sort key for 'human' ordering: image9 < image10 (PlotterFunctions.py:31-37)."""
    return [atoi(c) for c in re.split(r"(\d+)", text)]


def _bgr_to_gray(b, g, r):
    # OpenCV cvtColor BGR2GRAY, 8-bit: (B*1868 + G*9617 + R*4899 + 8192) >> 14
    return ((b.astype(np.int32) * 1868 + g.astype(np.int32) * 9617 + r.astype(np.int32) * 4899 + 8192)
            >> 14).astype(np.uint8)


def decode_bmp_gray(buf: bytes):
    """Uncompressed 8/24/32-bit BMP -> uint8 [H, W] grayscale; None if not such a file."""
    if len(buf) < 54 or buf[:2] != b"BM":
        return None
    data_off = struct.unpack_from("<I", buf, 10)[0]
    hdr = struct.unpack_from("<I", buf, 14)[0]
    if hdr < 40:
        return None
    w, h, planes, bpp, comp = struct.unpack_from("<iiHHI", buf, 18)
    ncol = struct.unpack_from("<I", buf, 46)[0]
    if comp not in (0, 3) or bpp not in (8, 24, 32) or w <= 0 or h == 0:
        return None
    flip = h > 0
    h = abs(h)
    row = ((w * bpp + 31) // 32) * 4
    if len(buf) < data_off + row * h:
        return None
    raw = np.frombuffer(buf, dtype=np.uint8, count=row * h, offset=data_off).reshape(h, row)
    if bpp == 8:
        n = ncol if ncol else 256
        pal = np.frombuffer(buf, dtype=np.uint8, count=n * 4, offset=14 + hdr).reshape(n, 4)
        lut = np.zeros(256, dtype=np.uint8)
        lut[:n] = _bgr_to_gray(pal[:, 0], pal[:, 1], pal[:, 2])
        if n == 256 and np.array_equal(lut, np.arange(256, dtype=np.uint8)):
            img = raw[:, :w]            # grey ramp palette (what cameras write): the index IS the value
        else:
            img = lut[raw[:, :w]]
    else:
        px = raw[:, : w * (bpp // 8)].reshape(h, w, bpp // 8)
        img = _bgr_to_gray(px[..., 0], px[..., 1], px[..., 2])
    if flip:
        img = img[::-1]
    return np.ascontiguousarray(img)


def imdecode_gray(path: str):
    """Grayscale uint8 [H, W] image or None when the file cannot be decoded
    (the reference then skips the pair, PIVbackend.py:138-139)."""
    try:
        with open(path, "rb") as f:
            buf = f.read()
    except OSError:
        return None
    img = decode_bmp_gray(buf)
    if img is not None:
        return img
    try:
        import io as _io

        from PIL import Image
        with Image.open(_io.BytesIO(buf)) as im:
            # cv2.imdecode(..., IMREAD_GRAYSCALE) semantics rather than Pillow's "L" conversion:
            # 16-bit samples are scaled down (>> 8; Pillow would clip at 255), colour goes through
            # OpenCV's fixed-point BGR2GRAY weights (Pillow's ITU-R 601 integers differ by one grey
            # level here and there)
            if im.mode in ("I;16", "I;16L", "I;16B", "I;16N"):
                return (np.asarray(im).astype(np.uint16) >> 8).astype(np.uint8)
            if im.mode == "L":
                return np.array(im, dtype=np.uint8)
            if im.mode in ("RGB", "RGBA", "P", "CMYK", "YCbCr"):
                rgb = np.asarray(im.convert("RGB"))
                return _bgr_to_gray(rgb[..., 2], rgb[..., 1], rgb[..., 0])
            return np.array(im.convert("L"), dtype=np.uint8)
    except Exception:
        return None


def decode_into(path: str, out: np.ndarray) -> bool:
    """Decode `path` straight into a preallocated uint8 [H, W] buffer (e.g. a view of pinned
    staging memory).  False if the file cannot be decoded or has another shape."""
    img = imdecode_gray(path)
    if img is None or img.shape != out.shape:
        return False
    np.copyto(out, img)
    return True


class ToTensor:
    """numpy array -> torch.Tensor of a fixed dtype (PIVbackend.py:103-112)."""

    def __init__(self, dtype) -> None:
        self.dtype = dtype

    def __call__(self, data):
        if data is None:
            return None
        # (the reference calls torch.tensor(data, dtype=...), a 19 ms element-wise copy for a 4 MP
        #  frame; from_numpy shares the decoded buffer instead)
        t = torch.from_numpy(np.ascontiguousarray(data))
        return t if t.dtype == self.dtype else t.to(self.dtype)


class PIVDataset(torch.utils.data.Dataset):
    """Image pairs of a folder (PIVbackend.py:114-144): names ending in `file_fmt`, natural
    sort, 'pairs' = (0,1),(2,3),...; 'sequential' = (0,1),(1,2),...; anything else = empty."""

    def __init__(self, folder, file_fmt, folder_mode, transform=None):
        self.transform = transform
        filenames = [os.path.join(folder, name) for name in os.listdir(folder) if name.endswith(file_fmt)]
        filenames.sort(key=natural_keys)
        if folder_mode == "pairs":
            self.img_pairs = list(zip(filenames[::2], filenames[1::2]))
        elif folder_mode == "sequential":
            self.img_pairs = list(zip(filenames[:-1], filenames[1:]))
        else:
            self.img_pairs = []

    def __len__(self):
        return len(self.img_pairs)

    def __getitem__(self, index):
        if torch.is_tensor(index):
            index = index.tolist()
        pair = self.img_pairs[index]
        frame_b = imdecode_gray(pair[-1])
        frame_a = imdecode_gray(pair[0])
        if frame_a is None or frame_b is None:
            return None, None
        if self.transform:
            return self.transform(frame_a), self.transform(frame_b)
        return frame_a, frame_b
