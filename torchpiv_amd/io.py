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


_DIGIT_RUNS = re.compile(r"(\d+)")


def natural_keys(text):
    """Sort key for 'human' ordering (image9 < image10), the order the reference lists a folder in
    (PlotterFunctions.py:27-37): the name split at its digit runs, the runs compared as numbers."""
    return [int(part) if part.isdigit() else part for part in _DIGIT_RUNS.split(text)]


def atoi(text):          # PlotterFunctions.py:27-28, kept for callers of the reference's helper name
    return int(text) if text.isdigit() else text


def _bgr_to_gray(b, g, r):
    # OpenCV cvtColor BGR2GRAY, 8-bit: (B*1868 + G*9617 + R*4899 + 8192) >> 14
    return ((b.astype(np.int32) * 1868 + g.astype(np.int32) * 9617 + r.astype(np.int32) * 4899 + 8192)
            >> 14).astype(np.uint8)


def bmp_layout(buf):
    """Header of an uncompressed 8/24/32-bit BMP -> dict(w, h, data_off, stride, bytes_pp, flip, lut) or
    None if `buf` (bytes-like, at least the header and palette) is not such a file.  `lut` (uint8 [256]) maps a
    palette index to its gray value (OpenCV's BGR2GRAY of the palette entry); None for 24/32-bit files."""
    if len(buf) < 54 or bytes(buf[:2]) != b"BM":
        return None
    data_off = struct.unpack_from("<I", buf, 10)[0]
    hdr = struct.unpack_from("<I", buf, 14)[0]
    if hdr < 40:
        return None
    w, h, planes, bpp, comp = struct.unpack_from("<iiHHI", buf, 18)
    ncol = struct.unpack_from("<I", buf, 46)[0]
    if comp not in (0, 3) or bpp not in (8, 24, 32) or w <= 0 or h == 0 or planes != 1:
        return None
    if comp == 3:
        # BI_BITFIELDS: only the standard layout of a 32-bit file (B, G, R in the low three bytes) is plain BGR(A);
        # other channel masks belong to the general decoder (imdecode_gray), which honours them
        masks_at = 14 + 40                                      # the masks follow the 40-byte header (V4 / V5: its first extra fields)
        if bpp != 32 or len(buf) < masks_at + 12 or struct.unpack_from("<III", buf, masks_at) != (0xFF0000, 0xFF00, 0xFF):
            return None
    lut = None
    if bpp == 8:
        n = ncol if ncol else 256
        if n > 256 or len(buf) < 14 + hdr + 4 * n:
            return None
        pal = np.frombuffer(buf, dtype=np.uint8, count=n * 4, offset=14 + hdr).reshape(n, 4)
        lut = np.zeros(256, dtype=np.uint8)
        lut[:n] = _bgr_to_gray(pal[:, 0], pal[:, 1], pal[:, 2])
    return {"w": w, "h": abs(h), "data_off": data_off, "stride": ((w * bpp + 31) // 32) * 4,
            "bytes_pp": bpp // 8, "flip": h > 0, "lut": lut}


def decode_bmp_gray(buf: bytes):
    """Uncompressed 8/24/32-bit BMP -> uint8 [H, W] grayscale; None if not such a file."""
    lay = bmp_layout(buf)
    if lay is None:
        return None
    w, h, row, nb = lay["w"], lay["h"], lay["stride"], lay["bytes_pp"]
    if len(buf) < lay["data_off"] + row * h:
        return None
    raw = np.frombuffer(buf, dtype=np.uint8, count=row * h, offset=lay["data_off"]).reshape(h, row)
    if nb == 1:
        lut = lay["lut"]
        if np.array_equal(lut, np.arange(256, dtype=np.uint8)):
            img = raw[:, :w]            # grey ramp palette (what cameras write): the index IS the value
        else:
            img = lut[raw[:, :w]]
    else:
        px = raw[:, : w * nb].reshape(h, w, nb)
        img = _bgr_to_gray(px[..., 0], px[..., 1], px[..., 2])
    if lay["flip"]:
        img = img[::-1]
    return np.ascontiguousarray(img)


IDENTITY_LUT = np.arange(256, dtype=np.uint8)


def stage_raw(path: str, slot: np.ndarray, H: int, W: int):
    """Put the file `path` into `slot` (a uint8 view of pinned staging memory) for the device unpack
    (tpiv_bmp_unpack): an uncompressed BMP of the right shape goes in as its raw file bytes; anything
    else is decoded on the host (imdecode_gray) and stored as headerless top-down pixels.  Returns
    (data_off, stride, bytes_pp, flip, lut) or None when the file cannot be decoded / has another shape."""
    try:
        with open(path, "rb", buffering=0) as f:
            size = os.fstat(f.fileno()).st_size
            if size <= slot.size:
                got = f.readinto(memoryview(slot)[:size])
                lay = bmp_layout(slot[:size]) if got == size else None
                if lay is not None and (lay["h"], lay["w"]) == (H, W) and \
                        lay["data_off"] + lay["stride"] * H <= size:
                    lut = lay["lut"] if lay["lut"] is not None else IDENTITY_LUT
                    return lay["data_off"], lay["stride"], lay["bytes_pp"], int(lay["flip"]), lut
    except OSError:
        return None
    img = imdecode_gray(path)
    if img is None or img.shape != (H, W) or H * W > slot.size:
        return None
    slot[:H * W] = img.reshape(-1)
    return 0, W, 1, 0, IDENTITY_LUT


def parse_bmp_headers(raw: np.ndarray, sizes: np.ndarray, H: int, W: int):
    """Headers of a batch of files sitting in the slots raw[i] (uint8 [n, cap]; sizes[i] bytes each, -1: not read) in
    one numpy sweep.  Returns a list with one (data_off, stride, bytes_pp, flip, lut) per file -- what stage_raw returns
    -- or None where the file needs the per-file path (other formats, odd headers, another frame shape, read errors)."""
    n = raw.shape[0]
    out = [None] * n
    ok = sizes >= 54 + 1024
    if not ok.any():
        return out
    hdr = np.ascontiguousarray(raw[:, :54 + 1024])
    u32 = lambda off: hdr[:, off:off + 4].copy().view("<u4")[:, 0].astype(np.int64)      # noqa: E731
    i32 = lambda off: hdr[:, off:off + 4].copy().view("<i4")[:, 0].astype(np.int64)      # noqa: E731
    u16 = lambda off: hdr[:, off:off + 2].copy().view("<u2")[:, 0].astype(np.int64)      # noqa: E731
    data_off, hsz, w, h, planes, bpp, comp, ncol = u32(10), u32(14), i32(18), i32(22), u16(26), u16(28), u32(30), u32(46)
    stride = ((w * bpp + 31) // 32) * 4
    # the common camera file: BITMAPINFOHEADER, one plane, uncompressed, 8 / 24 / 32 bit, full palette right behind the
    # header, pixel data behind header and palette.  BI_BITFIELDS (comp == 3) only with the standard channel masks of a
    # 32-bit file (B, G, R in the low three bytes, read from offset 54): any other mask layout -- and every other header --
    # goes to the per-file decoder, which honours the masks.
    std_masks = (u32(54) == 0x00FF0000) & (u32(58) == 0x0000FF00) & (u32(62) == 0x000000FF)
    plain = ok & (hdr[:, 0] == 0x42) & (hdr[:, 1] == 0x4D) & (hsz == 40) & (planes == 1) & \
        ((comp == 0) | ((comp == 3) & (bpp == 32) & std_masks & (data_off >= 54 + 12))) & (w == W) & \
        (np.abs(h) == H) & np.isin(bpp, (8, 24, 32)) & (data_off + stride * H <= sizes) & \
        ((bpp != 8) | (ncol == 0) | (ncol == 256)) & (data_off >= 54 + np.where(bpp == 8, 1024, 0))
    if plain.any():
        pal = hdr[:, 54:54 + 1024].reshape(n, 256, 4)
        luts = _bgr_to_gray(pal[..., 0], pal[..., 1], pal[..., 2])
        for i in np.flatnonzero(plain):
            lut = luts[i] if bpp[i] == 8 else IDENTITY_LUT
            out[i] = (int(data_off[i]), int(stride[i]), int(bpp[i] // 8), int(h[i] > 0), lut)
    return out


def stage_batch(paths, raw: np.ndarray, H: int, W: int, threads: int = 8):
    """stage_raw for a whole batch without the interpreter in the loop: the files are read into the slots raw[i] (uint8
    [n, cap], page-locked) by native reader threads (tpiv_read_files; the GIL is released for the duration) and the BMP
    headers of all of them are parsed in one numpy sweep (parse_bmp_headers)."""
    import ctypes as C

    from ._lib import check, lib
    n, cap = raw.shape
    arr = (C.c_char_p * n)(*[os.fsencode(p_) for p_ in paths])
    sizes = np.empty(n, dtype=np.int64)
    check(lib.tpiv_read_files(arr, n, raw.ctypes.data_as(C.c_void_p), cap, int(threads),
                              sizes.ctypes.data_as(C.POINTER(C.c_longlong))))
    return parse_bmp_headers(raw, sizes, H, W)


class ReadAhead:
    """The run's files streamed into page-locked staging buffers by native reader threads (tpiv_reader_*): batch k of
    `files_per_batch` files lands in bufs[k % len(bufs)], at most len(bufs) batches ahead of the consumer.  next()
    blocks without the GIL until the next batch is complete -> (buffer index, sizes) or None at the end; release() hands
    the oldest outstanding buffer back.  read=False marks every file unread (sizes -1) without touching the disk -- for
    formats the device cannot unpack, which take the per-file path anyway."""

    def __init__(self, paths, files_per_batch: int, bufs, slot_bytes: int, threads: int = 8, read: bool = True):
        import ctypes as C

        from ._lib import lib
        self._C, self._lib = C, lib
        self._n, self._fpb, self._read, self._k, self._nb = len(paths), int(files_per_batch), read, 0, len(bufs)
        self._h = None
        if read:
            arr = (C.c_char_p * len(paths))(*[os.fsencode(p_) for p_ in paths])
            ptrs = (C.c_void_p * len(bufs))(*[int(b_) for b_ in bufs])
            self._h = lib.tpiv_reader_open(arr, len(paths), self._fpb, ptrs, len(bufs), int(slot_bytes), int(threads))
            if not self._h:
                raise ValueError(lib.tpiv_last_error().decode("utf-8", "replace"))
        self._sizes = np.empty(self._fpb, dtype=np.int64)
        self._idx, self._cnt = C.c_int(0), C.c_int(0)

    def next(self):
        if not self._read:
            if self._k * self._fpb >= self._n:
                return None
            cnt = min(self._fpb, self._n - self._k * self._fpb)
            self._k += 1
            return (self._k - 1) % self._nb, np.full(cnt, -1, dtype=np.int64)
        from ._lib import check
        C = self._C
        check(self._lib.tpiv_reader_next(self._h, C.byref(self._cnt), C.byref(self._idx),
                                         self._sizes.ctypes.data_as(C.POINTER(C.c_longlong))))
        cnt = self._cnt.value
        if cnt == 0:
            return None
        return self._idx.value, self._sizes[:cnt].copy()

    def release(self):
        if self._h:
            from ._lib import check
            check(self._lib.tpiv_reader_release(self._h))

    def close(self):
        if self._h:
            self._lib.tpiv_reader_close(self._h)
            self._h = None

    __del__ = close


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
        """(frame_a, frame_b) of pair `index`, or (None, None) when either file cannot be decoded
        (the reference then skips the pair, B:138-139)."""
        index = index.tolist() if torch.is_tensor(index) else index
        path_a, path_b = self.img_pairs[index][0], self.img_pairs[index][-1]
        frames = [imdecode_gray(path_b), imdecode_gray(path_a)]          # (the reference reads b first)
        if any(f is None for f in frames):
            return None, None
        frame_b, frame_a = frames
        if self.transform is None:
            return frame_a, frame_b
        return self.transform(frame_a), self.transform(frame_b)
