"""Minimal OpenEXR scanline reader (and a writer for tests / tools) -- the input side of `SkyMap::new`
(lib/backgrounds.rs:58-81), which the reference implements with the `exr 1.4.2` crate's
`read_first_rgba_layer_from_file`.

Scope: single-part scanline files; compression NONE, RLE, ZIPS (1 line per chunk) and ZIP (16 lines); HALF and
FLOAT channels named R, G, B (A is read and dropped, as the reference drops it: backgrounds.rs:66-69); sampling 1.
Tiled, deep, multi-part files and PIZ / PXR24 / B44 / DWA compression raise `ExrError` -- convert such files once
(e.g. with `oiiotool in.exr --compression zip -o out.exr`) or supply the texels as a `.npy` array.

Pure host-side plumbing (numpy + zlib); nothing here is on the GPU path.  The HDRIs the reference's GUI scene loads
(assets/lake.exr, studio.exr) are absent from the reference tree, so this reader is exercised against files written
by `write_exr` below (round trips in every supported compression / pixel type): "parity unpinned" for real assets.
"""
from __future__ import annotations

import struct
import zlib
from typing import Dict, Tuple

import numpy as np

MAGIC = 20000630
COMPRESSION = {0: "none", 1: "rle", 2: "zips", 3: "zip"}
LINES_PER_CHUNK = {0: 1, 1: 1, 2: 1, 3: 16}
PIXEL_DTYPE = {0: np.dtype("<u4"), 1: np.dtype("<f2"), 2: np.dtype("<f4")}


class ExrError(ValueError):
    pass


def _cstr(buf: bytes, pos: int) -> Tuple[str, int]:
    end = buf.index(b"\0", pos)
    return buf[pos:end].decode("latin-1"), end + 1


def _undo_predictor_and_interleave(raw: bytes) -> bytes:
    """OpenEXR's byte-level post-processing of ZIP / RLE chunks: delta predictor, then the two halves interleaved."""
    t = np.frombuffer(raw, dtype=np.uint8).astype(np.int32)
    t[1:] -= 128
    t = (np.cumsum(t) & 0xFF).astype(np.uint8)  # t[i] = t[i-1] + t[i] - 128 (mod 256)
    n = t.size
    half = (n + 1) // 2
    out = np.empty(n, dtype=np.uint8)
    out[0::2] = t[:half]
    out[1::2] = t[half:]
    return out.tobytes()


def _apply_predictor_and_interleave(data: bytes) -> bytes:
    s = np.frombuffer(data, dtype=np.uint8)
    n = s.size
    half = (n + 1) // 2
    t = np.empty(n, dtype=np.uint8)
    t[:half] = s[0::2]
    t[half:] = s[1::2]
    d = t.astype(np.int32)
    d[1:] = (d[1:] - d[:-1] + 128 + 256) & 0xFF
    return d.astype(np.uint8).tobytes()


def _rle_decode(buf: bytes, expected: int) -> bytes:
    out = bytearray()
    i = 0
    while i < len(buf):
        c = struct.unpack_from("b", buf, i)[0]
        i += 1
        if c < 0:  # literal run of -c bytes
            out += buf[i:i - c]
            i += -c
        else:      # c+1 copies of the next byte
            out += bytes([buf[i]]) * (c + 1)
            i += 1
    if len(out) != expected:
        raise ExrError("corrupt RLE chunk")
    return bytes(out)


def _rle_encode(data: bytes) -> bytes:
    out = bytearray()
    i, n = 0, len(data)
    while i < n:
        j = i
        while j + 1 < n and data[j + 1] == data[i] and j - i < 126:
            j += 1
        if j - i >= 2:  # run of >= 3
            out += struct.pack("b", j - i) + bytes([data[i]])
            i = j + 1
        else:
            k = i
            while k < n and k - i < 127 and not (k + 2 < n and data[k] == data[k + 1] == data[k + 2]):
                k += 1
            out += struct.pack("b", -(k - i)) + data[i:k]
            i = k
    return bytes(out)


def read_header(buf: bytes) -> Tuple[Dict[str, object], int]:
    if len(buf) < 8 or struct.unpack_from("<i", buf, 0)[0] != MAGIC:
        raise ExrError("not an OpenEXR file")
    version = struct.unpack_from("<i", buf, 4)[0]
    if version & 0x200:
        raise ExrError("tiled OpenEXR files are not supported")
    if version & (0x800 | 0x1000):
        raise ExrError("deep / multi-part OpenEXR files are not supported")
    pos = 8
    attrs: Dict[str, object] = {}
    while True:
        if buf[pos] == 0:
            pos += 1
            break
        name, pos = _cstr(buf, pos)
        typ, pos = _cstr(buf, pos)
        size = struct.unpack_from("<i", buf, pos)[0]
        pos += 4
        val = buf[pos:pos + size]
        pos += size
        if typ == "chlist":
            chans = []
            p = 0
            while val[p] != 0:
                cname, p = _cstr(val, p)
                ptype, _plinear, xs, ys = struct.unpack_from("<iB3xii", val, p)
                p += 16
                chans.append((cname, ptype, xs, ys))
            attrs[name] = chans
        elif typ == "compression":
            attrs[name] = val[0]
        elif typ == "box2i":
            attrs[name] = struct.unpack("<4i", val)
        elif typ == "lineOrder":
            attrs[name] = val[0]
        else:
            attrs[name] = val
    for need in ("channels", "compression", "dataWindow"):
        if need not in attrs:
            raise ExrError(f"missing header attribute {need}")
    return attrs, pos


def read_exr(path: str) -> np.ndarray:
    """Returns the RGB texels as float32 [height, width, 3], row 0 = top scanline (y = dataWindow.min.y): the layout
    SkyMap::new builds with `image[x + y*width]` (backgrounds.rs:66-69)."""
    with open(path, "rb") as fh:
        buf = fh.read()
    attrs, pos = read_header(buf)
    comp = attrs["compression"]
    if comp not in COMPRESSION:
        raise ExrError(f"unsupported OpenEXR compression {comp} (supported: none, rle, zips, zip)")
    xmin, ymin, xmax, ymax = attrs["dataWindow"]
    width, height = xmax - xmin + 1, ymax - ymin + 1
    chans = attrs["channels"]  # stored alphabetically
    for cname, ptype, xs, ys in chans:
        if xs != 1 or ys != 1:
            raise ExrError("sub-sampled channels are not supported")
        if ptype not in PIXEL_DTYPE:
            raise ExrError("unknown pixel type")
    names = [c[0] for c in chans]
    for need in ("R", "G", "B"):
        if need not in names:
            raise ExrError(f"no {need} channel (channels: {names})")
    line_bytes = sum(PIXEL_DTYPE[c[1]].itemsize for c in chans) * width
    lpc = LINES_PER_CHUNK[comp]
    n_chunks = (height + lpc - 1) // lpc
    offsets = struct.unpack_from(f"<{n_chunks}Q", buf, pos)
    img = np.zeros((height, width, 3), dtype=np.float32)
    rgb_index = {"R": 0, "G": 1, "B": 2}
    for off in offsets:
        y0, size = struct.unpack_from("<ii", buf, off)
        data = buf[off + 8:off + 8 + size]
        lines = min(lpc, ymax - y0 + 1)
        raw_size = lines * line_bytes
        if comp != 0 and size < raw_size:
            if comp == 1:
                data = _undo_predictor_and_interleave(_rle_decode(data, raw_size))
            else:
                data = _undo_predictor_and_interleave(zlib.decompress(data))
        if len(data) != raw_size:
            raise ExrError("corrupt chunk")
        p = 0
        for ln in range(lines):
            row = y0 - ymin + ln
            for cname, ptype, _, _ in chans:
                dt = PIXEL_DTYPE[ptype]
                vals = np.frombuffer(data, dtype=dt, count=width, offset=p)
                p += dt.itemsize * width
                if cname in rgb_index:
                    img[row, :, rgb_index[cname]] = vals.astype(np.float32)
    return img


def write_exr(path: str, rgb: np.ndarray, compression: str = "zip", pixel_type: str = "half", alpha: bool = False,
              data_window_origin: Tuple[int, int] = (0, 0)) -> None:
    """Writes [height, width, 3] texels as a single-part scanline OpenEXR file (tests, tools)."""
    comp = {v: k for k, v in COMPRESSION.items()}[compression]
    ptype = {"half": 1, "float": 2}[pixel_type]
    dt = PIXEL_DTYPE[ptype]
    rgb = np.asarray(rgb, dtype=np.float32)
    height, width, _ = rgb.shape
    chans = (["A"] if alpha else []) + ["B", "G", "R"]  # alphabetical
    x0, y0 = data_window_origin

    def attr(name, typ, val):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(val)) + val

    chlist = b"".join(c.encode() + b"\0" + struct.pack("<iB3xii", ptype, 0, 1, 1) for c in chans) + b"\0"
    box = struct.pack("<4i", x0, y0, x0 + width - 1, y0 + height - 1)
    header = struct.pack("<ii", MAGIC, 2)
    header += attr("channels", "chlist", chlist) + attr("compression", "compression", bytes([comp]))
    header += attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box)
    header += attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    header += attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0))
    header += attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    lpc = LINES_PER_CHUNK[comp]
    chunks = []
    for cy in range(0, height, lpc):
        raw = bytearray()
        for row in range(cy, min(cy + lpc, height)):
            for c in chans:
                plane = np.ones(width, dtype=np.float32) if c == "A" else rgb[row, :, "RGB".index(c)]
                raw += plane.astype(dt).tobytes()
        raw = bytes(raw)
        data = raw
        if comp == 1:
            enc = _rle_encode(_apply_predictor_and_interleave(raw))
            data = enc if len(enc) < len(raw) else raw
        elif comp in (2, 3):
            enc = zlib.compress(_apply_predictor_and_interleave(raw))
            data = enc if len(enc) < len(raw) else raw
        chunks.append(struct.pack("<ii", y0 + cy, len(data)) + data)
    table_pos = len(header)
    pos = table_pos + 8 * len(chunks)
    offsets = []
    for ch in chunks:
        offsets.append(pos)
        pos += len(ch)
    with open(path, "wb") as fh:
        fh.write(header + struct.pack(f"<{len(chunks)}Q", *offsets) + b"".join(chunks))
