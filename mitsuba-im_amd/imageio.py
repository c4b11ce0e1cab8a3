"""Image files on either side of the path: texture / environment-map inputs and rendered films.

The reference reads images through `Bitmap(Bitmap::EAuto, stream)` (src/libcore/bitmap.cpp:2461-2560: OpenEXR, PNG, JPEG, RGBE, PFM, TGA, BMP)
and hands textures to the MIP map as linear floats (`Bitmap::convert(..., EFloat, gamma 1.0)`, include/mitsuba/core/mipmap.h:160-176).  Here:

  * OpenEXR   -- own reader for single-part scanline files with NONE / RLE / ZIPS / ZIP / PIZ compression and UINT / HALF / FLOAT channels (the layout of
                 the OpenEXR file format: magic 0x01312f76, attribute list, line-offset table, per-block [y, size, data], channels stored line by
                 line in name order; ZIP / RLE blocks are byte-delta predicted and split into even / odd halves).  PXR24 / B44 / DWA and tiled
                 or multi-part files are refused by name.  Own writer (FLOAT channels, ZIP) for rendered films.
  * PNG / JPEG / BMP / TGA -- through PIL when it is importable (it is in this image); 8-bit data is sRGB-encoded by default exactly as the
                 reference assumes (bitmap.cpp:284-287: gamma -1 for EUInt8) and is linearised with its curve (fmtconv.cpp:1092-1102); alpha stays linear.
                 16-bit grey PNGs are linear (gamma 1), as there.

Parity note: file DECODING is not pinned against the reference (its build in this container has no OpenEXR / libpng / libjpeg); the
decoders are lossless integer / IEEE-half unpacking, round-trip tested and, for PNG, checked against PIL's own encoder.  What happens to the decoded
texels afterwards (resampling, pyramid, lookups) is pinned (tests/test_oracle_golden.py::test_mip_pyramid_builder_vs_reference, bitmap_room).
"""
import os
import struct
import zlib

import numpy as np

f32 = np.float32


class ImageError(ValueError):
    pass


# ---- OpenEXR ---------------------------------------------------------------------------------------------------------------------------
_EXR_MAGIC = 20000630
_COMPRESSION = {0: ("NONE", 1), 1: ("RLE", 1), 2: ("ZIPS", 1), 3: ("ZIP", 16), 4: ("PIZ", 32), 5: ("PXR24", 16), 6: ("B44", 32), 7: ("B44A", 32), 8: ("DWAA", 32), 9: ("DWAB", 256)}
_PIXEL = {0: np.dtype("<u4"), 1: np.dtype("<f2"), 2: np.dtype("<f4")}


def _cstr(d, p):
    e = d.index(b"\0", p)
    return d[p:e].decode("latin-1"), e + 1


def _unpredict(buf):
    """Undo the byte-delta predictor and the even / odd split of ZIP and RLE blocks."""
    a = np.frombuffer(buf, np.uint8).astype(np.int64)
    if len(a) == 0:
        return b""
    a[1:] -= 128
    a = (np.cumsum(a) & 0xFF).astype(np.uint8)
    half = (len(a) + 1) // 2
    out = np.empty(len(a), np.uint8)
    out[0::2] = a[:half]; out[1::2] = a[half:]
    return out.tobytes()


def _predict(raw):
    a = np.frombuffer(raw, np.uint8)
    t = np.concatenate([a[0::2], a[1::2]]).astype(np.int64)
    d = t.copy(); d[1:] = (t[1:] - t[:-1] + 128 + 256) & 0xFF
    return d.astype(np.uint8).tobytes()


def _unrle(buf, expected):
    out = bytearray(); p = 0; n = len(buf)
    while p < n:
        c = buf[p] - 256 if buf[p] > 127 else buf[p]; p += 1
        if c < 0:
            out += buf[p:p - c]; p += -c
        else:
            out += bytes([buf[p]]) * (c + 1); p += 1
    if len(out) != expected:
        raise ImageError("OpenEXR: RLE block of unexpected size")
    return bytes(out)


# PIZ blocks (the OpenEXR PIZ codec): [minNonZero u16, maxNonZero u16, bitmap bytes, length i32, Huffman stream]; the Huffman stream decodes to
# 16-bit values, per channel a 2-D Haar-like wavelet (14-bit variant when the block uses fewer than 2^14 distinct values), then a lookup
# table built from the bitmap of used values maps them back.
def _huf_decode(d, nraw):
    im, iM, _tlen, nbits = struct.unpack_from("<4I", d, 0)
    if im > 65536 or iM > 65536 or im > iM:
        raise ImageError("OpenEXR: corrupt PIZ Huffman header")
    p = 20; c = 0; lc = 0
    lengths = [0] * 65537
    i = im
    while i <= iM:                                       # code lengths, 6 bits each; 59..62 = short runs of zeros, 63 = long run (8 more bits)
        while lc < 6:
            c = (c << 8) | d[p]; p += 1; lc += 8
        lc -= 6; l = (c >> lc) & 63
        if l == 63:
            while lc < 8:
                c = (c << 8) | d[p]; p += 1; lc += 8
            lc -= 8; run = ((c >> lc) & 255) + 6
            i += run
        elif l >= 59:
            i += l - 59 + 2
        else:
            lengths[i] = l; i += 1
        c &= (1 << lc) - 1
    if i > iM + 1:
        raise ImageError("OpenEXR: corrupt PIZ Huffman table")
    count = [0] * 59
    for l in lengths[im:iM + 1]:
        count[l] += 1
    code = 0; first = [0] * 59
    for l in range(58, 0, -1):                           # canonical codes, longest first
        nc = (code + count[l]) >> 1; first[l] = code; code = nc
    table = [0] * (1 << 14); long_codes = {}
    for sym in range(im, iM + 1):
        l = lengths[sym]
        if l == 0:
            continue
        cd = first[l]; first[l] += 1
        if l <= 14:
            base = cd << (14 - l); e = (sym << 6) | l
            table[base:base + (1 << (14 - l))] = [e] * (1 << (14 - l))
        else:
            long_codes[(l, cd)] = sym
    rlc = iM; out = []; n = len(d); c = 0; lc = 0; used = 0; app = out.append
    while len(out) < nraw:
        while lc < 14 and p < n:
            c = (c << 8) | d[p]; p += 1; lc += 8
        e = table[(c >> (lc - 14)) & 0x3FFF] if lc >= 14 else table[(c << (14 - lc)) & 0x3FFF]
        if e:
            l = e & 63; sym = e >> 6
            if l > lc:
                raise ImageError("OpenEXR: PIZ Huffman stream ends early")
            lc -= l
        else:
            sym = -1
            for l in range(15, 59):
                while lc < l and p < n:
                    c = (c << 8) | d[p]; p += 1; lc += 8
                if lc < l:
                    break
                s2 = long_codes.get((l, (c >> (lc - l)) & ((1 << l) - 1)))
                if s2 is not None:
                    sym = s2; lc -= l; break
            if sym < 0:
                raise ImageError("OpenEXR: corrupt PIZ Huffman stream")
        used += l
        if sym == rlc:
            while lc < 8 and p < n:
                c = (c << 8) | d[p]; p += 1; lc += 8
            lc -= 8; used += 8; run = (c >> lc) & 255
            if not out or len(out) + run > nraw:
                raise ImageError("OpenEXR: corrupt PIZ run")
            out.extend([out[-1]] * run)
        else:
            app(sym)
        c &= (1 << lc) - 1
    if used != nbits:
        raise ImageError("OpenEXR: PIZ Huffman stream of unexpected length")
    return np.asarray(out, np.uint16)


def _wdec(l, h, w14):
    if w14:
        ls = l.astype(np.int16).astype(np.int32); hs = h.astype(np.int16).astype(np.int32)
        ai = ls + (hs & 1) + (hs >> 1)
        return (ai & 0xFFFF).astype(np.uint16), ((ai - hs) & 0xFFFF).astype(np.uint16)
    m = l.astype(np.int32); dd = h.astype(np.int32)
    bb = (m - (dd >> 1)) & 0xFFFF
    aa = (dd + bb - 0x8000) & 0xFFFF
    return aa.astype(np.uint16), bb.astype(np.uint16)


def _wav2_decode(a, max_value):
    """In-place inverse wavelet of one 2-D plane a[ny, nx] (uint16 view)."""
    ny, nx = a.shape; w14 = max_value < (1 << 14)
    n = min(nx, ny); p = 1
    while p <= n:
        p <<= 1
    p >>= 1; p2 = p; p >>= 1
    while p >= 1:
        rows = np.arange(0, ny - p2 + 1, p2); cols = np.arange(0, nx - p2 + 1, p2)
        if len(rows) and len(cols):
            R, C = np.ix_(rows, cols); R1, C1 = np.ix_(rows + p, cols + p)
            px, p01, p10, p11 = a[R, C], a[R, C1], a[R1, C], a[R1, C1]
            i00, i10 = _wdec(px, p10, w14); i01, i11 = _wdec(p01, p11, w14)
            a[R, C], a[R, C1] = _wdec(i00, i01, w14)
            a[R1, C], a[R1, C1] = _wdec(i10, i11, w14)
        if (nx & p) and len(rows):
            cx = len(cols) * p2
            a[rows, cx], a[rows + p, cx] = _wdec(a[rows, cx], a[rows + p, cx], w14)
        if ny & p:
            ry = len(rows) * p2
            if len(cols):
                a[ry, cols], a[ry, cols + p] = _wdec(a[ry, cols], a[ry, cols + p], w14)
        p2 = p; p >>= 1


def _unpiz(data, channels, w, lines):
    """One PIZ block -> the block's bytes in the plain scanline layout."""
    min_nz, max_nz = struct.unpack_from("<HH", data, 0); p = 4
    bitmap = np.zeros(8192, np.uint8)
    if min_nz <= max_nz:
        if max_nz >= 8192:
            raise ImageError("OpenEXR: corrupt PIZ bitmap")
        bitmap[min_nz:max_nz + 1] = np.frombuffer(data, np.uint8, max_nz - min_nz + 1, p); p += max_nz - min_nz + 1
    used = np.unpackbits(bitmap, bitorder="little").astype(bool); used[0] = True
    lut = np.zeros(65536, np.uint16); vals = np.nonzero(used)[0]; lut[:len(vals)] = vals; max_value = len(vals) - 1
    length = struct.unpack_from("<i", data, p)[0]; p += 4
    sizes = [dt.itemsize // 2 for _, dt in channels]
    nraw = sum(sizes) * w * lines
    tmp = _huf_decode(data[p:p + length], nraw) if nraw else np.zeros(0, np.uint16)
    planes = []; q = 0
    for sz in sizes:
        block = tmp[q:q + lines * w * sz].reshape(lines, w * sz); q += lines * w * sz
        for j in range(sz):
            sub = np.ascontiguousarray(block[:, j::sz]); _wav2_decode(sub, max_value); block[:, j::sz] = sub
        planes.append(lut[block])
    return b"".join(planes[ci][ly].astype("<u2").tobytes() for ly in range(lines) for ci in range(len(channels)))


def read_exr(path):
    """-> (pixels float32 [h, w, n], channel names in file order)."""
    with open(path, "rb") as f:
        d = f.read()
    if len(d) < 8 or struct.unpack_from("<i", d, 0)[0] != _EXR_MAGIC:
        raise ImageError(f"\"{os.path.basename(path)}\" is not an OpenEXR file")
    version = struct.unpack_from("<i", d, 4)[0]
    if version & 0x1A00:
        raise ImageError("OpenEXR: tiled, deep and multi-part files are not supported")
    p = 8; attrs = {}
    while d[p] != 0:
        name, p = _cstr(d, p); typ, p = _cstr(d, p)
        n = struct.unpack_from("<i", d, p)[0]; p += 4
        attrs[name] = (typ, d[p:p + n]); p += n
    p += 1
    for need in ("channels", "compression", "dataWindow"):
        if need not in attrs:
            raise ImageError(f"OpenEXR: missing attribute \"{need}\"")
    channels = []; c = attrs["channels"][1]; q = 0
    while c[q] != 0:
        name, q = _cstr(c, q)
        ptype, _lin, xs, ys = struct.unpack_from("<iB3xii", c, q); q += 16
        if xs != 1 or ys != 1:
            raise ImageError("OpenEXR: subsampled channels are not supported")
        if ptype not in _PIXEL:
            raise ImageError("OpenEXR: unknown pixel type")
        channels.append((name, _PIXEL[ptype]))
    comp = attrs["compression"][1][0]
    cname, lines_per_block = _COMPRESSION.get(comp, ("?", 1))
    if cname not in ("NONE", "RLE", "ZIPS", "ZIP", "PIZ"):
        raise ImageError(f"OpenEXR: {cname} compression is not supported (re-save with ZIP or no compression)")
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    if w <= 0 or h <= 0:
        raise ImageError("OpenEXR: empty data window")
    nblocks = (h + lines_per_block - 1) // lines_per_block
    offsets = struct.unpack_from(f"<{nblocks}Q", d, p)
    line_bytes = sum(dt.itemsize for _, dt in channels) * w
    out = np.zeros((h, w, len(channels)), f32)
    for off in offsets:
        y, size = struct.unpack_from("<ii", d, off); data = d[off + 8:off + 8 + size]
        y -= y0
        if y < 0 or y >= h or len(data) != size:
            raise ImageError("OpenEXR: corrupt block table")
        lines = min(lines_per_block, h - y); expected = line_bytes * lines
        if size != expected:                      # blocks that do not shrink are stored raw
            if cname in ("ZIP", "ZIPS"):
                data = _unpredict(zlib.decompress(data))
            elif cname == "RLE":
                data = _unpredict(_unrle(data, expected))
            elif cname == "PIZ":
                data = _unpiz(data, channels, w, lines)
        if len(data) != expected:
            raise ImageError("OpenEXR: block of unexpected size")
        q = 0
        for ly in range(lines):
            for ci, (_, dt) in enumerate(channels):
                out[y + ly, :, ci] = np.frombuffer(data, dt, w, q).astype(f32); q += w * dt.itemsize
    return out, [n for n, _ in channels]


def write_exr(path, image, channels=None):
    """float32 [h, w, n] -> single-part scanline OpenEXR, FLOAT channels, ZIP compression.  channels: names per plane (default Y / RGB / RGBA)."""
    a = np.asarray(image, f32)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, n = a.shape
    names = list(channels) if channels else {1: ["Y"], 3: ["R", "G", "B"], 4: ["R", "G", "B", "A"]}.get(n)
    if not names or len(names) != n:
        raise ImageError("write_exr: channel names needed")
    order = sorted(range(n), key=lambda i: names[i])                      # channels are stored in name order

    def attr(name, typ, payload):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(payload)) + payload
    chlist = b"".join(names[i].encode() + b"\0" + struct.pack("<iB3xii", 2, 0, 1, 1) for i in order) + b"\0"
    box = struct.pack("<4i", 0, 0, w - 1, h - 1)
    head = struct.pack("<ii", _EXR_MAGIC, 2) + attr("channels", "chlist", chlist) + attr("compression", "compression", b"\x03") + \
        attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") + \
        attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0)) + \
        attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    blocks = []
    for y in range(0, h, 16):
        raw = b"".join(np.ascontiguousarray(a[ly, :, i], "<f4").tobytes() for ly in range(y, min(y + 16, h)) for i in order)
        z = zlib.compress(_predict(raw))
        blocks.append((y, z if len(z) < len(raw) else raw))
    pos = len(head) + 8 * len(blocks); table = []
    for _, b in blocks:
        table.append(pos); pos += 8 + len(b)
    with open(path, "wb") as f:
        f.write(head); f.write(struct.pack(f"<{len(table)}Q", *table))
        for y, b in blocks:
            f.write(struct.pack("<ii", y, len(b))); f.write(b)


def _exr_planes(px, names):
    """Order the planes as R, G, B(, A) or Y(, A); layers ("diffuse.R") are matched by their last component."""
    short = [n.split(".")[-1].upper() for n in names]
    pick = lambda c: px[:, :, short.index(c)]
    if all(c in short for c in "RGB"):
        planes = [pick("R"), pick("G"), pick("B")]
    elif "Y" in short:
        planes = [pick("Y")]
    elif len(names) == 1:
        planes = [px[:, :, 0]]
    else:
        raise ImageError("OpenEXR: neither R/G/B nor Y channels (found: " + ", ".join(names) + ")")
    if "A" in short:
        planes.append(pick("A"))
    return np.stack(planes, 2)


# ---- 8 / 16-bit formats through PIL ---------------------------------------------------------------------------------------------------------
def undo_gamma(v, gamma):
    """fmtconv.cpp:1092-1102 undoGamma: -1 = the sRGB curve, otherwise a power law."""
    v = np.asarray(v, f32)
    if gamma == -1:
        return np.where(v <= f32(0.04045), v * f32(1.0 / 12.92), np.power((v + f32(0.055)) * f32(1.0 / 1.055), f32(2.4))).astype(f32)
    if gamma == 1:
        return v
    return np.power(v, f32(gamma)).astype(f32)


def read_ldr(path, gamma=0.0):
    """PNG / JPEG / BMP / TGA -> linear float32 [h, w, n] (n = 1 Y, 2 YA, 3 RGB, 4 RGBA).  gamma: BitmapTexture's override (0 = the file's default:
    sRGB for 8-bit data, linear for 16-bit; bitmap.cpp:284-287)."""
    try:
        from PIL import Image
    except ImportError:
        raise ImageError(f"reading \"{os.path.basename(path)}\" needs PIL, which is not importable here (convert the image to .exr, .pfm, .hdr or .npy)")
    with Image.open(path) as im:
        im.load()
        mode = im.mode
        if mode in ("I;16", "I;16L", "I;16B", "I"):
            a = np.asarray(im).astype(f32)[:, :, None] / f32(65535.0); default_gamma = 1.0
        else:
            if mode == "P":
                im = im.convert("RGBA" if "transparency" in im.info else "RGB")
            elif mode == "1":
                im = im.convert("L")
            elif mode not in ("L", "LA", "RGB", "RGBA"):
                im = im.convert("RGB")
            a = np.asarray(im).astype(f32) / f32(255.0); default_gamma = -1.0
            if a.ndim == 2:
                a = a[:, :, None]
    g = default_gamma if gamma == 0 else gamma
    n = a.shape[2]; colour = n - 1 if n in (2, 4) else n                   # alpha is never gamma-encoded
    a = a.copy(); a[:, :, :colour] = undo_gamma(a[:, :, :colour], g)
    return np.ascontiguousarray(a, f32)


def write_ldr(path, image, gamma=-1.0):
    """Linear float [h, w, 3] -> 8-bit sRGB PNG / JPEG (ldrfilm's default `gamma` tonemapper without exposure, src/films/ldrfilm.cpp)."""
    from PIL import Image
    v = np.clip(np.asarray(image, np.float64), 0.0, 1.0)
    enc = np.where(v <= 0.0031308, 12.92 * v, 1.055 * np.power(v, 1 / 2.4) - 0.055) if gamma == -1 else np.power(v, 1.0 / gamma)
    Image.fromarray((enc * 255.0 + 0.5).astype(np.uint8)).save(path)
