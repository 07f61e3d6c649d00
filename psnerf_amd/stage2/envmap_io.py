"""Environment-map files of the relighting path without cv2 / imageio / OpenEXR (none of them is a dependency here):
Radiance ``.hdr`` (RGBE, flat or run-length encoded scanlines) and scanline OpenEXR ``.exr`` (half / float / uint channels,
compression NONE, RLE, ZIPS, ZIP) -> float32 RGB arrays [H, W, 3], which is what the reference's
``load_light`` / ``read_exr`` / ``read_hdr`` hand to the renderer (stage2/utils/eval_utils.py:11-38; the reference decodes with
cv2 and swaps BGR -> RGB).  Tiled / multi-part / deep EXR files and the PIZ, PXR24, B44, DWA compressions are rejected with a
clear message (convert once with any OpenEXR tool, or store the map as .npy).  Host-side data format code: no GPU involved."""
import struct
import zlib

import numpy as np


# --------------------------------------------------------------------------------------------------- Radiance .hdr (RGBE)
def _rgbe_to_float(rgbe):
    """[..., 4] uint8 -> [..., 3] float32: mantissa * 2^(e - 136), zero exponent = black (the decoder of the reference's cv2)."""
    e = rgbe[..., 3].astype(np.int32)
    scale = np.where(e == 0, 0.0, np.ldexp(1.0, e - 136)).astype(np.float32)
    return rgbe[..., :3].astype(np.float32) * scale[..., None]


def read_hdr(path):
    """Radiance picture -> float32 RGB [H, W, 3] (standard orientation ``-Y H +X W`` only)."""
    with open(path, 'rb') as f:
        data = f.read()
    pos = data.find(b'\n\n')
    if not (data.startswith(b'#?RADIANCE') or data.startswith(b'#?RGBE')) or pos < 0:
        raise ValueError('%s: not a Radiance .hdr file' % path)
    header = data[:pos].decode('ascii', 'replace')
    if 'FORMAT=32-bit_rle_xyze' in header:
        raise NotImplementedError('%s: XYZE pictures are not supported' % path)
    end = data.index(b'\n', pos + 2)
    res = data[pos + 2:end].decode('ascii').split()
    if len(res) != 4 or res[0] != '-Y' or res[2] != '+X':
        raise NotImplementedError('%s: resolution line %r (only "-Y H +X W")' % (path, ' '.join(res)))
    H, W = int(res[1]), int(res[3])
    buf = np.frombuffer(data, dtype=np.uint8, offset=end + 1)
    out = np.empty((H, W, 4), dtype=np.uint8)
    p = 0
    for y in range(H):
        if 8 <= W < 32768 and p + 4 <= buf.size and buf[p] == 2 and buf[p + 1] == 2 and ((int(buf[p + 2]) << 8) | int(buf[p + 3])) == W:
            p += 4  # adaptive run-length encoding: the four channels of the scanline one after the other
            for c in range(4):
                x = 0
                while x < W:
                    n = int(buf[p]); p += 1
                    if n > 128:
                        n -= 128
                        if n == 0 or x + n > W:
                            raise ValueError('%s: corrupt run in scanline %d' % (path, y))
                        out[y, x:x + n, c] = buf[p]; p += 1
                    else:
                        if n == 0 or x + n > W:
                            raise ValueError('%s: corrupt literal in scanline %d' % (path, y))
                        out[y, x:x + n, c] = buf[p:p + n]; p += n
                    x += n
        else:  # flat pixels
            if p + 4 * W > buf.size:
                raise ValueError('%s: truncated at scanline %d' % (path, y))
            out[y] = buf[p:p + 4 * W].reshape(W, 4); p += 4 * W
    return _rgbe_to_float(out)


def write_hdr(path, rgb, rle=True):
    """float32 RGB [H, W, 3] -> Radiance picture (tests and tooling; the reference writes its previews with cv2.imwrite)."""
    rgb = np.asarray(rgb, dtype=np.float32)
    H, W, _ = rgb.shape
    m = rgb.max(axis=-1)
    mant, e = np.frexp(m)  # m = mant * 2^e, mant in [0.5, 1)
    scale = np.where(m < 1e-32, 0.0, mant * 256.0 / np.maximum(m, 1e-38))
    rgbe = np.zeros((H, W, 4), dtype=np.uint8)
    rgbe[..., :3] = np.clip(rgb * scale[..., None], 0, 255).astype(np.uint8)
    rgbe[..., 3] = np.where(m < 1e-32, 0, e + 128).astype(np.uint8)
    with open(path, 'wb') as f:
        f.write(b'#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n' % (H, W))
        if not (rle and 8 <= W < 32768):
            f.write(rgbe.tobytes())
            return
        for y in range(H):
            f.write(bytes([2, 2, W >> 8, W & 255]))
            for c in range(4):
                row = rgbe[y, :, c]
                x = 0
                while x < W:
                    run = 1
                    while x + run < W and run < 127 and row[x + run] == row[x]:
                        run += 1
                    if run >= 4:
                        f.write(bytes([128 + run, int(row[x])])); x += run
                        continue
                    start = x
                    x += run
                    while x < W and x - start < 128:  # literal until the next run of >= 4 equal bytes
                        r2 = 1
                        while x + r2 < W and r2 < 4 and row[x + r2] == row[x]:
                            r2 += 1
                        if r2 >= 4:
                            break
                        x += 1
                    n = min(x - start, 128)
                    x = start + n
                    f.write(bytes([n]) + row[start:start + n].tobytes())


# --------------------------------------------------------------------------------------------------- OpenEXR (scanline)
_EXR_COMPRESSION = {0: ('NONE', 1), 1: ('RLE', 1), 2: ('ZIPS', 1), 3: ('ZIP', 16)}
_EXR_UNSUPPORTED = {4: 'PIZ', 5: 'PXR24', 6: 'B44', 7: 'B44A', 8: 'DWAA', 9: 'DWAB'}
_EXR_PIXEL = {0: np.dtype('<u4'), 1: np.dtype('<f2'), 2: np.dtype('<f4')}


def _exr_unpredict(raw):
    """Inverse of OpenEXR's byte predictor + half/half interleave (ZIP, ZIPS and RLE chunks)."""
    t = np.frombuffer(raw, dtype=np.uint8).astype(np.int64)
    t[1:] -= 128
    t = (np.cumsum(t) & 255).astype(np.uint8)
    half = (t.size + 1) // 2
    out = np.empty_like(t)
    out[0::2] = t[:half]
    out[1::2] = t[half:]
    return out.tobytes()


def _exr_unrle(raw, expected):
    out = bytearray()
    p = 0
    while p < len(raw):
        n = struct.unpack_from('b', raw, p)[0]; p += 1
        if n < 0:
            out += raw[p:p - n]; p += -n
        else:
            out += raw[p:p + 1] * (n + 1); p += 1
    if len(out) != expected:
        raise ValueError('EXR RLE chunk: %d bytes, expected %d' % (len(out), expected))
    return bytes(out)


def read_exr(path):
    """Single-part scanline OpenEXR -> float32 RGB [H, W, 3] (channels R, G, B; a single channel Y is replicated)."""
    with open(path, 'rb') as f:
        data = f.read()
    if len(data) < 8 or struct.unpack_from('<I', data, 0)[0] != 20000630:
        raise ValueError('%s: not an OpenEXR file' % path)
    version = struct.unpack_from('<I', data, 4)[0]
    if version & 0x1a00:  # tiled 0x200, deep 0x800, multi-part 0x1000
        raise NotImplementedError('%s: tiled / deep / multi-part EXR files are not supported (flags 0x%x)' % (path, version & 0xff00))
    p = 8
    attrs = {}
    while data[p] != 0:
        e = data.index(b'\0', p); name = data[p:e].decode('ascii'); p = e + 1
        e = data.index(b'\0', p); typ = data[p:e].decode('ascii'); p = e + 1
        size = struct.unpack_from('<i', data, p)[0]; p += 4
        attrs[name] = (typ, data[p:p + size]); p += size
    p += 1
    comp = attrs['compression'][1][0]
    if comp in _EXR_UNSUPPORTED:
        raise NotImplementedError('%s: %s compression is not supported (NONE, RLE, ZIPS, ZIP are); re-save the map with one of '
                                  'those or as .npy' % (path, _EXR_UNSUPPORTED[comp]))
    lines_per_chunk = _EXR_COMPRESSION[comp][1]
    x0, y0, x1, y1 = struct.unpack('<4i', attrs['dataWindow'][1])
    W, H = x1 - x0 + 1, y1 - y0 + 1
    chans = []  # (name, numpy dtype) in file order (alphabetical)
    c = attrs['channels'][1]
    q = 0
    while c[q] != 0:
        e = c.index(b'\0', q); name = c[q:e].decode('ascii'); q = e + 1
        ptype, _plinear, xs, ys = struct.unpack_from('<iB3xii', c, q); q += 16
        if xs != 1 or ys != 1:
            raise NotImplementedError('%s: sub-sampled channel %s' % (path, name))
        chans.append((name, _EXR_PIXEL[ptype]))
    n_chunks = (H + lines_per_chunk - 1) // lines_per_chunk
    offsets = struct.unpack_from('<%dQ' % n_chunks, data, p)
    planes = {name: np.empty((H, W), dtype=np.float32) for name, _ in chans}
    line_bytes = sum(dt.itemsize for _, dt in chans) * W
    for off in offsets:
        y, size = struct.unpack_from('<ii', data, off)
        raw = data[off + 8:off + 8 + size]
        rows = min(lines_per_chunk, y0 + H - y)
        expected = rows * line_bytes
        if comp in (2, 3) and size < expected:
            raw = _exr_unpredict(zlib.decompress(raw))
        elif comp == 1 and size < expected:
            raw = _exr_unpredict(_exr_unrle(raw, expected))
        if len(raw) != expected:
            raise ValueError('%s: chunk at scanline %d has %d bytes, expected %d' % (path, y, len(raw), expected))
        q = 0
        for r in range(rows):
            for name, dt in chans:
                planes[name][y - y0 + r] = np.frombuffer(raw, dtype=dt, count=W, offset=q).astype(np.float32)
                q += dt.itemsize * W
    if all(k in planes for k in 'RGB'):
        return np.stack([planes['R'], planes['G'], planes['B']], axis=-1)
    if 'Y' in planes:
        return np.repeat(planes['Y'][..., None], 3, axis=-1)
    raise NotImplementedError('%s: channels %s (need R, G, B or Y)' % (path, [n for n, _ in chans]))


def write_exr(path, rgb, compression='ZIP', half=False):
    """float32 RGB [H, W, 3] -> scanline OpenEXR with channels B, G, R (tests and tooling)."""
    rgb = np.asarray(rgb, dtype=np.float32)
    H, W, _ = rgb.shape
    comp = {'NONE': 0, 'ZIPS': 2, 'ZIP': 3}[compression]
    lines = _EXR_COMPRESSION[comp][1]
    dt = np.dtype('<f2') if half else np.dtype('<f4')

    def attr(name, typ, payload):
        return name.encode() + b'\0' + typ.encode() + b'\0' + struct.pack('<i', len(payload)) + payload
    chlist = b''.join(n.encode() + b'\0' + struct.pack('<iB3xii', 1 if half else 2, 0, 1, 1) for n in 'BGR') + b'\0'
    box = struct.pack('<4i', 0, 0, W - 1, H - 1)
    header = struct.pack('<II', 20000630, 2) + attr('channels', 'chlist', chlist) + attr('compression', 'compression', bytes([comp])) \
        + attr('dataWindow', 'box2i', box) + attr('displayWindow', 'box2i', box) + attr('lineOrder', 'lineOrder', b"\0") \
        + attr('pixelAspectRatio', 'float', struct.pack('<f', 1.0)) + attr('screenWindowCenter', 'v2f', struct.pack('<2f', 0, 0)) \
        + attr('screenWindowWidth', 'float', struct.pack('<f', 1.0)) + b"\0"
    chunks = []
    for y in range(0, H, lines):
        raw = b''.join(rgb[r, :, c].astype(dt).tobytes() for r in range(y, min(y + lines, H)) for c in (2, 1, 0))
        if comp:
            t = np.frombuffer(raw, dtype=np.uint8)
            half_n = (t.size + 1) // 2
            s = np.concatenate([t[0::2], t[1::2]]).astype(np.int64)
            assert s.size == t.size and half_n == t[0::2].size
            d = s.copy()
            d[1:] = (s[1:] - s[:-1] + 128 + 256) & 255
            z = zlib.compress(d.astype(np.uint8).tobytes())
            if len(z) < len(raw):
                raw = z
        chunks.append((y, raw))
    off = len(header) + 8 * len(chunks)
    table = b''
    body = b''
    for y, raw in chunks:
        table += struct.pack('<Q', off + len(body))
        body += struct.pack('<ii', y, len(raw)) + raw
    with open(path, 'wb') as f:
        f.write(header + table + body)
