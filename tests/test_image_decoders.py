"""The image containers behind map_Kd / map_Ks / map_bump (host/vct_image.h): PNG (own inflate), baseline
JPEG, BMP, TGA, PPM.  The reference gets them from stb_image (R/Model.h:141-226).  Files are written with Pillow (test-side
only), decoded by the library, and compared: lossless formats bit for bit, JPEG within the spread two conforming
decoders show between themselves."""
import io
import os
import struct
import zlib

import numpy as np
import pytest

import vctpkg

PIL = pytest.importorskip("PIL.Image")


@pytest.fixture(scope="module")
def sc():
    vctpkg.load()
    from voxel_cone_tracing_amd import scene
    return scene


def picture(h=37, w=53, seed=3):
    r = np.random.default_rng(seed)
    ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    img = np.zeros((h, w, 4), np.uint8)
    img[..., 0] = (xs * 255 // max(w - 1, 1))
    img[..., 1] = (ys * 255 // max(h - 1, 1))
    img[..., 2] = r.integers(0, 256, (h, w))
    img[..., 3] = np.where((xs // 5 + ys // 7) % 2 == 0, 255, r.integers(0, 200, (h, w)))
    return img


def expect(img_rgba_top_down):
    return img_rgba_top_down[::-1]          # the library stores rows bottom-up


@pytest.mark.parametrize("mode", ["RGBA", "RGB", "L", "LA", "P", "1", "I;16"])
@pytest.mark.parametrize("interlaced", [False, True])
def test_png_colour_types_filters_and_adam7(sc, tmp_path, mode, interlaced):
    img = picture()
    if mode == "RGBA":
        im, want = PIL.fromarray(img, "RGBA"), img
    elif mode == "RGB":
        im = PIL.fromarray(img[..., :3], "RGB")
        want = np.dstack([img[..., :3], np.full(img.shape[:2], 255, np.uint8)])
    elif mode in ("L", "1", "I;16"):
        g = img[..., 2]
        if mode == "1":
            g = np.where(g > 127, 255, 0).astype(np.uint8)
            im = PIL.fromarray(g, "L").convert("1")
        elif mode == "I;16":
            g16 = (g.astype(np.uint16) << 8) | 0x37
            im = PIL.fromarray(g16, "I;16")
        else:
            im = PIL.fromarray(g, "L")
        want = np.dstack([g, g, g, np.full(g.shape, 255, np.uint8)])
    elif mode == "LA":
        im = PIL.fromarray(np.dstack([img[..., 2], img[..., 3]]), "LA")
        want = np.dstack([img[..., 2]] * 3 + [img[..., 3]])
    else:
        im = PIL.fromarray(img[..., :3], "RGB").quantize(17)
        want = np.asarray(im.convert("RGBA"))
    path = str(tmp_path / f"t_{mode.replace(';', '')}_{int(interlaced)}.png")
    if interlaced:
        # Pillow cannot write Adam7: interlace by hand (filter 0 rows per pass), same IHDR fields otherwise
        im.save(path)
        write_adam7(path, im)
    else:
        im.save(path)
    got = sc.load_image(path)
    assert got.shape == want.shape and np.array_equal(got, expect(want))


def write_adam7(path, im):
    """Re-encode a Pillow image as an Adam7-interlaced PNG (8-bit RGBA / RGB / L / LA, 16-bit L, 1-bit, palette)."""
    with open(path, "rb") as fh:
        data = fh.read()
    pos, chunks = 8, []
    while pos < len(data):
        ln = struct.unpack(">I", data[pos:pos + 4])[0]
        chunks.append((data[pos + 4:pos + 8], data[pos + 8:pos + 8 + ln]))
        pos += 12 + ln
    ihdr = bytearray(dict(chunks)[b"IHDR"])
    w, h, depth, ctype = struct.unpack(">IIBB", bytes(ihdr[:10]))
    raw = zlib.decompress(b"".join(d for t, d in chunks if t == b"IDAT"))
    chans = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bits = chans * depth
    stride = (w * bits + 7) // 8
    bpp = max(1, bits // 8)
    rows, prev = [], bytearray(stride)
    for y in range(h):                       # undo Pillow's filters: plain rows
        ft, line = raw[y * (stride + 1)], bytearray(raw[y * (stride + 1) + 1:(y + 1) * (stride + 1)])
        for i in range(stride):
            a = line[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ft == 1: line[i] = (line[i] + a) & 255
            elif ft == 2: line[i] = (line[i] + b) & 255
            elif ft == 3: line[i] = (line[i] + ((a + b) >> 1)) & 255
            elif ft == 4:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                line[i] = (line[i] + pr) & 255
        rows.append(line)
        prev = line

    def pixel_bits(row, x):
        v = 0
        for k in range(bits):
            bit = x * bits + k
            v = (v << 1) | ((row[bit >> 3] >> (7 - (bit & 7))) & 1)
        return v
    out = bytearray()
    for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
        pw, ph = (w - x0 + dx - 1) // dx, (h - y0 + dy - 1) // dy
        if pw <= 0 or ph <= 0:
            continue
        for y in range(ph):
            acc, nb = 0, 0
            line = bytearray()
            for x in range(pw):
                acc = (acc << bits) | pixel_bits(rows[y0 + y * dy], x0 + x * dx)
                nb += bits
                while nb >= 8:
                    line.append((acc >> (nb - 8)) & 255)
                    nb -= 8
            if nb:
                line.append((acc << (8 - nb)) & 255)
            out += b"\x00" + line
    ihdr[12] = 1

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    with open(path, "wb") as fh:
        fh.write(data[:8] + chunk(b"IHDR", bytes(ihdr)))
        for t, d in chunks:
            if t in (b"PLTE", b"tRNS"):
                fh.write(chunk(t, d))
        fh.write(chunk(b"IDAT", zlib.compress(bytes(out), 6)) + chunk(b"IEND", b""))


def test_png_stored_and_fixed_huffman_blocks(sc, tmp_path):
    """DEFLATE block types 0 (stored) and 1 (fixed codes) -- zlib level 0 and Z_FIXED -- beside the usual dynamic ones."""
    img = picture(9, 11)
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(img.shape[0]))

    def png(comp):
        def chunk(t, d):
            return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
        return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 11, 9, 8, 6, 0, 0, 0)) + chunk(b"IDAT", comp)
                + chunk(b"IEND", b""))
    fixed = zlib.compressobj(9, zlib.DEFLATED, 15, 9, zlib.Z_FIXED)
    for name, comp in (("stored", zlib.compress(raw, 0)), ("fixed", fixed.compress(raw) + fixed.flush())):
        p = tmp_path / f"{name}.png"
        p.write_bytes(png(comp))
        assert np.array_equal(sc.load_image(str(p)), expect(img))


def test_png_trns_colour_key(sc, tmp_path):
    img = picture()[..., :3].copy()
    img[5:9, 7:20] = (10, 200, 30)
    im = PIL.fromarray(img, "RGB")
    p = str(tmp_path / "key.png")
    im.save(p, transparency=(10, 200, 30))
    got = sc.load_image(p)
    want = np.dstack([img, np.where((img == (10, 200, 30)).all(2), 0, 255).astype(np.uint8)])
    assert np.array_equal(got, expect(want))


@pytest.mark.parametrize("subsampling,quality,grey,restart", [(0, 95, False, 0), (2, 85, False, 0), (1, 75, False, 0),
                                                              (0, 90, True, 0), (2, 80, False, 4)])
def test_jpeg_baseline(sc, tmp_path, subsampling, quality, grey, restart):
    """Baseline JPEG against Pillow's (libjpeg) decode of the same file: decoders differ by IDCT and up-sampling
    rounding; 4:2:0 files also by the chroma interpolation filter (replication here, 'fancy' there)."""
    h, w = 67, 90
    ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    img = np.stack([128 + 100 * np.sin(xs * 0.11), 128 + 90 * np.cos(ys * 0.09), 128 + 60 * np.sin((xs + ys) * 0.05)], -1)
    img = np.clip(img, 0, 255).astype(np.uint8)
    im = PIL.fromarray(img[..., 1], "L") if grey else PIL.fromarray(img, "RGB")
    p = str(tmp_path / "t.jpg")
    kw = dict(quality=quality, progressive=False, optimize=True)
    if not grey:
        kw["subsampling"] = subsampling
    if restart:
        kw["restart_marker_rows"] = restart
    im.save(p, "JPEG", **kw)
    got = sc.load_image(p)
    ref = np.asarray(PIL.open(p).convert("RGB"))
    assert got.shape == (h, w, 4) and (got[..., 3] == 255).all()
    d = np.abs(got[::-1, :, :3].astype(int) - ref.astype(int))
    print(f"jpeg subsampling {subsampling} q{quality}: mean abs diff {d.mean():.3f}, max {d.max()}")
    assert d.mean() < (0.6 if subsampling == 0 or grey else 3.0) and d.max() <= (4 if subsampling == 0 or grey else 40)


def test_progressive_jpeg_is_refused_not_misread(sc, tmp_path):
    p = str(tmp_path / "p.jpg")
    PIL.fromarray(picture()[..., :3], "RGB").save(p, "JPEG", progressive=True)
    with pytest.raises(ValueError):
        sc.load_image(p)


def test_bmp_and_tga_variants(sc, tmp_path):
    img = picture()
    rgb = np.dstack([img[..., :3], np.full(img.shape[:2], 255, np.uint8)])
    p = str(tmp_path / "t.bmp")
    PIL.fromarray(img[..., :3], "RGB").save(p)
    assert np.array_equal(sc.load_image(p), expect(rgb))
    for rle in (False, True):
        p = str(tmp_path / f"t{int(rle)}.tga")
        PIL.fromarray(img, "RGBA").save(p, compression="tga_rle" if rle else None)
        assert np.array_equal(sc.load_image(p), expect(img))
        p = str(tmp_path / f"g{int(rle)}.tga")
        PIL.fromarray(img[..., 2], "L").save(p, compression="tga_rle" if rle else None)
        g = img[..., 2]
        assert np.array_equal(sc.load_image(p), expect(np.dstack([g, g, g, np.full(g.shape, 255, np.uint8)])))
    p = str(tmp_path / "g.pgm")
    PIL.fromarray(img[..., 2], "L").save(p)
    assert np.array_equal(sc.load_image(p)[..., 0], expect(img[..., 2]))


def test_corrupt_files_fail_cleanly(sc, tmp_path):
    """Truncation at every 97th byte, and flipped bytes: the decoder returns an error (or some image), never crashes
    or reads outside the file (the sanitizer run of oracle/Makefile `sanitize` covers the same inputs under ASan)."""
    buf = io.BytesIO()
    PIL.fromarray(picture(), "RGBA").save(buf, "PNG")
    png = buf.getvalue()
    buf = io.BytesIO()
    PIL.fromarray(picture()[..., :3], "RGB").save(buf, "JPEG", quality=80)
    jpg = buf.getvalue()
    r = np.random.default_rng(1)
    for name, data in (("c.png", png), ("c.jpg", jpg)):
        for cut in range(1, len(data), 97):
            p = tmp_path / name
            p.write_bytes(data[:cut])
            try:
                sc.load_image(str(p))
            except ValueError:
                pass
        for _ in range(60):
            b = bytearray(data)
            for i in r.integers(8, len(b), 3):
                b[i] ^= int(r.integers(1, 256))
            p = tmp_path / name
            p.write_bytes(bytes(b))
            try:
                sc.load_image(str(p))
            except ValueError:
                pass


def test_headers_that_promise_more_than_the_file_holds_are_refused_at_once(sc, tmp_path):
    """Round-5 advisor: nothing is sized from header fields before the limits and the file's own length have been checked --
    a crafted 200-byte texture must fail in milliseconds without reserving gigabytes or decoding millions of blocks."""
    import time
    # JPEG: a real greyscale file, truncated after its first scan bytes, SOF dimensions patched to 65535 x 65535 / 32768 x 8192
    buf = io.BytesIO()
    PIL.fromarray(picture()[..., 2], "L").save(buf, "JPEG", quality=80)
    jpg = bytearray(buf.getvalue())
    sof = jpg.index(b"\xff\xc0")
    sos = jpg.index(b"\xff\xda")
    for dims in ((65535, 65535), (8192, 32768)):
        b = bytearray(jpg[:sos + 40])                       # header + a few bytes of entropy-coded data, no EOI
        b[sof + 5:sof + 9] = struct.pack(">HH", *dims)
        (tmp_path / "huge.jpg").write_bytes(bytes(b))
        t0 = time.time()
        with pytest.raises(ValueError):
            sc.load_image(str(tmp_path / "huge.jpg"))
        assert time.time() - t0 < 2.0
    # PNG: 32768 x 8192 RGBA16 declared, a few IDAT bytes present
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    png = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 32768, 8192, 16, 6, 0, 0, 0)) +
           chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b""))
    (tmp_path / "huge.png").write_bytes(png)
    t0 = time.time()
    with pytest.raises(ValueError):
        sc.load_image(str(tmp_path / "huge.png"))
    assert time.time() - t0 < 2.0
    # TGA / BMP / PGM headers without their pixels
    tga = bytes([0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0]) + struct.pack("<HH", 32768, 8192) + bytes([32, 0])
    bmp = b"BM" + struct.pack("<IHHI", 54, 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, 30000, 8000, 1, 24, 0, 0, 0, 0, 0, 0)
    for name, data in (("h.tga", tga), ("h.bmp", bmp), ("h.pgm", b"P5 32768 8192 255\n")):
        (tmp_path / name).write_bytes(data)
        t0 = time.time()
        with pytest.raises(ValueError):
            sc.load_image(str(tmp_path / name))
        assert time.time() - t0 < 2.0


def test_single_component_jpeg_ignores_its_sampling_factors(sc, tmp_path):
    """T.81 A.2.2: a scan with one component is not interleaved -- its MCU is one 8 x 8 block whatever H and V say.  The same
    greyscale file with its SOF factors patched from 1x1 to 2x2 must decode to the same image (libjpeg agrees)."""
    h, w = 45, 70
    ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    g = np.clip(128 + 100 * np.sin(xs * 0.13) * np.cos(ys * 0.07), 0, 255).astype(np.uint8)
    buf = io.BytesIO()
    PIL.fromarray(g, "L").save(buf, "JPEG", quality=90)
    jpg = bytearray(buf.getvalue())
    (tmp_path / "a.jpg").write_bytes(bytes(jpg))
    sof = jpg.index(b"\xff\xc0")
    assert jpg[sof + 9] == 1 and jpg[sof + 11] == 0x11        # one component, 1x1
    jpg[sof + 11] = 0x22
    (tmp_path / "b.jpg").write_bytes(bytes(jpg))
    a, b = sc.load_image(str(tmp_path / "a.jpg")), sc.load_image(str(tmp_path / "b.jpg"))
    assert a.shape == (h, w, 4) and np.array_equal(a, b)
    ref = np.asarray(PIL.open(str(tmp_path / "b.jpg")).convert("L"))
    assert np.abs(b[::-1, :, 0].astype(int) - ref.astype(int)).max() <= 4


def test_image_load_never_writes_past_the_callers_capacity(sc, tmp_path):
    """vcth_image_load's second call reports -2 instead of copying an image that outgrew the buffer sized by the first call."""
    import ctypes as C
    p = str(tmp_path / "t.png")
    PIL.fromarray(picture(), "RGBA").save(p)
    lib = sc._lib
    w, h = C.c_int32(), C.c_int32()
    assert lib.vcth_image_load(os.fsencode(p), C.byref(w), C.byref(h), None, 0) == 0
    n = w.value * h.value * 4
    guard = np.full(n + 64, 0xAB, np.uint8)
    assert lib.vcth_image_load(os.fsencode(p), None, None, guard.ctypes.data, n - 4) == -2
    assert (guard == 0xAB).all()
    assert lib.vcth_image_load(os.fsencode(p), None, None, guard.ctypes.data, n) == 0
    assert (guard[n:] == 0xAB).all() and np.array_equal(guard[:n].reshape(h.value, w.value, 4), expect(picture()))


def test_mtl_maps_in_png_and_jpeg(sc, tmp_path):
    """The OBJ + MTL reader takes the new containers (by content, whatever the extension says)."""
    img = picture(16, 16)
    PIL.fromarray(img, "RGBA").save(str(tmp_path / "kd.png"))
    PIL.fromarray(img[..., :3], "RGB").save(str(tmp_path / "ks.jpg"), quality=95, subsampling=0)
    PIL.fromarray(img[..., 2], "L").save(str(tmp_path / "bump.tga"), "PNG")       # a PNG named .tga
    (tmp_path / "s.mtl").write_text("newmtl m\nKd 1 1 1\nmap_Kd kd.png\nmap_Ks ks.jpg\nmap_bump bump.tga\n")
    (tmp_path / "s.obj").write_text("mtllib s.mtl\nv 0 0 0\nv 100 0 0\nv 100 100 0\nvt 0 0\nvt 1 0\nvt 1 1\nvn 0 0 1\n"
                                    "usemtl m\nf 1/1/1 2/2/1 3/3/1\n")
    s = sc.Scene(str(tmp_path / "s.obj"))
    assert s.mat_tex.tolist() == [[0, 1, 2]] and len(s.textures) == 3
    assert np.array_equal(s.textures[0], expect(img))
    assert np.abs(s.textures[1][::-1, :, :3].astype(int) - img[..., :3].astype(int)).mean() < 6.0
    assert np.array_equal(s.textures[2][..., 0], expect(img[..., 2]))
