"""Generates the texture-decoding fixtures: small image files written with Pillow (an encoder only: nothing of it is used at test time)
and, next to each, the RGB8 pixels the REFERENCE's own stb_image returns for it — stbi_load(file, &w, &h, &c, 3), exactly the call of
OglScene::load_texture (src/Tracer/OglScene.cpp:26-34) — obtained from oracle/_ref/adypt_ref (`imgload`), i.e. from dep/stb_image.h
compiled in place by oracle/Makefile.  Run in the build container (needs /root/reference and Pillow):

    python tests/golden/make_golden_images.py

Commits: tests/golden/images/<name>.<ext> (input data) + <name>.rgb8 (expected output) + index.json.  JPEG variants cover what a
decoder can get wrong without noticing: chroma sub-sampling 4:4:4 / 4:2:2 / 4:2:0 (+ 4:4:0, 4:1:1 written by hand-set sampling),
odd sizes (edge replication of the up-sampling filters, partial MCUs), progressive scans with successive approximation, restart
intervals, grey, CMYK (Adobe marker), high and low quality (coefficient range, 16-bit wrap of dequantised values)."""
import io
import json
import os
import subprocess
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "images")
REF = os.path.join(HERE, "..", "..", "oracle", "_ref", "adypt_ref")


def picture(w, h, seed):
    """smooth gradients + hard edges + noise: exercises DC prediction, AC runs and clamping"""
    rs = np.random.RandomState(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, 3), np.float64)
    img[..., 0] = 127 + 120 * np.sin(x / 5.0 + seed) * np.cos(y / 7.0)
    img[..., 1] = (x * 255.0 / max(1, w - 1))
    img[..., 2] = (y * 255.0 / max(1, h - 1))
    img[(x // 6 + y // 5) % 2 == 0] *= 0.35
    img[h // 3:h // 3 + 3, :, :] = [255, 0, 255]
    img += rs.normal(scale=12.0, size=img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    os.makedirs(OUT, exist_ok=True)
    index = {}

    def emit(name, ext, data):
        path = os.path.join(OUT, name + "." + ext)
        with open(path, "wb") as f:
            f.write(data)
        raw = os.path.join(OUT, name + ".rgb8")
        r = subprocess.run([REF, "imgload", path, raw], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, (name, r.stderr.decode())
        w, h = map(int, r.stdout.split()[:2])
        assert os.path.getsize(raw) == w * h * 3
        index[name + "." + ext] = {"w": w, "h": h}

    def jpeg(name, arr, mode="RGB", **kw):
        b = io.BytesIO()
        Image.fromarray(arr, mode).save(b, "JPEG", **kw)
        emit(name, "jpg", b.getvalue())

    a = picture(45, 29, 1)
    jpeg("j444_q90", a, quality=90, subsampling=0)
    jpeg("j422_q85", a, quality=85, subsampling=1)
    jpeg("j420_q75", a, quality=75, subsampling=2)
    jpeg("j420_q20", a, quality=20, subsampling=2)
    jpeg("j420_q100", a, quality=100, subsampling=2)
    jpeg("j444_progressive", a, quality=80, subsampling=0, progressive=True)
    jpeg("j420_progressive", picture(37, 50, 2), quality=70, subsampling=2, progressive=True)
    jpeg("j420_restart", picture(64, 48, 3), quality=80, subsampling=2, restart_marker_blocks=3)
    jpeg("j422_restart_rows", picture(33, 17, 4), quality=60, subsampling=1, restart_marker_rows=1)
    jpeg("j420_progressive_restart", picture(40, 40, 5), quality=65, subsampling=2, progressive=True, restart_marker_blocks=2)
    jpeg("j_1x1", picture(1, 1, 6), quality=90, subsampling=2)
    jpeg("j_17x1", picture(17, 1, 7), quality=90, subsampling=2)
    jpeg("j_1x17", picture(1, 17, 8), quality=90, subsampling=1)
    jpeg("j_grey", picture(31, 23, 9)[..., 0], mode="L", quality=80)
    jpeg("j_grey_progressive", picture(24, 24, 10)[..., 1], mode="L", quality=50, progressive=True)
    cmyk = np.concatenate([picture(20, 14, 11), picture(20, 14, 12)[..., :1]], axis=-1)
    jpeg("j_cmyk", cmyk, mode="CMYK", quality=85)
    jpeg("j_optimized_tables", picture(52, 36, 13), quality=88, subsampling=2, optimize=True)
    # sampling factors Pillow's presets do not offer (4:4:0 = luma 1x2, 4:1:1 = luma 4x1, unequal chroma factors, 3x / 4x ratios that take
    # stb_image's nearest-neighbour path): written by the small baseline encoder below, with the tables of a Pillow file
    donor = open(os.path.join(OUT, "j444_q90.jpg"), "rb").read()
    for nm, samp in (("j440", ((1, 2), (1, 1), (1, 1))), ("j411", ((4, 1), (1, 1), (1, 1))), ("j_cb_cr_differ", ((2, 2), (2, 1), (1, 1))),
                     ("j_h3", ((3, 1), (1, 1), (1, 1))), ("j_v4", ((1, 4), (1, 1), (1, 2))), ("j_h4_h2", ((4, 1), (2, 1), (1, 1)))):
        emit(nm, "jpg", encode_baseline(picture(43, 31, 14), samp, donor, restart=5 if nm == "j411" else 0))
    # other container formats the loader reads, pinned the same way
    def save(name, ext, arr, mode, fmt, **kw):
        b = io.BytesIO()
        Image.fromarray(arr, mode).save(b, fmt, **kw)
        emit(name, ext, b.getvalue())
    p = picture(23, 19, 20)
    save("p_rgb", "png", p, "RGB", "PNG")
    save("p_rgba", "png", np.concatenate([p, picture(23, 19, 21)[..., :1]], -1), "RGBA", "PNG")
    save("p_grey", "png", p[..., 0], "L", "PNG")
    save("p_grey_alpha", "png", p[..., :2], "LA", "PNG")
    b = io.BytesIO(); Image.fromarray(p, "RGB").quantize(64).save(b, "PNG"); emit("p_palette", "png", b.getvalue())
    b = io.BytesIO(); Image.fromarray((p[..., 0].astype(np.uint16) * 257), "I;16").save(b, "PNG"); emit("p_grey16", "png", b.getvalue())
    for bits, colours in ((1, 2), (2, 4), (4, 16)):
        b = io.BytesIO(); Image.fromarray(p, "RGB").quantize(colours).save(b, "PNG", bits=bits); emit("p_palette%d" % bits, "png", b.getvalue())
    b = io.BytesIO(); Image.fromarray(p[..., 0] > 120).save(b, "PNG"); emit("p_grey1", "png", b.getvalue())
    for bits in (2, 4):   # grey at 2 and 4 bits: written by hand (filter 0), Pillow only writes them as palettes
        emit("p_grey%d" % bits, "png", raw_png(p[..., 1] >> (8 - bits), bits, 0, interlace=False))
    for nm, arr, depth, ctype in (("p_rgb_adam7", p, 8, 2), ("p_grey_adam7", p[..., 2], 8, 0), ("p_grey4_adam7", p[..., 0] >> 4, 4, 0),
                                  ("p_rgba16_adam7", np.concatenate([p, p[..., :1]], -1).astype(np.uint16) * 257 + 3, 16, 6)):
        emit(nm, "png", raw_png(arr, depth, ctype, interlace=True))
    emit("p_adam7_1x1", "png", raw_png(p[:1, :1], 8, 2, interlace=True))
    emit("p_adam7_3x5", "png", raw_png(p[:5, :3], 8, 2, interlace=True))
    save("b_rgb", "bmp", p, "RGB", "BMP")
    save("b_grey8", "bmp", p[..., 1], "L", "BMP")
    b = io.BytesIO(); Image.fromarray(p, "RGB").quantize(50).save(b, "BMP"); emit("b_palette8", "bmp", b.getvalue())
    # 4-bit palettised and top-down 32-bit: by hand
    import struct
    q16 = np.asarray(Image.fromarray(p, "RGB").quantize(16))
    pal16 = np.asarray(Image.fromarray(p, "RGB").quantize(16).getpalette()[:48], np.uint8).reshape(16, 3)
    hh, ww = q16.shape
    rows = b""
    for yy in range(hh - 1, -1, -1):
        r = q16[yy].astype(np.uint8)
        if ww % 2: r = np.append(r, 0)
        packed = (r[0::2] << 4 | r[1::2]).astype(np.uint8).tobytes()
        rows += packed + b"\0" * (-len(packed) % 4)
    palb = b"".join(bytes([c[2], c[1], c[0], 0]) for c in pal16)
    off = 14 + 40 + len(palb)
    emit("b_palette4", "bmp", b"BM" + struct.pack("<IHHI", off + len(rows), 0, 0, off) + struct.pack("<IiiHHIIiiII", 40, ww, hh, 1, 4, 0, len(rows), 2835, 2835, 16, 0) + palb + rows)
    bgra = np.concatenate([p[..., ::-1], np.full(p.shape[:2] + (1,), 255, np.uint8)], -1)
    emit("b_rgb32_topdown", "bmp", b"BM" + struct.pack("<IHHI", 54 + bgra.size, 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, ww, -hh, 1, 32, 0, bgra.size, 2835, 2835, 0, 0) + bgra.tobytes())
    save("t_rgb", "tga", p, "RGB", "TGA")
    save("t_rgb_rle", "tga", p, "RGB", "TGA", compression="tga_rle")
    save("t_rgba", "tga", np.concatenate([p, picture(23, 19, 22)[..., :1]], -1), "RGBA", "TGA")
    save("t_grey", "tga", p[..., 2], "L", "TGA")
    b = io.BytesIO(); Image.fromarray(p, "RGB").quantize(40).save(b, "TGA"); emit("t_palette", "tga", b.getvalue())
    b = io.BytesIO(); Image.fromarray(p, "RGB").quantize(200).save(b, "TGA", compression="tga_rle"); emit("t_palette_rle", "tga", b.getvalue())
    # 16-bit 5-5-5 true colour, 16-bit grey + alpha, 16-bit palette entries: written by hand (18-byte header, bottom-up rows)
    def tga(type_, bpp, body, cmap=b"", pal_len=0, pal_bits=0, desc=0):
        hh, ww = p.shape[:2]
        return bytes([0, 1 if cmap else 0, type_, 0, 0, pal_len & 255, pal_len >> 8, pal_bits, 0, 0, 0, 0, ww & 255, ww >> 8, hh & 255, hh >> 8, bpp, desc]) + cmap + body
    v555 = ((p[..., 0].astype(np.uint16) >> 3) << 10 | (p[..., 1].astype(np.uint16) >> 3) << 5 | (p[..., 2].astype(np.uint16) >> 3))
    emit("t_rgb555", "tga", tga(2, 16, v555[::-1].astype("<u2").tobytes()))
    emit("t_rgb555_top", "tga", tga(2, 15, v555.astype("<u2").tobytes(), desc=0x20))
    emit("t_grey_alpha16", "tga", tga(3, 16, np.stack([p[..., 0], p[..., 1]], -1)[::-1].tobytes()))
    pal = (np.arange(64, dtype=np.uint16) * 511 % 32768).astype("<u2")
    emit("t_palette16", "tga", tga(1, 8, (p[..., 2] >> 2)[::-1].astype(np.uint8).tobytes(), cmap=pal.tobytes(), pal_len=64, pal_bits=16))
    emit("t_palette_overrun", "tga", tga(1, 8, (p[..., 2])[::-1].astype(np.uint8).tobytes(), cmap=bytes(range(30)) , pal_len=10, pal_bits=24))
    with open(os.path.join(OUT, "index.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)
    print(len(index), "fixtures,", sum(os.path.getsize(os.path.join(OUT, n)) for n in os.listdir(OUT)), "bytes")


def raw_png(arr, depth, ctype, interlace):
    """PNG writer for the variants Pillow cannot produce: any depth / colour type, Adam7, filter type 0 on every row."""
    import struct
    import zlib
    arr = np.asarray(arr)
    if arr.ndim == 2:
        arr = arr[..., None]
    h, w, ch = arr.shape

    def pack_rows(a):
        rows = bytearray()
        for row in a:
            rows.append(0)
            flat = row.reshape(-1)
            if depth == 8:
                rows.extend(flat.astype(np.uint8).tobytes())
            elif depth == 16:
                rows.extend(flat.astype(">u2").tobytes())
            else:
                bits = "".join(format(int(v), "0%db" % depth) for v in flat)
                bits += "0" * (-len(bits) % 8)
                rows.extend(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))
        return bytes(rows)

    if interlace:
        xo, yo, xs, ys = (0, 4, 0, 2, 0, 1, 0), (0, 0, 4, 0, 2, 0, 1), (8, 8, 4, 4, 2, 2, 1), (8, 8, 8, 4, 4, 2, 2)
        data = b"".join(pack_rows(arr[yo[k]::ys[k], xo[k]::xs[k]]) for k in range(7) if arr[yo[k]::ys[k], xo[k]::xs[k]].size)
    else:
        data = pack_rows(arr)

    def chunk(tag, payload):
        return struct.pack(">I", len(payload)) + tag + payload + struct.pack(">I", zlib.crc32(tag + payload) & 0xffffffff)

    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0)) +
            chunk(b"IDAT", zlib.compress(data)) + chunk(b"IEND", b""))


# ---- a minimal baseline JPEG ENCODER (fixture generation only) with arbitrary sampling factors --------------------------------------
def _segments(jpg):
    """marker segments of a JPEG file up to SOS: {marker: [payload, ...]}"""
    out, i = {}, 2
    while i < len(jpg):
        assert jpg[i] == 0xFF
        m = jpg[i + 1]
        n = jpg[i + 2] << 8 | jpg[i + 3]
        out.setdefault(m, []).append(jpg[i + 4:i + 2 + n])
        if m == 0xDA:
            break
        i += 2 + n
    return out


_ZIG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
        35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def encode_baseline(rgb, sampling, donor, restart=0):
    from scipy.fft import dctn
    seg = _segments(donor)
    qt, ht = {}, {}
    for payload in seg[0xDB]:
        i = 0
        while i < len(payload):
            assert payload[i] >> 4 == 0
            qt[payload[i] & 15] = np.array(list(payload[i + 1:i + 65]), np.float64)  # zigzag order
            i += 65
    for payload in seg[0xC4]:
        i = 0
        while i < len(payload):
            tc_th, counts = payload[i], list(payload[i + 1:i + 17])
            n = sum(counts)
            vals = list(payload[i + 17:i + 17 + n])
            codes, code, k = {}, 0, 0
            for length in range(1, 17):
                for _ in range(counts[length - 1]):
                    codes[vals[k]] = (code, length)
                    code += 1
                    k += 1
                code <<= 1
            ht[tc_th] = codes
            i += 17 + n
    h, w = rgb.shape[:2]
    r, g, b = [rgb[..., k].astype(np.float64) for k in range(3)]
    planes = [0.299 * r + 0.587 * g + 0.114 * b, 128 - 0.168736 * r - 0.331264 * g + 0.5 * b, 128 + 0.5 * r - 0.418688 * g - 0.081312 * b]
    hmax, vmax = max(s[0] for s in sampling), max(s[1] for s in sampling)
    mcu_x, mcu_y = -(-w // (8 * hmax)), -(-h // (8 * vmax))
    comps = []
    for (ch, cv), pl in zip(sampling, planes):
        fx, fy = hmax // ch, vmax // cv
        assert hmax % ch == 0 and vmax % cv == 0
        pl = np.pad(pl, ((0, mcu_y * 8 * vmax - h), (0, mcu_x * 8 * hmax - w)), mode="edge")
        pl = pl.reshape(pl.shape[0] // fy, fy, pl.shape[1] // fx, fx).mean(axis=(1, 3))
        comps.append(pl)
    bits = []

    def put(code, length):
        bits.append((code, length))

    def category(v):
        return 0 if v == 0 else int(abs(v)).bit_length()

    def put_value(v, cat):
        if cat:
            put(v if v >= 0 else v + (1 << cat) - 1, cat)

    out = bytearray(b"\xff\xd8")

    def segment(marker, payload):
        out.extend(bytes([0xFF, marker]) + (len(payload) + 2).to_bytes(2, "big") + payload)

    segment(0xE0, b"JFIF\0\x01\x01\0\0\x01\0\x01\0\0")
    for payload in seg[0xDB]:
        segment(0xDB, payload)
    sof = bytes([8]) + h.to_bytes(2, "big") + w.to_bytes(2, "big") + bytes([3])
    for i, (ch, cv) in enumerate(sampling):
        sof += bytes([i + 1, ch << 4 | cv, 0 if i == 0 else 1])
    segment(0xC0, sof)
    for payload in seg[0xC4]:
        segment(0xC4, payload)
    if restart:
        segment(0xDD, restart.to_bytes(2, "big"))
    segment(0xDA, bytes([3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0]))

    def flush():
        acc, n, data = 0, 0, bytearray()
        for code, length in bits:
            acc = acc << length | code
            n += length
            while n >= 8:
                byte = acc >> (n - 8) & 0xFF
                data.append(byte)
                if byte == 0xFF:
                    data.append(0)
                n -= 8
        if n:
            byte = (acc << (8 - n) | (1 << (8 - n)) - 1) & 0xFF
            data.append(byte)
            if byte == 0xFF:
                data.append(0)
        bits.clear()
        return data

    pred, count, rst = [0, 0, 0], 0, 0
    for my in range(mcu_y):
        for mx in range(mcu_x):
            for ci, ((ch, cv), pl) in enumerate(zip(sampling, comps)):
                q = qt[0 if ci == 0 else 1]
                dc_t, ac_t = ht[0x00 if ci == 0 else 0x01], ht[0x10 if ci == 0 else 0x11]
                for by in range(cv):
                    for bx in range(ch):
                        blk = pl[(my * cv + by) * 8:(my * cv + by) * 8 + 8, (mx * ch + bx) * 8:(mx * ch + bx) * 8 + 8] - 128.0
                        co = dctn(blk, norm="ortho").reshape(64)[_ZIG]
                        qz = np.rint(co / q).astype(int)
                        d = int(qz[0]) - pred[ci]
                        pred[ci] = int(qz[0])
                        cat = category(d)
                        put(*dc_t[cat]); put_value(d, cat)
                        run = 0
                        last = max([k for k in range(1, 64) if qz[k] != 0], default=0)
                        for k in range(1, last + 1):
                            v = int(qz[k])
                            if v == 0:
                                run += 1
                                continue
                            while run > 15:
                                put(*ac_t[0xF0]); run -= 16
                            cat = category(v)
                            put(*ac_t[run << 4 | cat]); put_value(v, cat)
                            run = 0
                        if last < 63:
                            put(*ac_t[0x00])
            count += 1
            if restart and count % restart == 0 and not (my == mcu_y - 1 and mx == mcu_x - 1):
                out.extend(flush())
                out.extend(bytes([0xFF, 0xD0 + rst % 8]))
                rst += 1
                pred = [0, 0, 0]
    out.extend(flush())
    out.extend(b"\xff\xd9")
    return bytes(out)


if __name__ == "__main__":
    main()
