"""Seeded synthetic inputs for the parity tests and the bench.

Two levels: coefficient-level (what crosses the Worker boundary) and adversarial blocks
that exercise the wrap-around semantics of src/idct.rs."""
import numpy as np

# Annex K tables in natural order; quality scaling as libjpeg's jpeg_quality_scaling
_LUMA = np.array([
    16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
    14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99])
_CHROMA = np.array([
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
    47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99])


def quality_tables(quality=85):
    scale = 5000 // quality if quality < 50 else 200 - 2 * quality
    lum = np.clip((_LUMA * scale + 50) // 100, 1, 255).astype(np.uint16)
    chr_ = np.clip((_CHROMA * scale + 50) // 100, 1, 255).astype(np.uint16)
    return lum, chr_


def synthetic_rgb(width, height, seed=0x5EED):
    """I(x,y,c) = clamp(128 + 96 sin(x/37 + c) cos(y/29) + noise), noise uniform in [-12, 12]."""
    rng = np.random.default_rng(seed)
    x = np.arange(width, dtype=np.float64)[None, :, None]
    y = np.arange(height, dtype=np.float64)[:, None, None]
    c = np.arange(3, dtype=np.float64)[None, None, :]
    img = 128.0 + 96.0 * np.sin(x / 37.0 + c) * np.cos(y / 29.0) + rng.uniform(-12, 12, (height, width, 3))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def _fdct_quantize(plane, block_w, block_h, qt):
    """plane (H, W) float, already level-shifted; returns block-raster natural-order i16."""
    from scipy.fft import dctn
    ph, pw = block_h * 8, block_w * 8
    h, w = plane.shape
    padded = np.pad(plane, ((0, ph - h), (0, pw - w)), mode="edge")
    blocks = padded.reshape(block_h, 8, block_w, 8).transpose(0, 2, 1, 3)
    coefs = dctn(blocks, axes=(2, 3), norm="ortho")
    q = np.rint(coefs / qt.reshape(8, 8).astype(np.float64))
    return np.clip(q, -2047, 2047).astype(np.int16).reshape(-1)


def coefficients_from_rgb(rgb, comps, mode, qts):
    """Forward path of a baseline encoder (colour transform, box subsampling, FDCT, quantise)
    producing per-component coefficient planes in the Worker layout.
    mode: 'ycbcr' (3 comps), 'gray', 'cmyk' (Adobe-style inverted C, M, Y from the primaries, K from luma) or 'ycck' (Y, Cb, Cr
    and K = luma); every component is box-subsampled by its own sampling factors against the frame's largest ones.  The
    planes are those of the FULL-size frame whatever the components' dct_scale (a scaled decode reads the same coefficients)."""
    r, g, b = [rgb[..., i].astype(np.float64) for i in range(3)]
    y = 0.299 * r + 0.587 * g + 0.114 * b
    cb = -0.168736 * r - 0.331264 * g + 0.5 * b + 128.0
    cr = 0.5 * r - 0.418688 * g - 0.081312 * b + 128.0
    planes = {"gray": (y,), "cmyk": (r, g, b, y), "ycck": (y, cb, cr, y), "ycbcr": (y, cb, cr)}[mode]
    h_max = max(c.horizontal_sampling_factor for c in comps)
    v_max = max(c.vertical_sampling_factor for c in comps)
    height, width = rgb.shape[:2]
    out = []
    for c, plane in zip(comps, planes):
        fx, fy = h_max // c.horizontal_sampling_factor, v_max // c.vertical_sampling_factor
        p = plane
        if len(comps) > 1 and (fx > 1 or fy > 1):
            hh, ww = p.shape
            p = np.pad(p, ((0, (-hh) % fy), (0, (-ww) % fx)), mode="edge")
            p = p.reshape(p.shape[0] // fy, fy, p.shape[1] // fx, fx).mean(axis=(1, 3))
        if len(comps) > 1:  # ceil(W * h / h_max) x ceil(H * v / v_max): component.size at dct_scale 8 (src/parser.rs:292-310)
            p = p[: -(-height * c.vertical_sampling_factor // v_max), : -(-width * c.horizontal_sampling_factor // h_max)]
        p = p[: c.block_height * 8, : c.block_width * 8]
        out.append(_fdct_quantize(p - 128.0, c.block_width, c.block_height, qts[len(out)]))
    return out


def sparse_coefficients(rng, n_blocks, density=0.15, amp=96, dc_amp=700):
    """JPEG-like random blocks: mostly-zero AC decaying with frequency, large DC."""
    c = rng.integers(-amp, amp + 1, (n_blocks, 64))
    mask = rng.random((n_blocks, 64)) < density * np.linspace(2.0, 0.2, 64)[None, :]
    c = c * mask
    c[:, 0] = rng.integers(-dc_amp, dc_amp + 1, n_blocks)
    # a share of blocks with fully empty rows / columns / DC-only (short-cut paths)
    k = n_blocks // 8
    if k:
        c[rng.choice(n_blocks, k, replace=False), 1:] = 0
        idx = rng.choice(n_blocks, k, replace=False)
        c[idx.reshape(-1, 1), np.arange(8, 64)[None, :]] = 0
    return c.astype(np.int16).reshape(-1)


def adversarial_blocks(rng, n_random=256):
    """Blocks that only agree with the reference if i32 wrap-around and the column DC-only
    short-cut (src/idct.rs:279-295) are reproduced exactly."""
    blocks, qts = [], []
    blocks.append(np.full(64, 32767, np.int16)); qts.append(np.full(64, 65535, np.uint16))   # saturated KAT
    blocks.append(np.full(64, -32768, np.int16)); qts.append(np.full(64, 65535, np.uint16))
    for pos in range(0, 8):  # huge first-row coefficient, zeros below (column short-cut under wrap)
        for val, qv in ((20000, 60000), (-32768, 65535), (32767, 40000)):
            c = np.zeros(64, np.int16); c[pos] = val
            q = np.ones(64, np.uint16); q[pos] = qv
            blocks.append(c); qts.append(q)
    for pos in (8, 16, 24, 32, 40, 48, 56, 9, 63):  # one huge AC elsewhere
        c = np.zeros(64, np.int16); c[pos] = 32767; c[0] = -32768
        blocks.append(c); qts.append(np.full(64, 65535, np.uint16))
    return blocks, qts


def tight_blocks(rng, n, q):
    """Blocks at the edge of the TIGHT class: every column's sum of |c*q| is (nearly) as large as allowed
    (5900), spread over a random subset of the column's rows; one block in eight puts it all on one row."""
    qm = q.astype(np.int64).reshape(1, 8, 8)
    w = rng.random((n, 8, 8)) * (rng.random((n, 8, 8)) < 0.6)
    one = rng.integers(0, 8, (n, 1, 8))
    single = (np.arange(8).reshape(1, 8, 1) == one) & (rng.random((n, 1, 1)) < 0.125)
    w = np.where(single.any(axis=1, keepdims=True), single.astype(np.float64), w)
    tot = w.sum(axis=1, keepdims=True)
    tot[tot == 0] = 1.0
    mag = np.floor(w / tot * 5900.0 / qm).astype(np.int64)
    c = mag * rng.choice([-1, 1], (n, 8, 8))
    assert (np.abs(c * qm).sum(axis=1) <= 5900).all() and (np.abs(c * qm) < (1 << 15)).all()
    return c.astype(np.int16).reshape(-1)
