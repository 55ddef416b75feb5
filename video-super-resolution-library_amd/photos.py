"""Real picture content for tests, certification campaigns and bench legs: the photographs that ship INSIDE the Python
packages of this image (nothing is downloaded, nothing is copied into the repository).

The reference's own test assets are real clips (test/validation_suite/run_tests_avxout.sh:13,50-107) and its published numbers are
measured on one (docs/performance.md:14); neither is available offline.  What is: scikit-learn's `china.jpg` / `flower.jpg`,
matplotlib's `grace_hopper.jpg`, and -- in the image's conda tree -- scipy's `face.dat` / `ascent.dat` and scikit-image's sample
pictures (astronaut, camera, chelsea, coffee, rocket, hubble deep field, retina, brick / grass / gravel textures, coins, moon,
scanned page, text).  They are found by path (no heavy imports); a source that is absent is skipped, `available()` says what is left.

`photo_y(width, height, bits, index)` turns source `index % n` into one luma plane of the requested size:
BT.601 limited-range luma quantised DIRECTLY to the bit depth (a 10-bit frame of an RGB source has genuine low-order bits),
mirror-tiled (no artificial seams) from an index-dependent origin and cropped.  `index // n` walks through the VARIANTS:
  plain      the photograph as decoded
  jpeg30     re-encoded as JPEG quality 30 first: 8x8 block edges, ringing, flat blocks (what a low-bitrate source looks like)
  letterbox  black bars (video black) above and below a 2.39:1 picture area, and pillar bars for portrait sources
  soft2x     the photograph enlarged 2x with a bicubic kernel before tiling: the soft, band-limited content an upscaler is usually fed
  mosaic     the frame as a grid of DIFFERENT sources, hard seams between them (picture-in-picture / multi-view layouts)
Pure numpy + PIL; no dependency on the oracle or on the HIP library.
"""
import bz2
import glob
import importlib.util
import io
import os
import pickle

import numpy as np

VARIANTS = ("plain", "jpeg30", "letterbox", "soft2x", "mosaic")
_CACHE = {}


def _pkg_dir(name):
    try:
        spec = importlib.util.find_spec(name)
    except (ImportError, ValueError):
        return None
    if spec is None or not spec.origin:
        return None
    return os.path.dirname(spec.origin)


def _candidates():
    """name -> path of every picture this image carries (first hit per name wins)."""
    out = {}

    def add(name, path):
        if name not in out and os.path.isfile(path):
            out[name] = path
    sk = _pkg_dir("sklearn")
    roots = [os.path.join(sk, "datasets", "images")] if sk else []
    roots += sorted(glob.glob("/opt/conda/lib/python3*/site-packages/sklearn/datasets/images"))
    for r in roots:
        add("china", os.path.join(r, "china.jpg"))
        add("flower", os.path.join(r, "flower.jpg"))
    mp = _pkg_dir("matplotlib")
    roots = [os.path.join(mp, "mpl-data", "sample_data")] if mp else []
    roots += sorted(glob.glob("/opt/conda/lib/python3*/site-packages/matplotlib/mpl-data/sample_data"))
    for r in roots:
        add("grace_hopper", os.path.join(r, "grace_hopper.jpg"))
    sp = _pkg_dir("scipy")
    roots = ([os.path.join(sp, "misc"), os.path.join(sp, "datasets")] if sp else []) + sorted(glob.glob("/opt/conda/lib/python3*/site-packages/scipy/misc"))
    for r in roots:
        add("face", os.path.join(r, "face.dat"))
        add("ascent", os.path.join(r, "ascent.dat"))
    si = _pkg_dir("skimage")
    roots = [os.path.join(si, "data")] if si else []
    roots += sorted(glob.glob("/opt/conda/lib/python3*/site-packages/skimage/data"))
    for r in roots:
        for stem in ("astronaut.png", "camera.png", "chelsea.png", "coffee.png", "rocket.jpg", "hubble_deep_field.jpg", "retina.jpg",
                     "brick.png", "grass.png", "gravel.png", "coins.png", "moon.png", "page.png", "text.png", "motorcycle_left.png", "ihc.png"):
            add(stem.split(".")[0], os.path.join(r, stem))
    return out


def available():
    """Names of the photographs found in this image, in a fixed order."""
    if "names" not in _CACHE:
        try:
            import PIL  # noqa: F401
            _CACHE["paths"] = _candidates()
        except ImportError:                                   # without a decoder only the raw scipy arrays are readable
            _CACHE["paths"] = {k: v for k, v in _candidates().items() if v.endswith(".dat")}
        _CACHE["names"] = sorted(_CACHE["paths"])
    return list(_CACHE["names"])


def _decode(name):
    """-> float64 array (h, w, 3) RGB or (h, w) gray, values 0..255, as stored in the file."""
    path = _CACHE["paths"][name]
    if path.endswith("face.dat"):
        return np.frombuffer(bz2.decompress(open(path, "rb").read()), np.uint8).reshape(768, 1024, 3).astype(np.float64)
    if path.endswith("ascent.dat"):
        return np.array(pickle.load(open(path, "rb")), dtype=np.float64)
    from PIL import Image
    im = Image.open(path)
    if im.mode not in ("L", "RGB"):
        im = im.convert("RGB" if im.mode in ("RGBA", "P", "CMYK") else "L")
    return np.asarray(im, dtype=np.float64)


def _rgb_u8(name):
    a = _decode(name)
    return np.clip(np.rint(a), 0, 255).astype(np.uint8)


def _to_luma(a, bits):
    """BT.601 limited-range luma of an RGB / gray array (0..255), quantised to `bits`: 8-bit [16, 235], scaled by 2^(bits-8) above."""
    lin = a if a.ndim == 2 else 0.299 * a[..., 0] + 0.587 * a[..., 1] + 0.114 * a[..., 2]
    sc = float(1 << (bits - 8))
    y = np.rint((16.0 + 219.0 * lin / 255.0) * sc)
    return np.clip(y, 16 * sc, 235 * sc).astype(np.uint8 if bits == 8 else np.uint16)


def luma(name, bits=8, variant="plain"):
    """Luma plane of one photograph at its native size (2x for `soft2x`), after the variant's processing of the picture itself."""
    available()
    key = (name, bits, variant)
    if key in _CACHE:
        return _CACHE[key]
    if variant == "jpeg30":
        from PIL import Image
        buf = io.BytesIO()
        Image.fromarray(_rgb_u8(name)).save(buf, format="JPEG", quality=30)
        a = np.asarray(Image.open(io.BytesIO(buf.getvalue())), dtype=np.float64)
    elif variant == "soft2x":
        from PIL import Image
        im = Image.fromarray(_rgb_u8(name))
        a = np.asarray(im.resize((im.width * 2, im.height * 2), Image.BICUBIC), dtype=np.float64)
    else:
        a = _decode(name)
    y = _to_luma(a, bits)
    _CACHE[key] = y
    return y


def _mirror_tile(src, width, height, ox=0, oy=0):
    """`src` reflected about its edges to fill width x height, starting at (ox, oy) of the infinite mirror-tiled plane."""
    h, w = src.shape
    yy = (np.arange(height) + oy) % (2 * h)
    xx = (np.arange(width) + ox) % (2 * w)
    yy = np.where(yy < h, yy, 2 * h - 1 - yy)
    xx = np.where(xx < w, xx, 2 * w - 1 - xx)
    return src[yy[:, None], xx[None, :]]


def photo_y(width, height, bits=8, index=0):
    """One luma plane of real picture content (see the module docstring); deterministic in (width, height, bits, index)."""
    names = available()
    if not names:
        raise RuntimeError("no photographs found in this image (sklearn / matplotlib / scipy / skimage sample pictures)")
    n = len(names)
    name = names[index % n]
    variant = VARIANTS[(index // n) % len(VARIANTS)]
    black = (16 << (bits - 8))
    g = np.random.Generator(np.random.PCG64(9000 + index))
    if variant == "mosaic":
        cols, rows = (4, 3) if width >= 1920 else (2, 2)
        out = np.empty((height, width), np.uint8 if bits == 8 else np.uint16)
        xs = np.linspace(0, width, cols + 1).astype(int)
        ys = np.linspace(0, height, rows + 1).astype(int)
        for r in range(rows):
            for c in range(cols):
                src = luma(names[(index + r * cols + c) % n], bits, "plain")
                out[ys[r]:ys[r + 1], xs[c]:xs[c + 1]] = _mirror_tile(src, xs[c + 1] - xs[c], ys[r + 1] - ys[r],
                                                                      int(g.integers(0, src.shape[1])), int(g.integers(0, src.shape[0])))
        return out
    src = luma(name, bits, "plain" if variant == "letterbox" else variant)
    if variant == "letterbox":
        out = np.full((height, width), black, src.dtype)
        ph = min(height, int(round(width / 2.39)) & ~1)
        pw = width
        if src.shape[0] > src.shape[1]:                        # portrait source: pillar bars as well
            pw = min(width, int(round(ph * 4 / 3)) & ~1)
        y0, x0 = (height - ph) // 2, (width - pw) // 2
        out[y0:y0 + ph, x0:x0 + pw] = _mirror_tile(src, pw, ph, int(g.integers(0, src.shape[1])), int(g.integers(0, src.shape[0])))
        return out
    return np.ascontiguousarray(_mirror_tile(src, width, height, int(g.integers(0, src.shape[1])), int(g.integers(0, src.shape[0]))))


def frames(width, height, bits, indices):
    return [photo_y(width, height, bits, i) for i in indices]
