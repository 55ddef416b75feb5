"""Deterministic synthetic yuv420p test/bench frames.

Y: band-limited texture + hard edges + +-4 LSB noise clipped to video range; plus the three
adversarial frames (constant, uniform random full range, 1-px checkerboard).  Pure numpy, no
dependency on the oracle or on the HIP library.
"""
import numpy as np


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def natural_y(width, height, bits=8, seed=12345):
    """Texture + edges + noise, clipped to video range [16,235] (8-bit) / [64,940] (10-bit)."""
    g = _rng(seed)
    maxv = (1 << bits) - 1
    lo, hi = (16, 235) if bits == 8 else (64, 940)
    y, x = np.mgrid[0:height, 0:width].astype(np.float64)
    img = np.zeros((height, width))
    for _ in range(6):                                   # band-limited texture: a few oriented sinusoids
        fx, fy = g.uniform(0.01, 0.35, 2) * g.choice([-1, 1], 2)
        img += g.uniform(0.05, 0.2) * np.sin(2 * np.pi * (fx * x + fy * y) + g.uniform(0, 6.28))
    img = 0.5 + 0.35 * img / max(1e-9, np.abs(img).max())
    for _ in range(8):                                   # hard-edged rectangles and a diagonal half-plane
        x0, x1 = sorted(g.integers(0, width, 2)); y0, y1 = sorted(g.integers(0, height, 2))
        img[y0:y1 + 1, x0:x1 + 1] += g.uniform(-0.3, 0.3)
    a, b = g.uniform(-1, 1, 2)
    img[(a * (x - width / 2) + b * (y - height / 2)) > 0] += 0.1
    img = img * (hi - lo) + lo + g.integers(-4, 5, img.shape) * (maxv / 255.0)
    out = np.clip(np.rint(img), lo, hi)
    return out.astype(np.uint8 if bits == 8 else np.uint16)


def constant_y(width, height, bits=8, value=None):
    v = (128 if bits == 8 else 512) if value is None else value
    return np.full((height, width), v, dtype=np.uint8 if bits == 8 else np.uint16)


def random_y(width, height, bits=8, seed=777):
    g = _rng(seed)
    return g.integers(0, 1 << bits, (height, width)).astype(np.uint8 if bits == 8 else np.uint16)


def checker_y(width, height, bits=8):
    lo, hi = 0, (1 << bits) - 1
    y, x = np.mgrid[0:height, 0:width]
    return np.where((x + y) & 1, hi, lo).astype(np.uint8 if bits == 8 else np.uint16)


def chroma(width, height, bits=8):
    return np.full((height, width), 128 if bits == 8 else 512, dtype=np.uint8 if bits == 8 else np.uint16)


FRAME_KINDS = {"natural": natural_y, "constant": constant_y, "random": random_y, "checker": checker_y}
