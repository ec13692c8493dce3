"""Deterministic synthetic equirectangular sky for SkyMap (lib/backgrounds.rs:51-92).

The HDRIs the reference's GUI scene loads (assets/lake.exr, studio.exr) are not in the reference tree
(`.MISSING_LARGE_BLOBS`), and EXR decoding is outside the hot path, so BASELINE configs C3-C5 use this
procedural stand-in: a vertical sky gradient, a ground tint below the horizon and a bright sun disc.
Only +, -, *, / and comparisons on float32 are used, so the texels are identical on every host.
Row 0 is the top scanline, as `exr` delivers it (backgrounds.rs:68)."""
from __future__ import annotations

import numpy as np


def synthetic_sky(width: int = 2048, height: int = 1024, sun_uv=(0.30, 0.28), sun_radius: float = 0.02,
                  sun_radiance: float = 40.0) -> np.ndarray:
    f = np.float32
    v = ((np.arange(height, dtype=np.float32) + f(0.5)) / f(height))[:, None]  # 0 = top row
    u = ((np.arange(width, dtype=np.float32) + f(0.5)) / f(width))[None, :]
    img = np.zeros((height, width, 3), dtype=np.float32)
    up = np.clip((f(0.5) - v) * f(2.0), f(0.0), f(1.0))      # 1 at the zenith row, 0 at the horizon
    down = np.clip((v - f(0.5)) * f(2.0), f(0.0), f(1.0))
    zenith = np.array([0.25, 0.45, 0.95], dtype=np.float32)
    horizon = np.array([0.85, 0.88, 0.95], dtype=np.float32)
    ground = np.array([0.28, 0.25, 0.22], dtype=np.float32)
    sky = horizon[None, None, :] * (f(1.0) - up[..., None]) + zenith[None, None, :] * up[..., None]
    gnd = horizon[None, None, :] * (f(1.0) - down[..., None]) * f(0.6) + ground[None, None, :] * down[..., None]
    img[:] = np.where((v < f(0.5))[..., None], sky, gnd)
    # mild azimuthal variation so that columns differ
    img *= (f(1.0) + f(0.08) * (f(2.0) * np.abs(u - f(0.5)) - f(0.5)))[..., None]
    du = u - f(sun_uv[0])
    dv = (v - f(sun_uv[1])) * f(0.5)  # equirect: v spans pi, u spans 2 pi
    d2 = du * du + dv * dv
    sun = d2 < f(sun_radius) * f(sun_radius)
    img[sun] = np.array([1.0, 0.92, 0.8], dtype=np.float32) * f(sun_radiance)
    return np.ascontiguousarray(img.astype(np.float32))
