"""Final image / normal metrics of the stage-2 evaluation: host-side numpy in float64, as in the
reference (stage2/utils/metrics.py:17-51).  Not on the device path: they run once per evaluated view on
images that have already been copied back for writing."""
import math

import numpy as np


def MAE(vec1, vec2, mask=None, normalize=True):
    """Mean angular error in degrees between two normal maps [N,3] or [H,W,3] (metrics.py:17-37).
    Returns (mean, per-pixel errors of the masked pixels).  Zero vectors stay zero (=> 90 degrees)."""
    a = np.array(vec1, dtype=np.float64, copy=True)
    b = np.array(vec2, dtype=np.float64, copy=True)
    if normalize:
        na = np.linalg.norm(a, axis=-1)
        nb = np.linalg.norm(b, axis=-1)
        a = a / (na[..., None] + 1e-5)
        b = b / (nb[..., None] + 1e-5)
        a[na == 0] = 0
        b[nb == 0] = 0
    dots = np.clip((a * b).sum(-1), -1.0, 1.0)
    if mask is not None:
        dots = dots[np.asarray(mask).astype(bool)]
    err = np.degrees(np.arccos(dots))
    return err.mean(), err


def PSNR(img1, img2, mask=None):
    """-10 log10(mean squared error) over the masked pixels of two [H,W,3] images in [0,1]; 100 when identical
    (metrics.py:39-51)."""
    a = np.asarray(img1, dtype=np.float64)
    b = np.asarray(img2, dtype=np.float64)
    if mask is not None:
        m = np.asarray(mask).astype(bool)
        a, b = a[m], b[m]
    mse = np.mean((a - b) ** 2)
    return 100 if mse == 0 else -10.0 * math.log10(mse)
