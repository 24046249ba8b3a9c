"""Host-side helpers for rendered images: run-length compression of the JSON form, nearest-neighbour
upscaling and a median filter (the ``Post`` class of the reference, T3DR.py:1550-1647).  They work on the
finished 8-bit image; written with numpy instead of per-pixel loops.  ``upscale`` and ``denoise`` also run on the
GPU when handed a ``context`` (a ``raytracing_amd._native.Context``): same bytes, from the library's kernels."""
from __future__ import annotations

import copy

import numpy as np


class Post:
    @staticmethod
    def compressRenderedDict(rendered, tolerance=10):
        """Run-length code every row of ``rendered['imgData']`` as ``[[pixel, count], ...]``.  A run is
        extended while the pixel is within ``tolerance`` (Euclidean, RGB) of the run's FIRST pixel
        (T3DR.py:1552-1581)."""
        rows = []
        for row in rendered["imgData"]:
            px = np.asarray(row, dtype=np.float64).reshape(-1, 3)
            out, start = [], 0
            n = len(px)
            while start < n:
                far = np.nonzero(np.sqrt(((px[start + 1:] - px[start]) ** 2).sum(axis=1)) > tolerance)[0]
                end = start + 1 + (int(far[0]) if far.size else n - start - 1)
                out.append([row[start], end - start])
                start = end
            rows.append(out)
        result = copy.deepcopy({k: v for k, v in rendered.items() if k != "imgData"})
        result["imgData"] = rows
        return result

    @staticmethod
    def decompressRenderedDict(rendered):
        """Inverse of ``compressRenderedDict`` up to the tolerance (T3DR.py:1584-1598)."""
        rows = [[pixel for pixel, count in row for _ in range(count)] for row in rendered["imgData"]]
        result = copy.deepcopy({k: v for k, v in rendered.items() if k != "imgData"})
        result["imgData"] = rows
        return result

    @staticmethod
    def upscale(imageData, factor, context=None):
        """Every pixel becomes a factor x factor block (T3DR.py:1601-1613)."""
        img = np.asarray(imageData)
        if context is not None:
            return context.post_upscale(img, factor)
        return np.repeat(np.repeat(img, factor, axis=0), factor, axis=1).astype(np.uint8)

    @staticmethod
    def denoise(imageData, kernelSize, context=None):
        """Per-channel median over a kernelSize x kernelSize window; the border of width kernelSize // 2 stays
        (0, 0, 0); rows of tuples, as the reference returns them (T3DR.py:1616-1647).  The median of an even
        count is the upper middle element, like ``sorted(v)[len(v) // 2]``."""
        img = np.asarray(imageData)
        h, w = img.shape[:2]
        pad = kernelSize // 2
        out = np.zeros((h, w, 3), dtype=img.dtype)
        if context is not None:
            out = context.post_denoise(img, kernelSize)
        elif h > 2 * pad and w > 2 * pad:
            k = 2 * pad + 1
            win = np.lib.stride_tricks.sliding_window_view(img, (k, k), axis=(0, 1))   # [h-2p, w-2p, 3, k, k]
            flat = np.sort(win.reshape(h - 2 * pad, w - 2 * pad, 3, k * k), axis=-1)
            out[pad:h - pad, pad:w - pad] = flat[..., (k * k) // 2]
        return [[tuple(int(c) for c in px) for px in row] for row in out]
