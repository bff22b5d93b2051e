"""Harness-side colour/IO helpers (subset of pMCTF/utils/util.py used by test_pMCTF_flex.py:16-20)."""
import sys

import numpy as np
import torch
import torch.nn.functional as F


def image_import(filename, width, height, POC=0, bitdepth=np.uint8, colorformat=420, as444=False, as420=False):
    """Read picture number POC from a planar YUV file (util.py:239-296)."""
    assert colorformat in (420, 444, 400)
    bps = 2 if bitdepth in (np.uint16, np.int16) else 1
    frame_samples = {420: width * height * 3 // 2, 444: width * height * 3, 400: width * height}[colorformat]
    try:
        with open(filename, "rb") as f:
            f.seek(frame_samples * POC * bps)
            Y = np.fromfile(f, dtype=bitdepth, count=width * height).reshape(height, width)
            if colorformat == 400:
                if as420:
                    z = np.zeros((height // 2, height // 2), dtype=bitdepth)
                    return Y, z, z.copy()
                return Y
            cw, ch = (width // 2, height // 2) if colorformat == 420 else (width, height)
            Cb = np.fromfile(f, dtype=bitdepth, count=cw * ch).reshape(ch, cw)
            Cr = np.fromfile(f, dtype=bitdepth, count=cw * ch).reshape(ch, cw)
    except Exception as exc:  # same behaviour as the reference: report and stop
        print(f"Could not open {filename}: {exc}")
        sys.exit()
    if as444:
        return np.dstack((Y, Cb, Cr)) if colorformat == 444 else None
    return Y, Cb, Cr


def ycbcr2rgb(ycbcr, bitdpeth=8):
    """JPEG YCbCr -> RGB (util.py:43-70): R=Y+1.403(Cr-d), G=Y-0.714(Cr-d)-0.344(Cb-d), B=Y+1.773(Cb-d)"""
    delta = 128 if bitdpeth == 8 else 32768
    if isinstance(ycbcr, np.ndarray):
        y, cb, cr = ycbcr[:, :, 0], ycbcr[:, :, 1], ycbcr[:, :, 2]
        rgb = np.zeros_like(ycbcr)
        rgb[:, :, 0] = y + 1.403 * (cr - delta)
        rgb[:, :, 1] = y - 0.714 * (cr - delta) - 0.344 * (cb - delta)
        rgb[:, :, 2] = y + 1.773 * (cb - delta)
        return rgb
    if ycbcr.dim() == 4:
        y, cb, cr = ycbcr[:, 0], ycbcr[:, 1], ycbcr[:, 2]
        return torch.stack((y + 1.403 * (cr - delta), y - 0.714 * (cr - delta) - 0.344 * (cb - delta),
                            y + 1.773 * (cb - delta)), dim=1)
    y, cb, cr = ycbcr[0], ycbcr[1], ycbcr[2]
    return torch.stack((y + 1.403 * (cr - delta), y - 0.714 * (cr - delta) - 0.344 * (cb - delta),
                        y + 1.773 * (cb - delta)), dim=0)


def yuv_420_to_444(yuv, mode="bilinear", return_tuple=False):
    """Upsample chroma x2 and stack (util.py:108-143)."""
    if len(yuv) != 3 or any(not isinstance(c, torch.Tensor) for c in yuv):
        raise ValueError("Expected a tuple of 3 torch tensors")
    if mode not in ("bilinear", "nearest"):
        raise ValueError(f'Invalid upsampling mode "{mode}".')
    kw = {"align_corners": False} if mode == "bilinear" else {}
    y, u, v = yuv
    u = F.interpolate(u, scale_factor=2, mode=mode, **kw)
    v = F.interpolate(v, scale_factor=2, mode=mode, **kw)
    if return_tuple:
        return y, u, v
    return torch.cat((y, u, v), dim=1)


def normalize_tensor(im, im_name="lh"):
    """Map a tensor into [-1, 1] for debug plots (pMCTF/utils/util.py:327-348; used by test_pMCTF_CA.py:97).
    Approximation subbands / images ('ll', 'x', 'x_hat') are stretched to the full range when they exceed it; detail
    subbands keep zero at zero: the larger-magnitude end goes to 1 (flipping the sign when that end is negative)."""
    lo, hi = torch.min(im), torch.max(im)
    if im_name in ("ll", "x", "x_hat"):
        out_lo, out_hi = -1.0, 1.0
    else:
        if torch.abs(hi) <= torch.abs(lo):
            im = -im
            lo, hi = torch.min(im), torch.max(im)
        out_hi = 1.0
        out_lo = torch.sign(lo) * torch.abs(lo) / torch.abs(hi)
    if hi > 1 or lo < -1:
        im = (out_hi - out_lo) * (im - lo) / (hi - lo) + out_lo
    return im
