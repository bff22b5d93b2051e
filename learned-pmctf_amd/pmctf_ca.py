"""Content-adaptive GOP driver: for every group of `gop_size` frames choose the temporal decomposition depth (GOP size
in {gop_size, gop_size/2, ..., 4}) and the resolution the motion is estimated and coded at (me_downsample in
{1, 2, 4, 8}) by rate-distortion cost, as the reference's second harness does (test_pMCTF_CA.py: `code_one_gop`
:103-300, the search in `run_test` :341-414).  An own driver over the model API (`encode_one_stage`, `inverse_MCTF`),
usable with the HIP product and with the oracle; it multiplies the encode work by the number of options tried, which is
what the fast encode path is for.

Two modes, as in the reference (`--write_stream`):
  * write mode      — every trial writes real bitstreams (bits = file sizes);
  * estimate mode   — `encode_one_stage(output_path=None)`: Laplace / factorized bit estimates, no range coding.  (In the
                      reference this branch raises KeyError; the product implements what it was meant to do.)
"""
import math
import os

import numpy as np
import torch

import pmctf_gop

LAMDA_LIST = (1, 27)            # test_pMCTF_CA.py:27


def get_cur_lamda(q_index, qp_num=21):
    """test_pMCTF_CA.py:29-34"""
    step = (math.log(LAMDA_LIST[1]) - math.log(LAMDA_LIST[0])) / (qp_num - 1)
    return math.exp(math.log(LAMDA_LIST[0]) + step * q_index) * 0.003


def get_mse(psnrs, max_val=255):
    """test_pMCTF_CA.py:36-38"""
    return list(max_val ** 2 / (10 ** (np.array(psnrs) / 10)))


def pad_frames(frames, psize):
    """zero padding right/bottom to multiples of psize (chroma psize/2), test_pMCTF_CA.py:117-131"""
    out = []
    for y, c in frames:
        h, w = y.shape[2], y.shape[3]
        ph, pw = -(-h // psize) * psize, -(-w // psize) * psize
        yp = torch.nn.functional.pad(y, (0, pw - w, 0, ph - h))
        cp = torch.nn.functional.pad(c, (0, (pw - w) // 2, 0, (ph - h) // 2))
        out.append([yp, cp])
    return out


def code_one_gop(codec, frames_orig, pic_height, pic_width, q_index, me_downsample=1, bin_folder=None,
                 write_stream=True, skip_decoding=True):
    """One closed GOP with motion at 1/me_downsample resolution (test_pMCTF_CA.py:103-300): analysis, temporal
    synthesis, per-frame bits and YUV-PSNR.  frames_orig: un-padded [Y (1,1,h,w), UV (2,1,h/2,w/2)] device tensors."""
    psize = pmctf_gop.ca_psize(me_downsample)
    frames = pad_frames(frames_orig, psize)
    enc = pmctf_gop.encode_gop(codec, frames, pic_height, pic_width, q_index, bin_folder if write_stream else None,
                               skip_decoding=skip_decoding, psize=psize, me_downsample=me_downsample)
    # the CA harness reconstructs luma with the FIRST stage's lifting filters at every stage (inverse_MCTF called
    # without stage_idx, test_pMCTF_CA.py:239) and chroma with the stage's own (:240)
    rec = pmctf_gop.decode_gop(codec, enc["frames_coded"], luma_stage0=True)
    ps = pmctf_gop.gop_psnr(rec, frames, pic_height, pic_width)
    px = pic_height * pic_width
    gop = len(frames_orig)
    return {"bits": enc["bits"], "bpps": [b / px for b in enc["bits"]], "bpp_mv": [b / px for b in enc["bits_mv"]],
            "psnrs": [p["yuv"] for p in ps], "psnr_y": [p["y"] for p in ps], "psnr_cb": [p["cb"] for p in ps],
            "psnr_cr": [p["cr"] for p in ps], "frame_types": [0] + [1] * (gop - 1)}


def _merge(a, b):
    for k, v in b.items():
        a[k] = a[k] + v
    return a


def search_gop(codec, frames_orig, pic_height, pic_width, q_index, bin_folder=None, write_stream=True,
               skip_decoding=True, ds_factors=(1, 2, 4, 8), min_gop=4, on_trial=None):
    """The RD search of test_pMCTF_CA.py:341-414 for one group of len(frames_orig) frames.
    Pass 1 (full-resolution motion): GOP sizes G, G/2, ... are tried until the cost rises; the size before the rise
    wins (the smallest size if it never rises).  Then, for that size only, the motion resolution is halved again and
    again until the cost rises.  cost = sum(bpp) + lambda(q_index) * sum(mse(YUV-PSNR)) over the G frames; a size below
    G codes the group as G/size closed GOPs.  Returns the chosen logs plus the record of what was tried."""
    G = len(frames_orig)
    sizes = [G]
    while sizes[-1] // 2 >= min_gop:
        sizes.append(sizes[-1] // 2)
    lamda = get_cur_lamda(q_index, codec.get_qp_num() if hasattr(codec, "get_qp_num") else 21)
    results = {}                # (gop size, ds) -> logs with "rd"
    trials = []

    def trial(size, ds):
        logs = None
        for start in range(0, G, size):
            res = code_one_gop(codec, frames_orig[start:start + size], pic_height, pic_width, q_index, ds, bin_folder,
                               write_stream, skip_decoding)
            logs = res if logs is None else _merge(logs, res)
        logs["rd"] = sum(logs["bpps"]) + lamda * sum(get_mse(logs["psnrs"]))
        results[(size, ds)] = logs
        trials.append((size, ds, logs["rd"]))
        if on_trial is not None:
            on_trial(size, ds, logs)
        return logs["rd"]

    # pass 1: GOP size at the first motion resolution
    ds0 = ds_factors[0]
    best = len(sizes) - 1
    prev = None
    for i, size in enumerate(sizes):
        rd = trial(size, ds0)
        if prev is not None and prev < rd:
            best = i - 1
            break
        prev = rd
    best_size = sizes[best]
    # pass 2: motion resolution for that size
    best_ds = ds_factors[-1]
    for j in range(1, len(ds_factors)):
        rd = trial(best_size, ds_factors[j])
        if results[(best_size, ds_factors[j - 1])]["rd"] < rd:
            best_ds = ds_factors[j - 1]
            break
    return {"gop_choice": best_size, "ds_choice": best_ds, "tested_opts": len(trials), "trials": trials,
            "logs": results[(best_size, best_ds)]}


def run_sequence(codec, frames_orig, pic_height, pic_width, q_index, gop_size, bin_folder=None, write_stream=True,
                 skip_decoding=True, ds_factors=(1, 2, 4, 8)):
    """All groups of a sequence (test_pMCTF_CA.py:303-447): per-frame logs of the chosen options + the choices."""
    assert len(frames_orig) % gop_size == 0
    keys = ("frame_types", "psnrs", "psnr_y", "psnr_cb", "psnr_cr", "bits", "bpps", "bpp_mv")
    out = {k: [] for k in keys}
    out.update({"gop_choice": [], "ds_choice": [], "tested_opts": [], "rd": 0.0})
    for g in range(0, len(frames_orig), gop_size):
        r = search_gop(codec, frames_orig[g:g + gop_size], pic_height, pic_width, q_index, bin_folder, write_stream,
                       skip_decoding, ds_factors)
        for k in keys:
            out[k] += r["logs"][k]
        out["rd"] += r["logs"]["rd"]
        for k in ("gop_choice", "ds_choice", "tested_opts"):
            out[k].append(r[k])
    return out
