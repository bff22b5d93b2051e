"""Evaluation bookkeeping used by test_pMCTF_flex.py (pMCTF/utils/video_eval_utils.py:14-155)."""
import argparse
import json
import os
import re

import numpy as np


def str2bool(v):
    if isinstance(v, bool):
        return v
    s = v.lower()
    if s in ("yes", "true", "t", "y", "1"):
        return True
    if s in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def create_folder(path, print_if_create=False):
    if not os.path.exists(path):
        os.makedirs(path)
        if print_if_create:
            print(f"created folder: {path}")


def dump_json(obj, fid, float_digits=-1, **kwargs):
    """json.dump with floats printed with `float_digits` decimals (video_eval_utils.py:51-62)."""
    if float_digits < 0:
        json.dump(obj, fid, **kwargs)
        return
    tag = "@@PMCTF_FLOAT@@"

    def conv(o):
        if isinstance(o, (float, np.floating)):
            return tag + format(float(o), f".{float_digits}f")
        if isinstance(o, np.integer):
            return int(o)
        if isinstance(o, dict):
            return {k: conv(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [conv(v) for v in o]
        return o

    text = json.dumps(conv(obj), **kwargs)
    fid.write(re.sub(r'"' + tag + r'(-?[0-9.a-z]+)"', r"\1", text))


def generate_log_json(frame_num, frame_types, bits, bpp_mv, psnrs, rgb_psnrs, ssims, frame_pixel_num, test_time,
                      gop_choice=None, ds_choice=None, tested_opts=None):
    """Per-type (0=I/L frame, 1=P/H frame, other=B) and overall averages (video_eval_utils.py:65-155)."""
    acc = {t: {"bit": 0.0, "psnr": 0.0, "rgb": 0.0, "ssim": 0.0, "mv": 0.0, "n": 0} for t in ("i", "p", "b")}
    for idx in range(frame_num):
        t = "i" if frame_types[idx] == 0 else ("p" if frame_types[idx] == 1 else "b")
        a = acc[t]
        a["bit"] += bits[idx]
        a["psnr"] += psnrs[idx]
        a["rgb"] += rgb_psnrs[idx]
        a["ssim"] += ssims[idx]
        if t != "i":
            a["mv"] += bpp_mv[idx]
        a["n"] += 1
    i, p, b = acc["i"], acc["p"], acc["b"]
    log = {
        "frame_pixel_num": frame_pixel_num,
        "i_frame_num": i["n"], "p_frame_num": p["n"], "b_frame_num": b["n"],
        "ave_i_frame_bpp": i["bit"] / i["n"] / frame_pixel_num,
        "ave_i_frame_psnr": i["psnr"] / i["n"],
        "ave_i_frame_psnr_rgb": i["rgb"] / i["n"],
        "ave_i_frame_msssim": i["ssim"] / i["n"],
        "frame_bpp": list(np.array(bits) / frame_pixel_num),
        "frame_bpp_mv": bpp_mv, "frame_psnr": psnrs, "frame_psnr_rgb": rgb_psnrs, "frame_msssim": ssims,
        "frame_type": frame_types, "test_time": test_time,
    }
    if gop_choice is not None and ds_choice is not None:
        log["gop_choice"], log["ds_choice"], log["tested_opts"] = gop_choice, ds_choice, tested_opts
    for t, a in (("p", p), ("b", b)):
        if a["n"] > 0:
            log[f"ave_{t}_frame_bpp"] = a["bit"] / (a["n"] * frame_pixel_num)
            log[f"ave_{t}_frame_psnr"] = a["psnr"] / a["n"]
            log[f"ave_{t}_frame_psnr_rgb"] = a["rgb"] / a["n"]
            log[f"ave_{t}_frame_msssim"] = a["ssim"] / a["n"]
            log[f"ave_{t}_frame_bpp_mv"] = a["mv"] / a["n"]
        elif t == "p":
            log.update({"ave_p_frame_bpp": 0, "ave_p_frame_psnr": 0, "ave_p_frame_psnr_rgb": 0,
                        "ave_p_frame_msssim": 0})
    log["ave_all_frame_bpp"] = (i["bit"] + p["bit"] + b["bit"]) / (frame_num * frame_pixel_num)
    log["ave_all_frame_bpp_mv"] = (p["mv"] + b["mv"]) / (p["n"] + b["n"])
    log["ave_all_frame_psnr"] = (i["psnr"] + p["psnr"] + b["psnr"]) / frame_num
    log["ave_all_frame_psnr_rgb"] = (i["rgb"] + p["rgb"] + b["rgb"]) / frame_num
    log["ave_all_frame_msssim"] = (i["ssim"] + p["ssim"] + b["ssim"]) / frame_num
    if tested_opts is not None:
        log["ave_tested_opts"] = sum(tested_opts) / len(tested_opts)
    return log
