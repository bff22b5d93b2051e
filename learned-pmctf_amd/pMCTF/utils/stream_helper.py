"""Bitstream file framing and small helpers used by the harness and the model
(same names/semantics as pMCTF/utils/stream_helper.py:23-56,181-220; all integers big-endian)."""
import struct
from pathlib import Path

import numpy as np
import torch
from torch.nn.modules.utils import consume_prefix_in_state_dict_if_present


def get_padding_size(height, width, p=64):
    """right/bottom padding to the next multiple of p -> (left, right, top, bottom)"""
    new_h = -(-height // p) * p
    new_w = -(-width // p) * p
    return 0, new_w - width, 0, new_h - height


def get_downsampled_shape(height, width, p):
    new_h = -(-int(height) // p) * p if float(height).is_integer() else (int(height) + p - 1) // p * p
    new_w = -(-int(width) // p) * p if float(width).is_integer() else (int(width) + p - 1) // p * p
    return int(new_h / p + 0.5), int(new_w / p + 0.5)


def get_rounded_q(q_scale):
    q_scale = float(np.clip(np.asarray(q_scale, dtype=np.float64).reshape(-1)[0], 0.01, 655.))
    q_index = int(np.round(q_scale * 100))
    return q_index / 100, q_index


def get_state_dict(ckpt_path):
    ckpt = torch.load(ckpt_path, map_location=torch.device("cpu"))
    for key in ("state_dict", "net"):
        if key in ckpt:
            ckpt = ckpt[key]
    consume_prefix_in_state_dict_if_present(ckpt, prefix="module.")
    return ckpt


def mv_header(stream_len, mv_y_q_index=0):
    return struct.pack(">H", mv_y_q_index) + struct.pack(">I", stream_len)


def image_header(height, width, num_channels, stream_len):
    return struct.pack(">III", height, width, num_channels) + struct.pack(">I", stream_len)


def encode_p(string, mv_y_q_index, output):
    with Path(output).open("wb") as f:
        f.write(mv_header(len(string), mv_y_q_index))
        f.write(string)


def decode_p(inputpath):
    with Path(inputpath).open("rb") as f:
        (mv_y_q_index,) = struct.unpack(">H", f.read(2))
        (n,) = struct.unpack(">I", f.read(4))
        string = f.read(n)
    return mv_y_q_index, string


def encode_image(height, width, num_channels, bit_stream, output):
    with Path(output).open("wb") as f:
        f.write(image_header(height, width, num_channels, len(bit_stream)))
        f.write(bit_stream)


def decode_image(inputpath):
    with Path(inputpath).open("rb") as f:
        height, width, num_channel = struct.unpack(">III", f.read(12))
        (n,) = struct.unpack(">I", f.read(4))
        bit_stream = f.read(n)
    return height, width, num_channel, bit_stream
