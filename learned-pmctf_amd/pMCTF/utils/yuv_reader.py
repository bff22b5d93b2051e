"""Planar 8-bit 4:2:0 reader (pMCTF/utils/yuv_reader.py:11-40): one frame = W*H*3/2 bytes, Y then Cb then Cr."""
import os

import numpy as np

from pMCTF.utils.util import image_import


class YUVReader:
    def __init__(self, src_file, width, height, start_index=0):
        assert os.path.exists(src_file)
        self.src_file = src_file
        self.width = width
        self.height = height
        self.current_frame_index = start_index
        self.eof = False

    def read_one_frame(self, src_format="rgb"):
        if self.eof:
            return None if src_format == "rgb" else (None, None, None)
        Y, Cb, Cr = image_import(self.src_file, self.width, self.height, POC=self.current_frame_index,
                                 bitdepth=np.uint8, colorformat=420)
        assert Y.shape == (self.height, self.width)
        self.current_frame_index += 1
        return Y, Cb, Cr

    def close(self):
        self.current_frame_index = 0
