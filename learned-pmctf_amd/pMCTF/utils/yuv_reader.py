"""Sequential reader of planar 8-bit 4:2:0 video (the harness's input; interface of pMCTF/utils/yuv_reader.py:11-40).

A picture is W*H luma bytes followed by two (W/2)*(H/2) chroma planes.  The file stays open between pictures; reading
past the end raises the same way a short read does in the reference (assertion on the luma plane)."""
import os

import numpy as np


class YUVReader:
    def __init__(self, src_file, width, height, start_index=0):
        if not os.path.exists(src_file):
            raise AssertionError(f"no such sequence: {src_file}")
        self.src_file = src_file
        self.width, self.height = int(width), int(height)
        self.current_frame_index = int(start_index)
        self.eof = False
        self._luma = self.width * self.height
        self._chroma = (self.width // 2) * (self.height // 2)
        self._fh = None

    def _plane(self, count, rows, cols):
        data = np.fromfile(self._fh, dtype=np.uint8, count=count)
        assert data.size == count, "sequence ends inside a picture"
        return data.reshape(rows, cols)

    def read_one_frame(self, src_format="rgb"):
        """-> (Y, Cb, Cr) uint8 arrays of the next picture"""
        if self.eof:
            return None if src_format == "rgb" else (None, None, None)
        if self._fh is None:
            self._fh = open(self.src_file, "rb")
        self._fh.seek((self._luma + 2 * self._chroma) * self.current_frame_index)
        y = self._plane(self._luma, self.height, self.width)
        cb = self._plane(self._chroma, self.height // 2, self.width // 2)
        cr = self._plane(self._chroma, self.height // 2, self.width // 2)
        self.current_frame_index += 1
        return y, cb, cr

    def close(self):
        if self._fh is not None:
            self._fh.close()
            self._fh = None
        self.current_frame_index = 0
