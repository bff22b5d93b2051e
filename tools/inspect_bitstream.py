#!/usr/bin/env python3
"""Print what is inside pMCTF bitstream files (the on-disk format of stream_helper.py:181-207 + py_rans.cpp:99-118).

  image file  (k.bin, k_C_main.bin, 0_main.bin, 0_C_main.bin):  >III height width planes, >I n, n stream bytes
  motion file (k_mv.bin):                                        >H mv_y_q_index, >I n, n stream bytes
  stream:  flag byte = ((parts-1) << 4) | (1 if the part sizes are 16-bit else 0), then parts-1 sizes, then each part
           = little-endian 32-bit rANS words: the final 64-bit state first (low word, high word), payload after it.

Usage: tools/inspect_bitstream.py FILE [FILE ...]      (needs nothing but the standard library)
"""
import os
import struct
import sys


def describe_stream(b, indent="    "):
    if not b:
        print(indent + "empty stream")
        return
    flag = b[0]
    parts = (flag >> 4) + 1
    two = flag & 1
    hdr = 1 + (parts - 1) * (2 if two else 4)
    sizes = []
    for i in range(parts - 1):
        o = 1 + i * (2 if two else 4)
        sizes.append(struct.unpack("<H" if two else "<I", b[o:o + (2 if two else 4)])[0])
    sizes.append(len(b) - hdr - sum(sizes))
    print(f"{indent}flag 0x{flag:02x}: {parts} part(s), {'16' if two else '32'}-bit size fields, header {hdr} B")
    off = hdr
    for i, sz in enumerate(sizes):
        part = b[off:off + sz]
        off += sz
        note = ""
        if sz >= 8 and sz % 4 == 0:
            lo, hi = struct.unpack("<II", part[:8])
            note = f", final rANS state 0x{(hi << 32) | lo:016x}, {sz // 4 - 2} payload word(s)"
        elif sz % 4:
            note = "  (!) not a whole number of 32-bit words"
        print(f"{indent}part {i}: {sz} B{note}")
    if off != len(b):
        print(f"{indent}(!) {len(b) - off} trailing byte(s)")


def describe(path):
    data = open(path, "rb").read()
    name = os.path.basename(path)
    print(f"{path}: {len(data)} B = {8 * len(data)} bits")
    if name.endswith("_mv.bin"):
        if len(data) < 6:
            print("  (!) too short for a motion header")
            return
        (q,) = struct.unpack(">H", data[:2])
        (n,) = struct.unpack(">I", data[2:6])
        body = data[6:6 + n]
        print(f"  motion file: mv_y_q_index {q}, stream {n} B" + ("" if len(body) == n else "  (!) truncated"))
        describe_stream(body)
    else:
        if len(data) < 16:
            print("  (!) too short for an image header")
            return
        h, w, c = struct.unpack(">III", data[:12])
        (n,) = struct.unpack(">I", data[12:16])
        kind = {1: "Y", 2: "UV", 3: "RGB"}.get(c, f"{c} planes")
        body = data[16:16 + n]
        print(f"  image file: {w}x{h} ({kind}), stream {n} B" + ("" if len(body) == n else "  (!) truncated"))
        print(f"  rate: {8 * len(data) / (h * w):.4f} bit per pixel of this plane size")
        describe_stream(body)


if __name__ == "__main__":
    if len(sys.argv) < 2:
        print(__doc__)
        sys.exit(2)
    for p in sys.argv[1:]:
        describe(p)
