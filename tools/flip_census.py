#!/usr/bin/env python3
"""Flip census: how often does the product's arithmetic (PM-F32, what the HIP kernels compute bit for bit) take a
different entropy-coding decision than the reference's (ATen CPU ops)?

For every rate point of the RD sweep the same GOP is coded twice with the oracle — back-end "torch" (ATen: reproduces the
real reference's fixtures bit for bit, tests/test_oracle_vs_golden.py) and back-end "cdef" (PM-F32) — and the symbol
traces handed to the range coder are compared stream by stream, in coding order:
  * symbols flipped per million, CDF rows flipped per million,
  * the first differing stream (pair, kind) and the position of its first flip,
  * files whose bytes differ, and the per-frame bit-count differences.
Where the fixture holds the REAL reference's symbols (128x128, q_index 3) the ATen trace is checked against them first.
CPU only (test infrastructure); no GPU, no reference import.

  python tools/flip_census.py [--size 128x128 448x256] [--q 0 4 8 12 16 20 3] [--gop 4]
"""
import argparse
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("learned-pmctf_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))

import torch  # noqa: E402

import pmctf_gop  # noqa: E402
from helpers import frames, golden, synth_sd_cpu  # noqa: E402
from pmctf_oracle.model import Oracle  # noqa: E402

KINDS = ("mv", "H", "Hc", "L", "Lc")


def census(width, height, gop, q_index, sd, check_fixture):
    fr = frames(width, height, gop)
    enc = {}
    for be in ("torch", "cdef"):
        orc = Oracle(sd, 1, be)
        with tempfile.TemporaryDirectory() as td, torch.no_grad():
            enc[be] = pmctf_gop.encode_gop(orc, fr, height, width, q_index, td)
    note = ""
    if check_fixture:
        g = golden()
        # the reference pushes a pair's streams in the order mv, luma H, luma L, chroma H, chroma L
        ok = all(np.array_equal(np.concatenate([t[0] for k in ("mv", "H", "L", "Hc", "Lc")
                                                if r["traces"].get(k) is not None for t in r["traces"][k]]),
                                g[f"gop.pair{i}.symbols"])
                 for i, r in enumerate(enc["torch"]["results"]))
        note = "ATen trace == real reference's symbols: %s" % ok
    n_sym = n_flip = n_row = 0
    first = None
    files_diff = 0
    for i, (a, b) in enumerate(zip(enc["torch"]["results"], enc["cdef"]["results"])):
        for k in KINDS:
            ta, tb = a["traces"].get(k), b["traces"].get(k)
            if ta is None:
                continue
            sa, ia = (np.concatenate([t[j] for t in ta]) for j in (0, 1))     # per stream: list of (symbols, rows) pushes
            sb, ib = (np.concatenate([t[j] for t in tb]) for j in (0, 1))
            ds, di = sa != sb, ia != ib
            n_sym += sa.size
            n_flip += int(ds.sum())
            n_row += int(di.sum())
            if first is None and (ds.any() or di.any()):
                pos = int(np.argmax(ds | di))
                first = f"pair {i} {k} @ {pos}/{sa.size}"
            files_diff += a["files"][k] != b["files"][k]
    dbits = (np.array(enc["cdef"]["bits"]) - np.array(enc["torch"]["bits"])).astype(int).tolist()
    return {"symbols": n_sym, "sym_ppm": 1e6 * n_flip / n_sym, "row_ppm": 1e6 * n_row / n_sym, "first": first or "-",
            "files_differ": files_diff, "dbits": dbits, "note": note}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", nargs="+", default=["128x128", "448x256"])
    ap.add_argument("--q", nargs="+", type=int, default=[0, 3, 4, 8, 12, 16, 20])
    ap.add_argument("--gop", type=int, default=4)
    args = ap.parse_args()
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    sd = synth_sd_cpu(1)
    print("| size | q_index | symbols | symbols flipped / M | CDF rows flipped / M | first differing stream | files with "
          "different bytes | per-frame bit deltas (PM-F32 - ATen) |")
    print("|---|---|---|---|---|---|---|---|")
    for size in args.size:
        w, h = (int(v) for v in size.split("x"))
        for q in args.q:
            c = census(w, h, args.gop, q, sd, check_fixture=(size == "128x128" and q == 3 and args.gop == 4))
            print(f"| {size} | {q} | {c['symbols']} | {c['sym_ppm']:.1f} | {c['row_ppm']:.1f} | {c['first']} | "
                  f"{c['files_differ']} | {c['dbits']} | {c['note']}", flush=True)


if __name__ == "__main__":
    main()
