#!/usr/bin/env python3
"""Which bitstream files of a full-size encode differ from the real reference's (digest fixtures under tests/golden:
SHA-1 + length of every file the reference's CPU run wrote)?  One line per pair: mv / H / Hc / L / Lc = identical bytes,
'x' = same length but different bytes (at least one symbol or CDF row differs inside), '+n'/'-n' = n bytes longer/shorter.
usage: parity_files.py GOP Q [pan|layers] [WxH]"""
import hashlib, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("learned-pmctf_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import pmctf_gop, pmctf_synth
from helpers import frames, product_model
gop, q = int(sys.argv[1]), int(sys.argv[2])
seq = sys.argv[3] if len(sys.argv) > 3 else "pan"
w, h = (int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "1920x1080").split("x"))
name = "reference_%dx%d_gop%d_me4%s%s_digest.npz" % (w, h, gop, "" if q == 3 else f"_q{q}", "" if seq == "pan" else "_" + seq)
g = np.load(os.path.join(ROOT, "tests", "golden", name))
net, _ = product_model(4)
net.engine().keep_streams = True
fr = frames(w, h, gop, device="cuda") if seq == "pan" else \
    [list(pmctf_synth.frames_to_tensors(f, device="cuda")) for f in pmctf_synth.synth_yuv420_layers(w, h, gop)]
with tempfile.TemporaryDirectory() as td, torch.no_grad():
    enc = pmctf_gop.encode_gop(net, fr, h, w, q, td)
print(f"{w}x{h} GOP {gop} q {q} ({seq}); bit deltas {(np.array(enc['bits']) - g['gop.bits']).astype(int).tolist()}")
tot = {}
for i, r in enumerate(enc["results"]):
    ref_i, cur = int(g[f"gop.pair{i}.meta"][1]), int(g[f"gop.pair{i}.meta"][2])
    cells = []
    for kind, fkey in (("mv", f"{cur}_mv.bin"), ("H", f"{cur}.bin"), ("Hc", f"{cur}_C_main.bin"), ("L", "0_main.bin"), ("Lc", "0_C_main.bin")):
        k = f"gop.pair{i}.filesha1.{fkey}"
        if kind not in r["files"] or k not in g.files:
            continue
        data = r["files"][kind]
        n_ref = int(g[k.replace("filesha1", "filelen")])
        if hashlib.sha1(data).digest() == g[k].tobytes():
            c = "="
        elif len(data) == n_ref:
            c = "x"
        else:
            c = f"{len(data) - n_ref:+d}"
        cells.append(f"{kind}:{c}")
        t = tot.setdefault(kind, [0, 0]); t[0] += c == "="; t[1] += 1
    print(f"pair {i:2d} (frames {ref_i:2d},{cur:2d}): " + "  ".join(cells))
print("identical / total per kind:", {k: f"{a}/{b}" for k, (a, b) in tot.items()})
