import os, sys, tempfile, hashlib, json
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "learned-pmctf_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import product_model, frames
import pmctf_gop
g = np.load(os.path.join(ROOT, "tests/golden/reference_1920x1080_gop8_me4_ds2_digest.npz"))
net, _ = product_model(4)
net.engine().keep_streams = True
w, h, gop = 1920, 1080, 8
fr = frames(w, h, gop, device="cuda")
with tempfile.TemporaryDirectory() as td, torch.no_grad():
    enc = pmctf_gop.encode_gop(net, fr, h, w, 3, td, me_downsample=2)
    rec = pmctf_gop.decode_gop(net, enc["frames_coded"])
    ps = pmctf_gop.gop_psnr(rec, fr, h, w)
same = diff = 0
for i, r in enumerate(enc["results"]):
    cur = int(g[f"gop.pair{i}.meta"][2])
    for name, fkey in (("mv", f"{cur}_mv.bin"), ("H", f"{cur}.bin"), ("Hc", f"{cur}_C_main.bin"), ("L", "0_main.bin"), ("Lc", "0_C_main.bin")):
        k = f"gop.pair{i}.filesha1.{fkey}"
        if name in r["files"] and k in g.files:
            if hashlib.sha1(r["files"][name]).digest() == g[k].tobytes(): same += 1
            else: diff += 1; print("differs:", i, name, len(r["files"][name]), int(g[k.replace("filesha1","filelen")]))
print("dbits", (np.array(enc["bits"]) - g["gop.bits"]).tolist(), "dbits_mv", (np.array(enc["bits_mv"]) - g["gop.bits_mv"]).tolist())
print("psnr_err", float(np.abs(np.array([p["yuv"] for p in ps]) - g["gop.psnr_yuv"]).max()), "same", same, "diff", diff)
