#!/usr/bin/env python3
"""Stage-by-stage comparison of the oracle's two back-ends at FULL size: every stage of the signal path of one frame pair
(warp, temporal predict / update, each level of the lifting DWT, motion estimation, motion codec) is evaluated with ATen
(what the reference computes) and with PM-F32 (what the HIP kernels compute, bit for bit) on the SAME inputs — the
ATen result of the stage before — so a stage whose last bits differ shows up on its own, with the fraction of elements
that differ.  CPU only (test infrastructure).   usage: stage_parity.py [WxH] [--mv] [--spynet]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("learned-pmctf_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from helpers import frames, synth_sd_cpu
from pmctf_oracle.model import Oracle, get_curr_q

size = next((a for a in sys.argv[1:] if "x" in a), "1920x1080")
W, H = (int(v) for v in size.split("x"))
torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
sd = synth_sd_cpu(1)
A, P = Oracle(sd, 1, "torch"), Oracle(sd, 1, "cdef")
fr = frames(W, H, 2)
(ry, rc), (cy, cc) = fr


def report(name, a, b):
    a, b = a.contiguous().numpy(), b.contiguous().numpy()
    d = a.view(np.int32) != b.view(np.int32)
    print(f"{name:46s} {tuple(a.shape)!s:22s} differing elements {int(d.sum()):9d} / {a.size} "
          f"({100.0 * d.mean():.4f} %)  max |diff| {np.abs(a.astype(np.float64) - b).max():.3e}", flush=True)


with torch.no_grad():
    t0 = time.time()
    if "--spynet" in sys.argv:
        x, r = cy.tile((1, 3, 1, 1)) / 255, ry.tile((1, 3, 1, 1)) / 255
        ea, ep = A.spynet(x, r), P.spynet(x, r)
        report("SpyNet (six levels, chained)", ea, ep)
    mv = torch.stack([torch.full((H_ := ry.size(2), ry.size(3)), 2.3), torch.full((H_, ry.size(3)), -1.1)])[None]
    mv = mv + 0.37 * torch.sin(torch.arange(ry.size(3)) * 0.013)[None, None, None, :]
    for name, ref, cur, m in (("luma", ry, cy, mv), ("chroma", rc, cc, None)):
        if m is None:
            m = (A.K.bilinear_down2(mv) / 2).tile((2, 1, 1, 1))
        wa, wp = A.K.flow_warp(ref, m), P.K.flow_warp(ref, m)
        report(f"{name}: flow_warp", wa, wp)
        pa, pp = A.predict_filter(0, wa), P.predict_filter(0, wa)
        report(f"{name}: temporal predict filter", pa, pp)
        Hn = cur - pa
        ua, up = A.update_filter(0, A.K.flow_warp(Hn, -m)), P.update_filter(0, A.K.flow_warp(Hn, -m))
        report(f"{name}: temporal update filter", ua, up)
        ll = Hn
        for lvl in range(4):
            sa, sp_ = A.forward_lift_2d("hp_coder", ll), P.forward_lift_2d("hp_coder", ll)
            for k in ("ll", "lh", "hl", "hh"):
                report(f"{name}: DWT level {lvl} {k}", sa[k], sp_[k])
            ll = sa["ll"]
        # inverse of the last level with ATen's subbands on both sides
        ia, ip = A.backward_lift_2d("hp_coder", sa), P.backward_lift_2d("hp_coder", sa)
        report(f"{name}: inverse DWT of level 3", ia, ip)
    if "--mv" in sys.argv:
        est = mv + 0.01 * torch.randn(mv.shape, generator=torch.Generator().manual_seed(3))
        dpb = {"mv_feature": None, "ref_mv_y": None}
        q_enc, q_dec = A.get_mv_y_q(3, 0)
        ya, yp = A.mv_enc(0, est, None, q_enc), P.mv_enc(0, est, None, q_enc)
        report("MV encoder (first pair of a stage)", ya, yp)
        za, zp = A.mv_hyper_enc(0, ya), P.mv_hyper_enc(0, ya)
        report("MV hyper encoder", za, zp)
        zh = torch.round(za)
        pa_, pp_ = A.mv_prior_param_decoder(zh, dpb, 0), P.mv_prior_param_decoder(zh, dpb, 0)
        report("MV hyper decoder + prior fusion", pa_, pp_)
        qa, sa_, yha = A.compress_four_part_prior(0, ya, pa_)
        qp, sp2, yhp = P.compress_four_part_prior(0, ya, pa_)
        report("MV four-part prior: y_hat", yha, yhp)
        ma, fa = A.mv_dec(0, yha, q_dec)
        mp, fp = P.mv_dec(0, yha, q_dec)
        report("MV decoder: mv_hat", ma, mp)
        report("MV decoder: mv_feature", fa, fp)
        dpb2 = {"mv_feature": fa, "ref_mv_y": yha}
        ya2, yp2 = A.mv_enc(0, est, fa, q_enc), P.mv_enc(0, est, fa, q_enc)
        report("MV encoder (later pair: with context)", ya2, yp2)
        pa2, pp2 = A.mv_prior_param_decoder(zh, dpb2, 0), P.mv_prior_param_decoder(zh, dpb2, 0)
        report("MV prior fusion (later pair)", pa2, pp2)
    print(f"# {time.time() - t0:.0f} s")
