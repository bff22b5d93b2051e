"""HIP product (through the drop-in pMCTF API) vs the oracle's PM-F32 restatement: bit-exact tensors, identical
symbol streams and identical bitstream bytes; and vs the fixtures generated from the real reference."""
import os
import tempfile

import numpy as np
import pytest
import torch

from helpers import assert_same, frames, golden, product_model

pytestmark = pytest.mark.gpu

W = H = 128


@pytest.fixture(scope="module")
def setup(cuda):
    from pmctf_oracle.model import Oracle
    net, sd = product_model(1)
    net.engine().keep_streams = True
    orc = Oracle(sd, 1, "cdef")
    return net, orc


def test_loaded_native_libraries(setup):
    import ctypes, os
    from pMCTF.hip import lib
    maps = open("/proc/self/maps").read()
    assert "libpmctf_hip.so" in maps and "libpmctf_rans.so" in maps


def test_mctf_bitexact(setup):
    import pmctf_synth
    net, orc = setup
    fr = frames(W, H, 2)
    (Y0, C0), (Y1, C1) = fr
    flow = torch.from_numpy(pmctf_synth.hashed_normal("golden.flow", (1, 2, H, W), 3.0))
    L, Ht, pred, inv = net.forward_MCTF(Y0.cuda(), Y1.cuda(), flow.cuda())
    oL, oH, opred, oinv = orc.forward_MCTF(Y0, Y1, flow)
    assert_same(pred, opred, "pred"); assert_same(Ht, oH, "H_t"); assert_same(inv, oinv, "inv"); assert_same(L, oL, "L_t")
    r, c = net.inverse_MCTF(L, Ht, flow.cuda())
    orr, oc = orc.inverse_MCTF(oL, oH, flow)
    assert_same(r, orr, "inverse ref"); assert_same(c, oc, "inverse cur")
    # chroma: batch of two planes, motion = down2(mv)/2
    rc, cc = net.inverse_MCTF(C0.cuda(), C1.cuda(), flow.cuda(), downscale=True)
    orc_, occ = orc.inverse_MCTF(C0, C1, flow, downscale=True)
    assert_same(rc, orc_, "chroma inverse ref"); assert_same(cc, occ, "chroma inverse cur")
    g = golden()
    assert np.abs(Ht.cpu().numpy() - g["unit.mctf.H"]).max() < 1e-3      # vs the real reference (fp noise only)


def test_spynet_bitexact(setup):
    net, orc = setup
    (Y0, _), (Y1, _) = frames(W, H, 2)
    est = net.engine().spynet(Y1.cuda(), Y0.cuda())
    oest = orc.spynet(Y1.tile((1, 3, 1, 1)) / 255, Y0.tile((1, 3, 1, 1)) / 255)
    assert_same(est, oest, "spynet flow")
    assert np.abs(est.cpu().numpy() - golden()["unit.spynet"]).max() < 1e-5


def test_dwt_postprocess_bitexact(setup):
    net, orc = setup
    eng = net.engine()
    g = golden()
    Hg = torch.from_numpy(g["unit.mctf.H"])
    sb = eng.forward_lift_2d("hp_coder", Hg.cuda())
    osb = orc.forward_lift_2d("hp_coder", Hg)
    for k in ("ll", "lh", "hl", "hh"):
        assert_same(sb[k], osb[k].contiguous(), f"dwt {k}")
    rec = eng.backward_lift_2d("hp_coder", sb)
    assert_same(rec, orc.backward_lift_2d("hp_coder", osb), "idwt")
    pp = eng.post_process("hp_coder", Hg.cuda(), 256.0, 1.0)
    assert_same(pp, orc.post_process("hp_coder", Hg / 256.0), "postprocess")


def test_pwave_compress_stream_identical(setup):
    from pmctf_oracle.model import get_curr_q
    net, orc = setup
    eng = net.engine()
    g = golden()
    Hg = torch.from_numpy(g["unit.mctf.H"])
    qp = get_curr_q(orc.sd["hp_q_scale.0"], 3)
    x_hat, stream = eng.pwave_compress("hp_coder", Hg.cuda(), 3, qp)
    ox, odata, otrace = orc.pwave_compress("hp_coder", Hg, [1, 1, H, W], 3, qp)
    from pMCTF.utils.stream_helper import image_header
    size, data, (sym, idx) = eng.coder.submit(stream, eng.tables, lambda n: image_header(H, W, 1, n), None, True).result()
    osym = np.concatenate([t[0] for t in otrace]); oidx = np.concatenate([t[1] for t in otrace])
    assert_same(sym, osym, "symbols"); assert_same(idx, oidx, "cdf rows")
    assert data == odata, "bitstream bytes differ"
    assert_same(x_hat, ox, "x_hat")
    # vs the real reference: identical file at this size, reconstruction within fp noise
    assert data == g["pwave.file"].tobytes()
    assert np.abs(x_hat.cpu().numpy() - g["pwave.x_hat"]).max() < 2e-3


def test_gop4_files_bits_psnr(setup):
    import pmctf_gop
    net, orc = setup
    fr = frames(W, H, 4)
    frd = [[y.cuda(), c.cuda()] for y, c in fr]
    with tempfile.TemporaryDirectory() as td, tempfile.TemporaryDirectory() as td2:
        enc = pmctf_gop.encode_gop(net, frd, H, W, 3, td)
        oenc = pmctf_gop.encode_gop(orc, fr, H, W, 3, td2)
        for i, (r, o) in enumerate(zip(enc["results"], oenc["results"])):
            for k in o["files"]:
                assert r["files"][k] == o["files"][k], f"pair {i} file {k} differs"
            assert_same(r["mv_hat"], o["mv_hat"], f"pair {i} mv_hat")
            assert_same(r["H_t"], o["H_t"], f"pair {i} H_t")
            assert_same(r["L_t"], o["L_t"], f"pair {i} L_t")
            assert_same(r["H_tc"], o["H_tc"], f"pair {i} H_tc")
        assert enc["bits"] == oenc["bits"]
        rec = pmctf_gop.decode_gop(net, enc["frames_coded"])
        orec = pmctf_gop.decode_gop(orc, oenc["frames_coded"])
        for i in range(4):
            assert_same(rec[i][0], orec[i][0], f"rec {i} luma"); assert_same(rec[i][1], orec[i][1], f"rec {i} chroma")
        ps = pmctf_gop.gop_psnr(rec, frd, H, W)
    g = golden()
    # the fixtures come from the real reference: bpp bit-exact (file sizes), PSNR within 1e-4 dB
    assert enc["bits"] == list(g["gop.bits"])
    assert np.abs(np.array([p["yuv"] for p in ps]) - g["gop.psnr_yuv"]).max() < 1e-4


def test_f32_chain_profile_equals_the_oracle_without_aten_all(cuda):
    """The faster profile "f32-chain" (entropy-parameter networks as one chain from the bias, no thread tails) against the
    oracle's PM-F32 back-end with aten_all=False — the same rules layer for layer: files, reconstructions and the decoder's
    output identical, the sequential LL decoder included, which runs the rules the encoder's one-shot LL network ran.
    (The default profile's equality with the oracle is what every other test of this file checks.)"""
    import pmctf_gop
    from pmctf_oracle.model import Oracle
    net, sd = product_model(1)
    net.precision = "f32-chain"
    assert not net.engine().aten_all
    net.engine().keep_streams = True
    orc = Oracle(sd, 1, "cdef", aten_all=False)
    for (w, h, gop) in ((128, 128, 4), (200, 120, 2)):
        fr = frames(w, h, gop)
        frd = [[y.cuda(), c.cuda()] for y, c in fr]
        with tempfile.TemporaryDirectory() as td, tempfile.TemporaryDirectory() as td2:
            enc = pmctf_gop.encode_gop(net, frd, h, w, 3, td)
            oenc = pmctf_gop.encode_gop(orc, fr, h, w, 3, td2)
            for i, (r, o) in enumerate(zip(enc["results"], oenc["results"])):
                for k in o["files"]:
                    assert r["files"][k] == o["files"][k], f"{w}x{h} pair {i} file {k} differs"
                for k in ("mv_hat", "H_t", "L_t", "H_tc", "L_tc"):
                    assert_same(r[k], o[k], f"{w}x{h} pair {i} {k}")
            rec = pmctf_gop.decode_gop(net, enc["frames_coded"])
            orec = pmctf_gop.decode_gop(orc, oenc["frames_coded"])
            for i in range(gop):
                assert_same(rec[i][0], orec[i][0], f"rec {i} luma"); assert_same(rec[i][1], orec[i][1], f"rec {i} chroma")
            # the real decoder from the files
            dpb = {"mv_feature": None, "ref_mv_y": None}
            e = net.encode_one_stage(frd[0], frd[1], True, dpb, output_path=os.path.join(td, "1.bin"), pic_width=w,
                                     pic_height=h, skip_decoding=True, stage_idx=0, q_index=3)
            d = net.encode_one_stage(frd[0], frd[1], True, dpb, output_path=os.path.join(td, "1.bin"), pic_width=w,
                                     pic_height=h, skip_decoding=False, stage_idx=0, q_index=3)
            for k in ("L_t", "H_t", "L_tc", "H_tc", "mv_hat"):
                assert_same(d[k], e[k], f"decoded {k} vs encoder reconstruction")


def test_decoder_round_trip_and_oracle(setup):
    """skip_decoding=False: bitstreams written by the HIP encoder are decoded by the HIP decoder (LL subband inside the
    persistent AR kernel) to exactly the encoder's reconstruction, and everything equals the oracle's decoder."""
    import os
    net, orc = setup
    fr = frames(W, H, 2)
    dpb = {"mv_feature": None, "ref_mv_y": None}
    frd = [[y.cuda(), c.cuda()] for y, c in fr]
    with tempfile.TemporaryDirectory() as td:
        enc = net.encode_one_stage(frd[0], frd[1], True, dpb, output_path=os.path.join(td, "1.bin"), pic_width=W,
                                   pic_height=H, skip_decoding=True, stage_idx=0, q_index=3)
        r = net.encode_one_stage(frd[0], frd[1], True, dpb, output_path=os.path.join(td, "1.bin"), pic_width=W,
                                 pic_height=H, skip_decoding=False, stage_idx=0, q_index=3)
    o = orc.encode_one_stage(fr[0], fr[1], True, dpb, pic_width=W, pic_height=H, q_index=3, skip_decoding=False)
    for k in o["files"]:
        assert r["files"][k] == o["files"][k], f"file {k} differs from the oracle (decoder-order stream)"
    for k in ("L_t", "H_t", "L_tc", "H_tc", "mv_hat"):
        assert_same(r[k], enc[k], f"decoded {k} vs encoder reconstruction")
        assert_same(r[k], o[k], f"decoded {k} vs oracle decoder")
    assert_same(r["dpb"]["mv_feature"], o["dpb"]["mv_feature"], "decoded mv_feature")
    g = golden()   # the real reference with its decoder in the loop: same sizes, reconstruction within fp noise
    assert np.abs(r["H_t"].cpu().numpy() - g["dec.H_t"]).max() < 2e-3
    assert r["bit_ME"] == g["dec.bits"][2]


def test_still_image_coder_256(setup):
    """BASELINE configs[0]: the pWave++ still-image coder on one 256x256 frame through pWave.compress / decompress
    (Y plane, then three planes as an RGB image), decoder-order stream, against the oracle's coder and decoder."""
    import os
    net, orc = setup
    img = frames(256, 256, 1)[0][0]
    coder = net.lp_coder
    with tempfile.TemporaryDirectory() as td:
        fn = os.path.join(td, "img.bin")
        x_hat = coder.compress(img.cuda(), [1, 1, 256, 256], fn, q_index=3, skip_decoding=False)
        data = open(fn, "rb").read()
        dec = coder.decompress(fn, padding=64, q_index=3)["x_hat"]
        ox, odata, _ = orc.pwave_compress("lp_coder", img, [1, 1, 256, 256], 3, None, skip_decoding=False)
        assert data == odata, "still-image bitstream differs from the oracle"
        assert_same(x_hat, ox, "still image x_hat")
        assert_same(dec, x_hat, "decoded still image vs encoder reconstruction")
        assert_same(dec, orc.pwave_decompress("lp_coder", odata, 64, 3), "decoded still image vs oracle decoder")
        # three planes in one stream (the RGB branch, pWave.py:394-396,459-460,524-525), 128x128
        img = img[:, :, :128, :128].contiguous()
        rgb = torch.cat([img, img.flip(2), img.flip(3)], dim=1)
        x3 = coder.compress(rgb.cuda(), [1, 3, 128, 128], fn, q_index=7, skip_decoding=False)
        d3 = coder.decompress(fn, padding=64, q_index=7)["x_hat"]
        planes = torch.cat([rgb[:, c:c + 1] for c in range(3)], dim=0)
        o3, o3data, _ = orc.pwave_compress("lp_coder", planes, [1, 3, 128, 128], 7, None, skip_decoding=False)
        assert open(fn, "rb").read() == o3data
        assert x3.shape == (1, 3, 128, 128) and d3.shape == (1, 3, 128, 128)
        assert_same(d3, x3, "decoded RGB vs encoder reconstruction")
        assert_same(torch.cat([x3[:, c:c + 1] for c in range(3)], dim=0), o3, "RGB x_hat vs oracle")


def test_two_me_stages_cropped_frame(cuda):
    """num_me_stages=2 parameter tree, second stage (stage_idx=1), a 176x144 picture padded to 256x256, no L coding,
    q_index 0: files and tensors identical to the oracle."""
    import os
    from pmctf_oracle.model import Oracle
    net, sd = product_model(2)
    orc = Oracle(sd, 2, "cdef")
    w, h = 176, 144
    fr = frames(w, h, 2, seed=99)
    assert fr[0][0].shape[-2:] == (256, 256)
    dpb = {"mv_feature": None, "ref_mv_y": None}
    frd = [[y.cuda(), c.cuda()] for y, c in fr]
    with tempfile.TemporaryDirectory() as td:
        r = net.encode_one_stage(frd[0], frd[1], False, dpb, output_path=os.path.join(td, "3.bin"), pic_width=w,
                                 pic_height=h, skip_decoding=True, stage_idx=1, q_index=0)
        names = sorted(os.listdir(td))
    o = orc.encode_one_stage(fr[0], fr[1], False, dpb, pic_width=w, pic_height=h, q_index=0, stage_idx=1)
    assert "0_main.bin" not in names and len(names) == 3
    for k in ("H_t", "H_tc", "mv_hat"):
        assert_same(r[k], o[k], k)
    assert r["bit_H"] == o["bit_H"] == 8 * (len(o["files"]["H"]) + len(o["files"]["Hc"])) and r["bit_L"] is None
    assert r["bit_ME"] == o["bit_ME"] == 8 * len(o["files"]["mv"])


@pytest.mark.parametrize("q_index", [0, 20])
def test_rate_points_match_oracle(setup, q_index):
    """RD sweep end points (q_index 0 and 20; 3 and 12 are covered by the other tests): one pair per rate point, files
    and reconstructions identical to the oracle."""
    import os
    net, orc = setup
    fr = frames(W, H, 2, seed=7)
    dpb = {"mv_feature": None, "ref_mv_y": None}
    frd = [[y.cuda(), c.cuda()] for y, c in fr]
    with tempfile.TemporaryDirectory() as td:
        r = net.encode_one_stage(frd[0], frd[1], True, dpb, output_path=os.path.join(td, "1.bin"), pic_width=W,
                                 pic_height=H, skip_decoding=True, stage_idx=0, q_index=q_index)
    o = orc.encode_one_stage(fr[0], fr[1], True, dpb, pic_width=W, pic_height=H, q_index=q_index)
    for k in o["files"]:
        assert r["files"][k] == o["files"][k], f"q_index {q_index}: file {k} differs"
    for k in ("L_t", "H_t", "L_tc", "H_tc", "mv_hat"):
        assert_same(r[k], o[k], f"q_index {q_index}: {k}")


@pytest.mark.parametrize("size", [(1920, 1080), (3840, 2160), (1366, 768)], ids=["1080p", "2160p", "1366x768"])
def test_full_size_1080p_properties(cuda, size):
    """At the benchmark's full size (1920x1080, 4 ME stages) — and at four times that, and at a size that is a multiple
    of nothing the path tiles by — the oracle is too slow, so check what must hold at any size: (1) the decoder
    reproduces the encoder's reconstruction bit for bit from the written files, (2) the files account for the reported
    bits, (3) the lifting is inverted by inverse_MCTF up to rounding, (4) coding is deterministic (same bytes twice)."""
    import os
    net, _ = product_model(4)
    w, h = size
    fr = frames(w, h, 2, device="cuda", seed=5)
    dpb = {"mv_feature": None, "ref_mv_y": None}
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "1.bin")
        e = net.encode_one_stage(fr[0], fr[1], True, dpb, output_path=out, pic_width=w, pic_height=h,
                                 skip_decoding=True, stage_idx=0, q_index=3)
        first = {n: open(os.path.join(td, n), "rb").read() for n in sorted(os.listdir(td))}
        e2 = net.encode_one_stage(fr[0], fr[1], True, dpb, output_path=out, pic_width=w, pic_height=h,
                                  skip_decoding=True, stage_idx=0, q_index=3)
        again = {n: open(os.path.join(td, n), "rb").read() for n in sorted(os.listdir(td))}
        assert first == again and len(first) == 5
        assert e["bit_H"] + e["bit_L"] + e["bit_ME"] == 8 * sum(len(v) for v in first.values())
        d = net.encode_one_stage(fr[0], fr[1], True, dpb, output_path=out, pic_width=w, pic_height=h,
                                 skip_decoding=False, stage_idx=0, q_index=3)
    for k in ("L_t", "H_t", "L_tc", "H_tc", "mv_hat"):
        assert_same(d[k], e[k], f"1080p decoded {k} vs encoder reconstruction")
        assert_same(e2[k], e[k], f"1080p {k} second run")
    ph, pw = (h + 127) // 128 * 128, (w + 127) // 128 * 128
    assert e["H_t"].shape == (1, 1, ph, pw) and e["H_tc"].shape == (2, 1, ph // 2, pw // 2)
    # (3) analysis -> synthesis with the coded motion field
    L_t, H_t, _, _ = net.forward_MCTF(fr[0][0], fr[1][0], e["mv_hat"])
    ref, cur = net.inverse_MCTF(L_t, H_t, e["mv_hat"])
    assert (ref - fr[0][0]).abs().max().item() < 1e-3 and (cur - fr[1][0]).abs().max().item() < 1e-3


def test_estimate_mode_forward(setup):
    """forward_one_stage (bit estimates, pMCTF_L.py:332-379): tensors bit-exact vs the oracle's PM-F32 restatement,
    scalars to 1e-6 relative (f64 totals of per-element f32 values; the oracle sums in numpy order), and within 1e-4
    relative of the real reference's fixtures."""
    net, orc = setup
    g = golden()
    (Y0, C0), (Y1, C1) = frames(W, H, 2)
    dpb = {"mv_feature": None, "ref_mv_y": None}
    ry = net.forward_one_stage(Y0.cuda(), Y1.cuda(), 3, True, dpb)
    rc = net.forward_one_stage(C0.cuda(), C1.cuda(), 3, True, dpb, mv_hat=ry["mv_hat"])
    rn = net(Y0.cuda(), Y1.cuda(), 12, False, ry["dpb"], stage_idx=0)
    oy = orc.forward_one_stage(Y0, Y1, 3, True, dpb)
    oc = orc.forward_one_stage(C0, C1, 3, True, dpb, mv_hat=oy["mv_hat"])
    on = orc.forward_one_stage(Y0, Y1, 12, False, oy["dpb"], stage_idx=0)
    n_scalars = 0
    for tag, r, o in (("y", ry, oy), ("c", rc, oc), ("n", rn, on)):
        for k, ov in o.items():
            if k == "dpb":
                for kk in ov:
                    if ov[kk] is None:
                        assert r["dpb"][kk] is None
                    else:
                        assert_same(r["dpb"][kk], ov[kk], f"{tag} dpb.{kk}")
            elif ov is None:
                assert r[k] is None, (tag, k)
            elif isinstance(ov, torch.Tensor):
                assert_same(r[k], ov, f"{tag} {k}")
            else:
                pv = float(r[k])
                assert abs(pv - ov) <= 1e-6 * max(1.0, abs(ov)), (tag, k, pv, ov)
                ref = float(g[f"est.{tag}.{k}"])
                assert abs(pv - ref) <= 1e-4 * max(1.0, abs(ref)), (tag, k, pv, ref)
                n_scalars += 1
    assert n_scalars == 34
    # the standalone coder's forward gives the L numbers of the luma pair
    L_t = net.forward_MCTF(Y0.cuda(), Y1.cuda(), ry["mv_hat"])[0]
    f = net.lp_coder(L_t, 3)
    assert abs(float(f["bits_total"]) - oy["bit_L"]) <= 1e-6 * oy["bit_L"]
    assert_same(f["x_hat"], oy["L_t"], "pWave.forward x_hat")


def test_reduced_resolution_motion(setup):
    """me_downsample: motion estimated and coded at 1/2 (128x128, vs oracle and the real reference's files, with the
    standalone motion decoder and the estimate-mode twin) and at 1/4 (256x256, vs oracle; HIP decoder in the loop)."""
    import os
    net, orc = setup
    g = golden()
    fr = frames(W, H, 2)
    frd = [[y.cuda(), c.cuda()] for y, c in fr]
    dpb = {"mv_feature": None, "ref_mv_y": None}
    with tempfile.TemporaryDirectory() as td:
        r = net.encode_one_stage(frd[0], frd[1], False, dpb, output_path=os.path.join(td, "1.bin"), pic_width=W,
                                 pic_height=H, skip_decoding=True, stage_idx=0, q_index=3, me_downsample=2)
        for n in ("1.bin", "1_mv.bin", "1_C_main.bin"):
            assert open(os.path.join(td, n), "rb").read() == g[f"ds2.file.{n}"].tobytes(), n
        from pMCTF.utils.stream_helper import decode_p
        _, string = decode_p(os.path.join(td, "1_mv.bin"))
    o = orc.encode_one_stage(fr[0], fr[1], False, dpb, pic_width=W, pic_height=H, q_index=3, me_downsample=2)
    for k in ("H_t", "H_tc", "mv_hat"):
        assert_same(r[k], o[k], f"ds2 {k}")
        assert np.abs(r[k].cpu().numpy() - g[f"ds2.{k}"]).max() < 2e-3
    d = net.decompress_mv(string, torch.float32, H // 2, W // 2, dpb, stage_idx=0, q_index=3, me_downsample=2)
    assert_same(d["mv_hat"], r["mv_hat"], "ds2 decoded mv_hat")
    assert np.abs(d["mv_feature"].cpu().numpy() - g["ds2.dec.mv_feature"]).max() < 2e-3
    e = net.forward_one_stage(frd[0][0], frd[1][0], 3, False, dpb, me_downsample=2)
    oe = orc.forward_one_stage(fr[0][0], fr[1][0], 3, False, dpb, me_downsample=2)
    for k in ("bpp_mv_y", "bpp_mv_z", "bpp", "bit_H", "me_mse"):
        assert abs(float(e[k]) - oe[k]) <= 1e-6 * max(1.0, abs(oe[k])), k
        assert abs(float(e[k]) - float(g[f"ds2.est.{k}"])) <= 1e-4 * max(1.0, abs(float(g[f"ds2.est.{k}"]))), k
    assert_same(e["mv_hat"], oe["mv_hat"], "ds2 estimate mv_hat")
    # 1/4 resolution needs frames padded to 256 (test_pMCTF_CA.py:121); decoder in the loop
    fr4 = frames(256, 256, 2, seed=3)
    frd4 = [[y.cuda(), c.cuda()] for y, c in fr4]
    with tempfile.TemporaryDirectory() as td:
        r4 = net.encode_one_stage(frd4[0], frd4[1], False, dpb, output_path=os.path.join(td, "1.bin"), pic_width=256,
                                  pic_height=256, psize=256, skip_decoding=False, stage_idx=0, q_index=5, me_downsample=4)
    o4 = orc.encode_one_stage(fr4[0], fr4[1], False, dpb, pic_width=256, pic_height=256, psize=256, q_index=5,
                              me_downsample=4, skip_decoding=False)
    for k in o4["files"]:
        assert r4["files"][k] == o4["files"][k], f"ds4 file {k}"
    for k in ("H_t", "H_tc", "mv_hat"):
        assert_same(r4[k], o4[k], f"ds4 {k}")


def test_config_448x256_lifting_step(setup):
    """BASELINE configs[1]: one temporal-lifting step (SpyNet, warp, predict/update, both spatial coders) on a 448x256
    pair, num_me_stages=1 — every tensor, symbol file and bit count identical to the oracle."""
    import os
    net, orc = setup
    w, h = 448, 256
    fr = frames(w, h, 2, seed=21)
    assert fr[0][0].shape[-2:] == (256, 512)
    frd = [[y.cuda(), c.cuda()] for y, c in fr]
    dpb = {"mv_feature": None, "ref_mv_y": None}
    est = net.engine().spynet(frd[1][0], frd[0][0])
    assert_same(est, orc.spynet(fr[1][0].tile((1, 3, 1, 1)) / 255, fr[0][0].tile((1, 3, 1, 1)) / 255), "SpyNet flow")
    L_t, H_t, pred, inv = net.forward_MCTF(frd[0][0], frd[1][0], est)
    oL, oH, opred, oinv = orc.forward_MCTF(fr[0][0], fr[1][0], est.cpu())
    for a, b, n in ((L_t, oL, "L_t"), (H_t, oH, "H_t"), (pred, opred, "prediction"), (inv, oinv, "update")):
        assert_same(a, b, n)
    with tempfile.TemporaryDirectory() as td:
        r = net.encode_one_stage(frd[0], frd[1], True, dpb, output_path=os.path.join(td, "1.bin"), pic_width=w,
                                 pic_height=h, skip_decoding=True, stage_idx=0, q_index=3)
    o = orc.encode_one_stage(fr[0], fr[1], True, dpb, pic_width=w, pic_height=h, q_index=3)
    for k in o["files"]:
        assert r["files"][k] == o["files"][k], f"file {k}"
    for k in ("L_t", "H_t", "L_tc", "H_tc", "mv_hat"):
        assert_same(r[k], o[k], k)
    assert r["bit_H"] == o["bit_H"] and r["bit_L"] == o["bit_L"] and r["bit_ME"] == o["bit_ME"]
    ref, cur = net.inverse_MCTF(r["L_t"], r["H_t"], r["mv_hat"])
    oref, ocur = orc.inverse_MCTF(o["L_t"], o["H_t"], o["mv_hat"])
    assert_same(ref, oref, "inverse_MCTF ref"); assert_same(cur, ocur, "inverse_MCTF cur")


def test_corrupt_streams_fail_cleanly(setup):
    """A truncated or garbled file must end in an exception or in (wrong) pixels, never in a fault or a hang: CDF rows
    are clamped (NaN included), every stream read is bounds-checked on the host and inside the LL kernel."""
    import os
    net, _ = setup
    img = frames(128, 128, 1)[0][0]
    coder = net.lp_coder
    with tempfile.TemporaryDirectory() as td:
        fn = os.path.join(td, "img.bin")
        coder.compress(img.cuda(), [1, 1, 128, 128], fn, q_index=3, skip_decoding=False)
        good = open(fn, "rb").read()
        rng = np.random.default_rng(0)
        import struct
        cases = {
            "truncated": good[:16 + (len(good) - 16) // 3],
            "garbled payload": good[:24] + rng.integers(0, 256, len(good) - 24, dtype=np.uint8).tobytes(),
            "length field too long": good[:12] + struct.pack(">I", len(good) * 2) + good[16:],
        }
        ref = coder.decompress(fn, padding=64, q_index=3)["x_hat"]
        outcomes = {}
        for name, data in cases.items():
            if name == "truncated":      # keep the header's length field consistent with what is there
                data = data[:12] + struct.pack(">I", len(data) - 16) + data[16:]
            open(fn, "wb").write(data)
            try:
                out = coder.decompress(fn, padding=64, q_index=3)["x_hat"]
                torch.cuda.synchronize()
                assert out.shape == (1, 1, 128, 128), name
                assert torch.isfinite(out).all(), name           # garbage in must not become NaN/inf out
                if name != "length field too long":              # (that file still holds the whole payload)
                    assert not torch.equal(out, ref), name       # and must not be mistaken for the intact picture
                outcomes[name] = "pixels"
            except (ValueError, RuntimeError) as e:
                outcomes[name] = type(e).__name__
        assert set(outcomes) == set(cases), outcomes
        open(fn, "wb").write(good)       # and the coder still works afterwards, bit for bit
        again = coder.decompress(fn, padding=64, q_index=3)["x_hat"]
        assert torch.equal(again, ref)


def test_gop4_448x256_vs_reference(setup):
    """North-star parity bar at a second size, HIP product against the REAL reference's outputs (fixture generated by
    tools/make_golden.py --width 448 --height 256): bits per frame identical (bpp bit-exact), PSNR within 1e-4 dB.
    PM-F32 differs from ATen in the last bit of some conv sums, so a rounding tie may flip a symbol: at this size one
    chroma L file has different bytes of the SAME length; every other file is byte-identical."""
    import pmctf_gop
    from helpers import golden_448
    net, _ = setup
    g = golden_448()
    w, h = 448, 256
    fr = frames(w, h, 4, device="cuda")
    with tempfile.TemporaryDirectory() as td:
        enc = pmctf_gop.encode_gop(net, fr, h, w, 3, td)
        rec = pmctf_gop.decode_gop(net, enc["frames_coded"])
        ps = pmctf_gop.gop_psnr(rec, fr, h, w)
    assert enc["bits"] == g["gop.bits"].tolist(), "bits per frame differ from the reference"
    assert enc["bits_mv"] == g["gop.bits_mv"].tolist()
    assert np.abs(np.array([p["yuv"] for p in ps]) - g["gop.psnr_yuv"]).max() < 1e-4
    assert np.abs(np.array([p["y"] for p in ps]) - g["gop.psnr_y"]).max() < 1e-4
    same = diff = 0
    for i, r in enumerate(enc["results"]):
        cur = int(g[f"gop.pair{i}.meta"][2])
        for name, key in (("mv", f"{cur}_mv.bin"), ("H", f"{cur}.bin"), ("Hc", f"{cur}_C_main.bin"),
                          ("L", "0_main.bin"), ("Lc", "0_C_main.bin")):
            k = f"gop.pair{i}.file.{key}"
            if name in r["files"] and k in g.files:
                assert len(r["files"][name]) == len(g[k]), (i, name)
                if r["files"][name] == g[k].tobytes():
                    same += 1
                else:
                    diff += 1
    print(f"448x256 GOP 4: {same} of {same + diff} files byte-identical to the reference's")
    assert same == 11 and diff == 0, (same, diff)


def test_gop4_960x544_vs_reference(setup):
    """The same bar at a quarter of 1080p (planes padded to 1024x640), against digests of the real reference's GOP-4
    output (tools/make_golden.py --width 960 --height 544 --gop_only; file bytes reduced to SHA-1 + length)."""
    import hashlib
    import pmctf_gop
    net, _ = setup
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_960x544_digest.npz"))
    w, h = 960, 544
    fr = frames(w, h, 4, device="cuda")
    with tempfile.TemporaryDirectory() as td:
        enc = pmctf_gop.encode_gop(net, fr, h, w, 3, td)
        rec = pmctf_gop.decode_gop(net, enc["frames_coded"])
        ps = pmctf_gop.gop_psnr(rec, fr, h, w)
    assert enc["bits"] == g["gop.bits"].tolist(), "bits per frame differ from the reference"
    assert enc["bits_mv"] == g["gop.bits_mv"].tolist()
    assert np.abs(np.array([p["yuv"] for p in ps]) - g["gop.psnr_yuv"]).max() < 1e-4
    same = diff = 0
    for i, r in enumerate(enc["results"]):
        cur = int(g[f"gop.pair{i}.meta"][2])
        for name, key in (("mv", f"{cur}_mv.bin"), ("H", f"{cur}.bin"), ("Hc", f"{cur}_C_main.bin"),
                          ("L", "0_main.bin"), ("Lc", "0_C_main.bin")):
            k = f"gop.pair{i}.filesha1.{key}"
            if name in r["files"] and k in g.files:
                assert len(r["files"][name]) == int(g[k.replace("filesha1", "filelen")]), (i, name)
                if hashlib.sha1(r["files"][name]).digest() == g[k].tobytes():
                    same += 1
                else:
                    diff += 1
    print(f"960x544 GOP 4: {same} of {same + diff} files byte-identical to the reference's")
    assert same == 11 and diff == 0, (same, diff)


def _digest_path(gop, q_index, sequence="pan", size=(1920, 1080), me_downsample=1, weights_seed=0, threads=0):
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                        "reference_%dx%d_gop%d_me4%s%s%s%s%s_digest.npz" % (size[0], size[1], gop, "" if q_index == 3 else f"_q{q_index}",
                                                                            "" if sequence == "pan" else "_" + sequence,
                                                                            "" if me_downsample == 1 else f"_ds{me_downsample}",
                                                                            "" if weights_seed == 0 else f"_w{weights_seed}",
                                                                            "" if not threads else f"_t{threads}"))


# BASELINE configs[2] (GOP 8, q_index 3) and configs[3] (GOP 16, the six points of the RD sweep {0,4,8,12,16,20}) plus the
# benchmark's own point (GOP 16, q_index 3): every one has a digest of the REAL reference's CPU run
# (tools/make_golden.py --width 1920 --height 1080 --gop_only --gop G --me_stages 4 --q_index q; 20-45 min of CPU each)
HEADLINE_CONFIGS = [(g, q) for g, q in ((16, 3), (8, 3), (16, 0), (16, 4), (16, 8), (16, 12), (16, 16), (16, 20))
                    if os.path.exists(_digest_path(g, q))]
# Where PM-F32 and ATen round a conv sum differently and the difference lands on a decision boundary, a symbol or a CDF
# row flips; a flip can move a stream by one 32-bit rANS word.  What was observed on this sequence, per configuration
# (per-frame bit deltas product - reference, max |PSNR error| in dB): pinned exactly, so any drift fails.
HEADLINE_PINS_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "headline_pins.json")
_headline_cache = {}


def _headline_run(gop, q_index, sequence="pan", size=(1920, 1080), me_downsample=1, precision="f32", weights_seed=0,
                  threads=0):
    """threads: the intra-op thread count of the reference run behind the digest (0: 8, as for every fixture without a _tN
    suffix); the product follows it through PMCTF_ATEN_THREADS"""
    import hashlib
    import pmctf_gop
    key = (gop, q_index, sequence, size, me_downsample, precision, weights_seed, threads)
    if key in _headline_cache:
        return _headline_cache[key]
    g = np.load(_digest_path(gop, q_index, sequence, size, me_downsample, weights_seed, threads))
    net, _ = product_model(4, weights_seed=weights_seed)
    net.precision = precision
    old_t = os.environ.get("PMCTF_ATEN_THREADS")
    try:
        if threads:
            os.environ["PMCTF_ATEN_THREADS"] = str(threads)
            net._engine = None
        assert net.engine().precision == precision and (not threads or net.engine().aten_threads == threads)
    finally:
        if threads:
            if old_t is None:
                os.environ.pop("PMCTF_ATEN_THREADS", None)
            else:
                os.environ["PMCTF_ATEN_THREADS"] = old_t
    net.engine().keep_streams = True
    w, h = size
    if sequence == "pan":
        fr = frames(w, h, gop, device="cuda")
    else:
        import pmctf_synth
        fr = [list(pmctf_synth.frames_to_tensors(f, device="cuda")) for f in pmctf_synth.synth_yuv420_layers(w, h, gop)]
    with tempfile.TemporaryDirectory() as td:
        enc = pmctf_gop.encode_gop(net, fr, h, w, q_index, td, me_downsample=me_downsample)
        rec = pmctf_gop.decode_gop(net, enc["frames_coded"])
        ps = pmctf_gop.gop_psnr(rec, fr, h, w)
    same = diff = 0
    lengths_equal = True
    for i, r in enumerate(enc["results"]):
        cur = int(g[f"gop.pair{i}.meta"][2])
        for name, fkey in (("mv", f"{cur}_mv.bin"), ("H", f"{cur}.bin"), ("Hc", f"{cur}_C_main.bin"),
                           ("L", "0_main.bin"), ("Lc", "0_C_main.bin")):
            k = f"gop.pair{i}.filesha1.{fkey}"
            if name in r["files"] and k in g.files:
                lengths_equal &= len(r["files"][name]) == int(g[k.replace("filesha1", "filelen")])
                if hashlib.sha1(r["files"][name]).digest() == g[k].tobytes():
                    same += 1
                else:
                    diff += 1
    out = {"bits": enc["bits"], "bits_mv": enc["bits_mv"], "ref_bits": g["gop.bits"].tolist(),
           "ref_bits_mv": g["gop.bits_mv"].tolist(),
           "dbits": (np.array(enc["bits"]) - g["gop.bits"]).astype(np.int64).tolist(),
           "psnr_err": float(np.abs(np.array([p["yuv"] for p in ps]) - g["gop.psnr_yuv"]).max()),
           "same": same, "diff": diff, "lengths_equal": bool(lengths_equal),
           "bpp": sum(enc["bits"]) / (gop * w * h), "bpp_ref": float(g["gop.bits"].sum()) / (gop * w * h)}
    print(f"{w}x{h} GOP-{gop} q_index {q_index} ({sequence}): bpp {out['bpp']:.6f} (reference {out['bpp_ref']:.6f}), max PSNR error "
          f"{out['psnr_err']:.3e} dB, {same} of {same + diff} files byte-identical, bit deltas {out['dbits']}")
    d = os.environ.get("PMCTF_HEADLINE_REPORT") if precision == "f32" and weights_seed == 0 and not threads else None
    if d:       # builder's measuring run: collect what the pins file is written from
        import json
        os.makedirs(d, exist_ok=True)
        json.dump({k: out[k] for k in ("dbits", "psnr_err", "same", "diff", "lengths_equal", "bpp", "bpp_ref")},
                  open(os.path.join(d, f"gop{gop}_q{q_index}{'' if sequence == 'pan' else '_' + sequence}"
                                       f"{'' if size == (1920, 1080) else '_%dx%d' % size}"
                                       f"{'' if me_downsample == 1 else '_ds%d' % me_downsample}.json"), "w"))
    _headline_cache[key] = out
    del enc, rec, fr
    torch.cuda.empty_cache()
    return out


def _strict_params():
    import json
    pins = json.load(open(HEADLINE_PINS_FILE)) if os.path.exists(HEADLINE_PINS_FILE) else {}
    out = []
    for g, q in HEADLINE_CONFIGS:
        pin = pins.get(f"gop{g}_q{q}")
        marks = []
        if pin is not None and (any(pin["dbits"]) or pin["psnr_err"] >= 1e-4):
            marks = [pytest.mark.xfail(strict=True, reason=f"measured: per-frame bit deltas {pin['dbits']}, max PSNR error "
                                                           f"{pin['psnr_err']:.2e} dB (last-bit rounding of conv sums, "
                                                           f"PM-F32 vs ATen; pinned exactly by the test below)")]
        out.append(pytest.param(g, q, marks=marks, id=f"gop{g}-q{q}"))
    return out


@pytest.mark.parametrize("gop,q_index", _strict_params())
def test_headline_configs_1080p_vs_reference(cuda, gop, q_index):
    """The north star's bar, exactly, at full size against the REAL reference's CPU run of the same synthetic sequence
    and weights: bits of every frame identical (bpp bit-exact), PSNR within 1e-4 dB — for BASELINE configs[2] (GOP 8),
    every point of configs[3]'s RD sweep (GOP 16, q_index 0/4/8/12/16/20) and the benchmark's point (GOP 16, q_index 3).
    A configuration that misses the bar is an expected failure carrying the measured numbers, never a widened bound."""
    r = _headline_run(gop, q_index)
    assert r["same"] + r["diff"] == 3 * (gop - 1) + 2
    assert r["bits"] == r["ref_bits"], f"bits per frame differ from the reference: {r['dbits']}"
    assert r["bits_mv"] == r["ref_bits_mv"]
    assert r["psnr_err"] < 1e-4
    assert r["lengths_equal"]
    # ... and every FILE — motion, H and L pictures, luma and chroma — is the reference's byte for byte (SHA-1 in the digest)
    assert r["diff"] == 0, (r["same"], r["diff"])


@pytest.mark.parametrize("gop,q_index", HEADLINE_CONFIGS, ids=[f"gop{g}-q{q}" for g, q in HEADLINE_CONFIGS])
def test_headline_configs_pinned_deviation(cuda, gop, q_index):
    """Regression pin of what was measured against the reference for every configuration: the exact per-frame bit deltas,
    the PSNR error (the arithmetic is deterministic, so it reproduces to the last digit) and the number of files whose
    bytes differ at equal length."""
    import json
    pins = json.load(open(HEADLINE_PINS_FILE))
    pin = pins[f"gop{gop}_q{q_index}"]
    r = _headline_run(gop, q_index)
    assert r["dbits"] == pin["dbits"]
    assert abs(r["psnr_err"] - pin["psnr_err"]) < 1e-9
    assert (r["same"], r["diff"]) == (pin["same"], pin["diff"])


SECOND_CONFIGS = [(g, q) for g, q in ((8, 3), (8, 0), (8, 4), (8, 8), (8, 12), (8, 16), (8, 20), (16, 3))
                  if os.path.exists(_digest_path(g, q, "layers"))]
SECOND_PINS_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "second_sequence_pins.json")


def _second_params():
    import json
    pins = json.load(open(SECOND_PINS_FILE)) if os.path.exists(SECOND_PINS_FILE) else {}
    out = []
    for g, q in SECOND_CONFIGS:
        pin = pins.get(f"gop{g}_q{q}")
        marks = []
        if pin is not None and (any(pin["dbits"]) or pin["psnr_err"] >= 1e-4):
            marks = [pytest.mark.xfail(strict=True, reason=f"measured: per-frame bit deltas {pin['dbits']}, max PSNR error "
                                                           f"{pin['psnr_err']:.2e} dB (pinned exactly below)")]
        out.append(pytest.param(g, q, marks=marks, id=f"gop{g}-q{q}"))
    return out


@pytest.mark.parametrize("gop,q_index", _second_params())
def test_second_sequence_1080p_vs_reference(cuda, gop, q_index):
    """The same bar on a SECOND synthetic sequence (pmctf_synth.synth_yuv420_layers: two motion layers, an occluding
    square — motion boundaries, occlusion and dis-occlusion, which the panning sequence of the headline does not have),
    1080p against digests of the real reference's CPU runs of it
    (tools/make_golden.py --width 1920 --height 1080 --gop_only --gop G --me_stages 4 --q_index q --sequence layers)."""
    r = _headline_run(gop, q_index, "layers")
    assert r["same"] + r["diff"] == 3 * (gop - 1) + 2
    assert r["bits"] == r["ref_bits"], f"bits per frame differ from the reference: {r['dbits']}"
    assert r["bits_mv"] == r["ref_bits_mv"]
    assert r["psnr_err"] < 1e-4
    assert r["lengths_equal"]


@pytest.mark.parametrize("gop,q_index", SECOND_CONFIGS, ids=[f"gop{g}-q{q}" for g, q in SECOND_CONFIGS])
def test_second_sequence_pinned_deviation(cuda, gop, q_index):
    """what was measured against the reference on the second sequence, pinned exactly (as for the headline sequence)"""
    import json
    pin = json.load(open(SECOND_PINS_FILE))[f"gop{gop}_q{q_index}"]
    r = _headline_run(gop, q_index, "layers")
    assert r["dbits"] == pin["dbits"]
    assert abs(r["psnr_err"] - pin["psnr_err"]) < 1e-9
    assert (r["same"], r["diff"]) == (pin["same"], pin["diff"])


@pytest.mark.skipif(not os.path.exists(_digest_path(2, 3, "layers", (3840, 2160))), reason="4K fixture not generated")
def test_2160p_pair_vs_reference(cuda):
    """Largest size: one 3840x2160 pair (H and L coded, four ME stages, q_index 3) of the second sequence against the
    digest of the real reference's CPU run (tools/make_golden.py --width 3840 --height 2160 --gop_only --digest --gop 2
    --me_stages 4 --sequence layers): bits identical, PSNR within 1e-4 dB."""
    r = _headline_run(2, 3, "layers", (3840, 2160))
    assert r["same"] + r["diff"] == 5
    assert r["bits"] == r["ref_bits"], f"bits per frame differ from the reference: {r['dbits']}"
    assert r["bits_mv"] == r["ref_bits_mv"]
    assert r["psnr_err"] < 1e-4
    assert r["lengths_equal"]


def test_1366x768_gop8_vs_reference(cuda):
    """A frame size that is a multiple of nothing the path tiles by (1366x768 -> padded to 1408x768; chroma 683x384): GOP 8,
    q_index 3, four ME stages against the digest of the real reference's CPU run (tools/make_golden.py --gop_only --width
    1366 --height 768 --gop 8 --me_stages 4): bits of every frame identical, PSNR within 1e-4 dB."""
    r = _headline_run(8, 3, size=(1366, 768))
    assert r["same"] + r["diff"] == 3 * 7 + 2
    assert r["bits"] == r["ref_bits"], f"bits per frame differ from the reference: {r['dbits']}"
    assert r["bits_mv"] == r["ref_bits_mv"]
    assert r["psnr_err"] < 1e-4
    assert r["lengths_equal"]


def test_1366x768_gop8_pinned_deviation(cuda):
    """What the strict test above measured, pinned: bits of every frame and of every motion stream equal to the
    reference's; the count of files whose bytes differ at equal length (symbols / CDF rows inside, from the
    entropy-parameter networks' last bits) may only move with a change of the arithmetic."""
    r = _headline_run(8, 3, size=(1366, 768))
    assert r["dbits"] == [0] * 8 and r["bits_mv"] == r["ref_bits_mv"]
    assert r["psnr_err"] < 1e-4 and r["same"] == 23 and r["diff"] == 0


def test_1080p_gop8_reduced_resolution_motion_vs_reference(cuda):
    """The content-adaptive script's GOP schedule at full size: 1920x1080, GOP 8, q_index 3, motion estimated and coded
    at HALF resolution (me_downsample=2, test_pMCTF_CA.py:code_one_gop) against the digest of the real reference's CPU run
    (tools/make_golden.py --gop_only --width 1920 --height 1080 --gop 8 --me_stages 4 --me_downsample 2): bits of every
    frame identical, PSNR within 1e-4 dB."""
    r = _headline_run(8, 3, me_downsample=2)
    assert r["same"] + r["diff"] == 3 * 7 + 2
    assert r["bits"] == r["ref_bits"], f"bits per frame differ from the reference: {r['dbits']}"
    assert r["bits_mv"] == r["ref_bits_mv"]
    assert r["psnr_err"] < 1e-4
    assert r["lengths_equal"]


_ATEN_OTHER = [c for c in ((8, 3, "layers", (1920, 1080), 1), (16, 3, "layers", (1920, 1080), 1), (8, 3, "pan", (1366, 768), 1),
                            (2, 3, "layers", (3840, 2160), 1), (8, 3, "pan", (1920, 1080), 2), (4, 3, "layers", (1280, 720), 1),
                            (4, 12, "pan", (832, 480), 1), (2, 3, "layers", (2560, 1440), 1),
                            (4, 3, "layers", (640, 360), 1), (4, 3, "pan", (416, 240), 1),
                            (8, 10, "layers", (1920, 1080), 2), (8, 16, "layers", (1366, 768), 1))
               if os.path.exists(_digest_path(c[0], c[1], c[2], c[3], c[4]))]


@pytest.mark.parametrize("gop,q_index,sequence,size,ds", _ATEN_OTHER,
                         ids=[f"gop{c[0]}-{c[2]}-{c[3][0]}x{c[3][1]}-ds{c[4]}" for c in _ATEN_OTHER])
def test_every_file_is_the_reference_s_other_sequences_and_sizes(cuda, gop, q_index, sequence, size, ds):
    """The other reference configurations that have digests — the second sequence (occlusion),
    1366x768 (planes small enough that ATen leaves oneDNN for some layers: the "gemm" / "gemv 3x3" rules), a 3840x2160
    pair, motion at half resolution, and five configurations the summation rules were NOT fitted on, their digests
    generated from the real reference after the rules were final (1280x720, 832x480 at q 12, one 2560x1440 pair, 640x360,
    416x240, 1080p GOP 8 at q 10 with half-resolution motion, 1366x768 GOP 8 at q 16 on the second sequence): every file
    byte-identical to the reference's, PSNR within 1e-4 dB."""
    r = _headline_run(gop, q_index, sequence, size, ds)
    assert r["diff"] == 0 and r["same"] == 3 * (gop - 1) + 2, (r["same"], r["diff"])
    assert r["bits"] == r["ref_bits"] and r["bits_mv"] == r["ref_bits_mv"]
    assert r["psnr_err"] < 1e-4


@pytest.mark.skipif(not os.path.exists(_digest_path(4, 8, "layers", (1920, 1080), 1, 1)), reason="fixture not generated")
def test_every_file_is_the_reference_s_with_other_weights(cuda):
    """Every fixture above uses ONE set of synthetic weights (seed 0).  Here another set (seed 1), run through the real
    reference after the arithmetic was final (tools/make_golden.py --weights_seed 1: 1080p GOP 4, q_index 8, second
    sequence): every file byte-identical, PSNR within 1e-4 dB."""
    r = _headline_run(4, 8, "layers", (1920, 1080), 1, weights_seed=1)
    assert r["diff"] == 0 and r["same"] == 11, (r["same"], r["diff"])
    assert r["bits"] == r["ref_bits"] and r["bits_mv"] == r["ref_bits_mv"]
    assert r["psnr_err"] < 1e-4


@pytest.mark.parametrize("threads", [t for t in (4, 16) if os.path.exists(_digest_path(2, 3, threads=t))])
def test_every_file_is_the_reference_s_at_other_thread_counts(cuda, threads):
    """The reference's own results depend on its intra-op thread count in one place — which elements of torch.sigmoid take
    ATen's scalar tail (the convolution rules are the same for 2, 4, 8 and 16 threads: measured on the CPU); PMCTF_ATEN_THREADS
    names it.  A 1080p pair coded by the real reference with 4 and with 16 threads: every file byte-identical."""
    r = _headline_run(2, 3, threads=threads)
    assert r["diff"] == 0 and r["same"] == 5, (r["same"], r["diff"])
    assert r["bits"] == r["ref_bits"] and r["bits_mv"] == r["ref_bits_mv"]
    assert r["psnr_err"] < 1e-4


def test_gop_with_reduced_resolution_motion(setup):
    """The content-adaptive harness's GOP schedule (test_pMCTF_CA.py:code_one_gop) with me_downsample=2: a GOP of 4
    through the same loop on the product and on the oracle — identical bits, files and reconstruction."""
    import pmctf_gop
    net, orc = setup
    fr = frames(W, H, 4, seed=9)
    frd = [[y.cuda(), c.cuda()] for y, c in fr]
    assert pmctf_gop.ca_psize(2) == 128 and pmctf_gop.ca_psize(4) == 256 and pmctf_gop.ca_psize(8) == 512
    with tempfile.TemporaryDirectory() as td:
        enc = pmctf_gop.encode_gop(net, frd, H, W, 3, td, me_downsample=2)
        rec = pmctf_gop.decode_gop(net, enc["frames_coded"])
    with tempfile.TemporaryDirectory() as td:
        oenc = pmctf_gop.encode_gop(orc, fr, H, W, 3, td, me_downsample=2)
        orec = pmctf_gop.decode_gop(orc, oenc["frames_coded"])
    assert enc["bits"] == oenc["bits"] and enc["bits_mv"] == oenc["bits_mv"]
    for r, o in zip(enc["results"], oenc["results"]):
        for k in o["files"]:
            assert r["files"][k] == o["files"][k], k
    for (ry, rc, _), (oy, oc, _) in zip(rec, orec):
        assert_same(ry, oy, "reconstructed luma"); assert_same(rc, oc, "reconstructed chroma")


def test_batched_stage_equals_pair_by_pair(cuda):
    """encode_stage_pairs (all pairs of a temporal stage as one batch) against the pair-by-pair harness schedule:
    GOP 8, four ME stages, 128x128 — every file, bit count and tensor identical."""
    import pmctf_gop
    net, _ = product_model(4)
    net.engine().keep_streams = True
    fr = frames(W, H, 8, device="cuda", seed=13)
    with tempfile.TemporaryDirectory() as td:
        ref = pmctf_gop.encode_gop(net, fr, H, W, 3, td)
        ref_files = {n: open(os.path.join(td, n), "rb").read() for n in sorted(os.listdir(td))}
    with tempfile.TemporaryDirectory() as td:
        bat = pmctf_gop.encode_gop_batched(net, fr, H, W, 3, td)
        bat_files = {n: open(os.path.join(td, n), "rb").read() for n in sorted(os.listdir(td))}
    assert bat["bits"] == ref["bits"] and bat["bits_mv"] == ref["bits_mv"]
    assert bat_files.keys() == ref_files.keys() and len(ref_files) == 3 * 7 + 2
    for n in ref_files:
        assert bat_files[n] == ref_files[n], n
    for a, b in zip(bat["frames_coded"], ref["frames_coded"]):
        for x, y in zip(a, b):
            assert (x is None and y is None) or torch.equal(x, y)
    for rb, rr in zip(bat["results"], ref["results"]):
        for k in rr["files"]:
            assert rb["files"][k] == rr["files"][k], k
            assert np.array_equal(rb["traces"][k][0], rr["traces"][k][0]) and \
                np.array_equal(rb["traces"][k][1], rr["traces"][k][1]), k


def test_sequence_driver_1080p_vs_reference(cuda):
    """What the unmodified evaluation script relies on, asserted as a CONTRACT on the headline configuration (1920x1080,
    GOP 16, q_index 3, four ME stages, write_stream + skip_decoding), the product in its DEFAULT mode, driven by
    pmctf_gop.encode_sequence (.yuv file -> YUVReader -> padding -> one encode_one_stage call per pair, the two per-pair
    report lines formatted before the next call -> synthesis -> PSNR -> log record):
      * every call returns the reference's result keys, tensors as torch.Tensor, bit counts / times as Python floats,
        the five (three) files of the pair on disk with their sizes behind the bit counts;
      * nothing was deferred or batched (no encode_stage_pairs), from the second GOP on every pair replays a plan;
      * bits of every frame identical to the real reference's CPU run, PSNR within 1e-4 dB;
      * the log record has the script's keys and is plain JSON."""
    import json
    import pmctf_gop
    import pmctf_synth
    from pMCTF.hip import pair_plan
    from pMCTF.models.video.pMCTF_L import pMCTF
    g = np.load(_digest_path(16, 3))
    net = pMCTF(num_me_stages=4).eval()                     # what the script builds: ctor, strict load, .to, update
    net.load_state_dict(pmctf_synth.synth_state_dict(net.state_dict(), seed=0), strict=True)
    net = net.to("cuda").eval()
    net.update(force=True)
    w, h, gop = 1920, 1080, 16
    seen, batches, runs = [], [], []
    orig_one, orig_stage, orig_run = net.encode_one_stage, net.encode_stage_pairs, pair_plan.PairPlan.run
    keys = {"L_t", "H_t", "L_tc", "H_tc", "bit_H", "bit_L", "bit_Lc", "bit_Hc", "bit_ME", "mv_hat", "dpb",
            "decoding_time", "encoding_time"}

    def spy(**kw):
        r = orig_one(**kw)
        out = kw["output_path"]
        files = [out, out.replace(".bin", "_mv.bin"), out.replace(".bin", "_C_main.bin")]
        assert all(os.path.getsize(f) > 0 for f in files), "a pair's files exist when its call returns"
        assert r["bit_H"] + r["bit_ME"] == 8.0 * sum(os.path.getsize(f) for f in files)
        seen.append(r)
        return r
    net.encode_one_stage = spy
    net.encode_stage_pairs = lambda pairs, *a, **k: (batches.append(len(pairs)), orig_stage(pairs, *a, **k))[1]
    pair_plan.PairPlan.run = lambda self, *a, **k: (runs.append(self), orig_run(self, *a, **k))[1]
    try:
        with tempfile.TemporaryDirectory() as td:
            yuv = os.path.join(td, "Synth_1920x1080_120fps_420_8bit_YUV.yuv")
            fr8 = pmctf_synth.synth_yuv420(w, h, gop)
            pmctf_gop.write_yuv(yuv, fr8 + fr8)             # two GOPs of the same frames: the second one only replays
            bins = os.path.join(td, "bin")
            os.makedirs(bins)
            out = pmctf_gop.encode_sequence(net, yuv, w, h, 2 * gop, gop, 3, bins, "cuda")
            assert len(os.listdir(bins)) == 3 * (gop - 1) + 2
    finally:
        pair_plan.PairPlan.run = orig_run
    assert batches == [] and len(seen) == 2 * (gop - 1)
    assert len(runs) == (gop - 1) - 7 + (gop - 1)           # first GOP: 7 configurations recorded, the rest replayed
    for r in seen:                                          # finished values, call by call
        assert keys <= set(r)
        assert all(type(r[k]) is torch.Tensor for k in ("L_t", "H_t", "L_tc", "H_tc", "mv_hat"))
        assert all(type(r[k]) is float for k in ("bit_H", "bit_Hc", "bit_ME", "encoding_time"))
        assert all(type(v) is torch.Tensor for v in r["dpb"].values())
    ref_bits = g["gop.bits"].tolist()
    bits, psnrs = out["bits"], out["psnr"]
    assert bits[:gop] == ref_bits and bits[gop:] == ref_bits
    assert np.abs(np.array(psnrs[:gop]) - g["gop.psnr_yuv"]).max() < 1e-4
    assert psnrs[gop:] == psnrs[:gop]
    parsed = json.loads(out["json"])
    assert {"i_frame_num", "p_frame_num", "ave_all_frame_bpp", "ave_all_frame_psnr", "frame_bpp", "frame_psnr",
            "frame_type", "test_time"} <= set(parsed)
    assert parsed["i_frame_num"] == 2 and parsed["p_frame_num"] == 2 * (gop - 1)
    assert abs(parsed["ave_all_frame_bpp"] - float(g["gop.bits"].sum()) / (gop * w * h)) < 1e-6
    assert sum(l.startswith("percentage MV") for l in out["lines"]) == 2 * (gop - 1)


def test_pair_plan_equals_stream_launches(cuda):
    """encode_one_stage replays captured launch plans (HIP graphs, luma / chroma coders on two streams) from the second
    pair of a configuration on (pMCTF.hip.pair_plan).  GOP 8 with four ME stages at 128x128, coded three times: stream
    launches only (plans off), the GOP that records the plans, and a GOP that only replays — every file, bit count,
    symbol trace and tensor identical, and the replayed GOP really went through the plans."""
    import pmctf_gop
    from pMCTF.hip import pair_plan
    net, _ = product_model(4)
    eng = net.engine()
    eng.keep_streams = True
    fr = frames(W, H, 8, device="cuda", seed=29)
    fr2 = frames(W, H, 8, device="cuda", seed=31)

    def gop(frames_):
        with tempfile.TemporaryDirectory() as td:
            enc = pmctf_gop.encode_gop(net, frames_, H, W, 3, td)
            return enc, {n: open(os.path.join(td, n), "rb").read() for n in sorted(os.listdir(td))}
    eng.use_graphs = False
    ref, ref_files = gop(fr)
    ref2, ref2_files = gop(fr2)
    eng.use_graphs = True
    eng.pair_plans.clear()
    runs = []
    orig = pair_plan.PairPlan.run
    pair_plan.PairPlan.run = lambda self, *a, **k: (runs.append(self), orig(self, *a, **k))[1]
    try:
        first, first_files = gop(fr)          # stage 0: pair 0 and pair 1 record (unchained / chained), pairs 2, 3 replay
        assert len(runs) == 2 + 0 + 0 and len(eng.pair_plans) == 2 + 2 + 1
        del runs[:]
        second, second_files = gop(fr2)       # other frames: every pair replays a plan recorded on the first GOP
        assert len(runs) == 7
    finally:
        pair_plan.PairPlan.run = orig
    for enc, files, r, rf in ((first, first_files, ref, ref_files), (second, second_files, ref2, ref2_files)):
        assert files == rf and len(files) == 3 * 7 + 2
        assert enc["bits"] == r["bits"] and enc["bits_mv"] == r["bits_mv"]
        for a, b in zip(enc["frames_coded"], r["frames_coded"]):
            for x, y in zip(a, b):
                assert (x is None and y is None) or torch.equal(x, y)
        for ra, rb in zip(enc["results"], r["results"]):
            assert torch.equal(ra["dpb"]["mv_feature"], rb["dpb"]["mv_feature"])
            assert torch.equal(ra["dpb"]["ref_mv_y"], rb["dpb"]["ref_mv_y"])
            for k in rb["files"]:
                assert ra["files"][k] == rb["files"][k], k
                assert np.array_equal(ra["traces"][k][0], rb["traces"][k][0]) and \
                    np.array_equal(ra["traces"][k][1], rb["traces"][k][1]), k
    rec = pmctf_gop.decode_gop(net, [list(f) for f in second["frames_coded"]])
    rec_ref = pmctf_gop.decode_gop(net, [list(f) for f in ref2["frames_coded"]])
    for a, b in zip(rec, rec_ref):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_dropped_model_never_destroys_graphs_inside_another_recording(cuda):
    """The failure behind last round's intermittent abort, as a deterministic sequence: model A records launch plans and
    is dropped WITHOUT release() while its engine sits in a reference cycle (so nothing dies by reference count); model
    B then records its first plan and a full collection is forced INSIDE B's open capture — exactly when the collector
    used to find A's engine and destroy its HIP graphs, which ends the process.  Now the plans' device objects are
    co-owned by pair_plan.RESOURCES: the collection only retires them (counted), nothing is destroyed while the capture
    is open, B's plan records and replays with the stream path's bits, and the graphs go at the next safe point.  A
    second thread dropping the last reference to a plan during a recording takes the same route.  Recording also leaves
    the process-wide dispatcher knobs and the cyclic collector alone."""
    import gc
    import threading
    import pmctf_gop
    from pMCTF.hip import lib as hiplib
    from pMCTF.hip import pair_plan
    R = pair_plan.RESOURCES
    L = hiplib.hip()
    knobs = {k: L.pmctf_conv2d_get_option(k) for k in (b"SPLIT", b"MSPLIT_PX")}
    fr = frames(W, H, 4, device="cuda", seed=41)

    def gop(net):
        with tempfile.TemporaryDirectory() as td:
            enc = pmctf_gop.encode_gop(net, fr, H, W, 3, td)
            return enc["bits"], {n: open(os.path.join(td, n), "rb").read() for n in sorted(os.listdir(td))}

    gc.collect()
    R.drain()
    net_a, _ = product_model(1)
    bits_a, files_a = gop(net_a)                    # pairs 2 and 3 record plans ("later pair of the chain", "with L")
    eng_a = net_a.engine()
    n_plans = len(eng_a.pair_plans)
    assert n_plans >= 2
    spare = next(iter(eng_a.pair_plans.values()))   # one plan whose LAST reference a worker thread will drop
    eng_a.cycle = eng_a                             # the engine is garbage only the cyclic collector finds
    net_a._engine = None                            # ... and the model's finaliser cannot release it
    gc.disable()                                    # (the TEST holds the collector back so that it strikes inside the capture)
    del net_a, eng_a
    before = dict(R.stats)
    events = []

    def in_capture():
        pair_plan.IN_CAPTURE_HOOK = None            # the first recording only
        assert torch.cuda.is_current_stream_capturing()
        destroyed = R.stats["destroyed"]
        gc.collect()                                # finds A's engine: its plans retire, their graphs stay alive
        box = [spare_holder.pop()]
        t = threading.Thread(target=box.clear)      # another thread lets go of the last reference to a plan
        t.start()
        t.join()
        assert R.drain() == 0                       # a drain attempt inside a recording is a no-op
        events.append((R.stats["retired_while_recording"], R.stats["destroyed"] - destroyed, len(R.retired)))
    spare_holder = [spare]
    del spare
    pair_plan.IN_CAPTURE_HOOK = in_capture
    try:
        net_b, _ = product_model(1)
        bits_b, files_b = gop(net_b)
    finally:
        pair_plan.IN_CAPTURE_HOOK = None
        gc.enable()
    assert events, "model B recorded no plan"
    retired_in_capture, destroyed_in_capture, parked = events[0]
    assert retired_in_capture - before["retired_while_recording"] >= n_plans and destroyed_in_capture == 0 and parked >= n_plans
    assert bits_b == bits_a and files_b == files_a
    assert net_b.engine().use_graphs and len(net_b.engine().pair_plans) >= 2        # no recording failed
    assert gc.isenabled()
    assert {k: L.pmctf_conv2d_get_option(k) for k in knobs} == knobs
    bits_c, files_c = gop(net_b)                    # replays B's plans
    assert bits_c == bits_a and files_c == files_a
    net_b._drop_engine()                            # a safe point: everything retired so far is destroyed
    gc.collect()
    R.drain()
    assert not R.retired and R.stats["destroyed"] - before["destroyed"] >= n_plans
    assert len(R.broken) == 0

    # a recording that FAILS half-way (here: an exception inside the open capture) must leave no stream capturing: the
    # engine warns, goes on with stream launches and produces the same files
    def boom():
        pair_plan.IN_CAPTURE_HOOK = None
        raise RuntimeError("injected failure inside an open capture")
    pair_plan.IN_CAPTURE_HOOK = boom
    try:
        net_c, _ = product_model(1)
        with pytest.warns(UserWarning, match="recording the launch plan failed"):
            bits_d, files_d = gop(net_c)
    finally:
        pair_plan.IN_CAPTURE_HOOK = None
    torch.cuda.synchronize()                        # would raise "operation not permitted when stream is capturing"
    assert not net_c.engine().use_graphs and bits_d == bits_a and files_d == files_a
    net_c._drop_engine()
    R.drain()
    assert not R.retired


@pytest.mark.parametrize("gop", [16, 8])
def test_headline_config_stage_batched_vs_reference(cuda, gop):
    """The stage-batched schedule (pairs of a temporal stage as one batch; bench.py's auxiliary figure) on the headline
    configuration and on BASELINE configs[2] (GOP 8), against the real reference's digests: bits of every frame
    identical, PSNR within 1e-4 dB, and the very same files as the pair-by-pair schedule."""
    import hashlib
    import pmctf_gop
    g = np.load(_digest_path(gop, 3))
    net, _ = product_model(4)
    net.engine().keep_streams = True
    w, h = 1920, 1080
    fr = frames(w, h, gop, device="cuda")
    with tempfile.TemporaryDirectory() as td:
        enc = pmctf_gop.encode_gop_batched(net, fr, h, w, 3, td)
        batched = {n: hashlib.sha1(open(os.path.join(td, n), "rb").read()).hexdigest() for n in sorted(os.listdir(td))}
        rec = pmctf_gop.decode_gop(net, enc["frames_coded"])
        ps = pmctf_gop.gop_psnr(rec, fr, h, w)
    assert enc["bits"] == g["gop.bits"].tolist() and enc["bits_mv"] == g["gop.bits_mv"].tolist()
    assert np.abs(np.array([p["yuv"] for p in ps]) - g["gop.psnr_yuv"]).max() < 1e-4
    del enc, rec
    with tempfile.TemporaryDirectory() as td:
        pmctf_gop.encode_gop(net, fr, h, w, 3, td)
        paired = {n: hashlib.sha1(open(os.path.join(td, n), "rb").read()).hexdigest() for n in sorted(os.listdir(td))}
    assert batched == paired and len(batched) == 3 * (gop - 1) + 2


def test_bench_contract_small_run(cuda):
    """bench.py prints ONE JSON line with the driver's fields (exercised on a tiny GOP so that it takes seconds)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--width", "256", "--height", "128", "--gop", "4",
                          "--steps", "1", "--warmup", "1", "--no_cpu_baseline"], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["unit"] == "frames/s" and j["n_gpus"] == 1 and j["steps"] == 1 and j["higher_is_better"] is True
    assert j["dtype"] == "f32" and j["data"] == "synthetic" and "workload" in j["config"] and j["value"] > 0
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in j["roofline"], k


@pytest.mark.parametrize("gop,size", [(2, (128, 128)), (32, (128, 128)), (4, (200, 120))])
def test_batched_schedule_other_gop_sizes(cuda, gop, size):
    """GOP 2 (one stage), GOP 32 (five stages, 16 pairs in the first batch, more stages than motion models) and a frame
    size that needs padding: stage-batched files, bits and subband tree equal the pair-by-pair schedule."""
    import pmctf_gop
    net, _ = product_model(4)
    w, h = size
    fr = frames(w, h, gop, device="cuda", seed=gop)
    with tempfile.TemporaryDirectory() as td:
        a = pmctf_gop.encode_gop(net, fr, h, w, 5, td)
        fa = {n: open(os.path.join(td, n), "rb").read() for n in os.listdir(td)}
    with tempfile.TemporaryDirectory() as td:
        b = pmctf_gop.encode_gop_batched(net, fr, h, w, 5, td)
        fb = {n: open(os.path.join(td, n), "rb").read() for n in os.listdir(td)}
    assert a["bits"] == b["bits"] and a["bits_mv"] == b["bits_mv"] and fa == fb and len(fa) == 3 * (gop - 1) + 2
    for p, q in zip(a["frames_coded"], b["frames_coded"]):
        for x, y in zip(p, q):
            assert (x is None and y is None) or torch.equal(x, y)


def test_ll_params_and_conv_lstm_context_bitexact(setup):
    """§8 a11 / a13 on their own: the LL subband's masked-conv parameter network and the conv-LSTM subband context
    (three cells, the init_sequential quirk of a 1-channel LSTM3 cell state, the nearest-x2 + conv state upsampling after
    an `hh` subband) against the oracle, tensor by tensor."""
    net, orc = setup
    eng = net.engine()
    g = golden()
    ll = torch.round(torch.from_numpy(g["unit.dwt.ll"])[:, :, :16, :16].contiguous() * 3.0)
    p = eng.context_fusion_ll("hp_coder", ll.cuda())
    op = orc.context_fusion_ll("hp_coder", ll)
    assert_same(p.permute(0, 3, 1, 2), op, "LL parameters (scale, mean)")
    # context: LL at level 3, then lh/hl/hh of level 3 (upsampling of all six states after hh), then lh of level 2
    st = eng.ctx_init(1, 16, 16)
    orc.ctx_init((1, 1, 16, 16))
    rng = np.random.default_rng(5)
    seq = [("ll", 3, 16), ("lh", 3, 16), ("hl", 3, 16), ("hh", 3, 16), ("lh", 2, 32)]
    for name, lvl, n in seq:
        sb = torch.from_numpy(rng.normal(0, 4, (1, 1, n, n)).astype(np.float32))
        c = eng.ctx_forward_one_subband("hp_coder", st, sb.cuda(), name, lvl)
        oc = orc.ctx_forward_one_subband("hp_coder", sb, name, lvl)
        assert_same(c.permute(0, 3, 1, 2), oc, f"context after {name}{lvl}")
        for key, ost in (("l1", orc.l1), ("l2", orc.l2), ("l3", orc.l3)):
            assert_same(st[key][0].permute(0, 3, 1, 2), ost[0], f"{key} hidden after {name}{lvl}")
            assert_same(st[key][1].permute(0, 3, 1, 2), ost[1], f"{key} cell after {name}{lvl}")


def test_advance_dpb_equals_encode_one_stage_dpb(cuda):
    """pMCTF.advance_dpb (the motion part of a pair only) hands on exactly the `dpb` that encode_one_stage returns —
    what pair-level sharding rests on.  Two chained pairs, second motion model (stage_idx 1)."""
    net, _ = product_model(2)
    fr = frames(W, H, 4, device="cuda", seed=17)
    dpb0 = {"mv_feature": None, "ref_mv_y": None}
    with tempfile.TemporaryDirectory() as td:
        for stage_idx in (0, 1):
            r1 = net.encode_one_stage(fr[0], fr[1], False, dpb0, output_path=os.path.join(td, "1.bin"), pic_width=W,
                                      pic_height=H, skip_decoding=True, stage_idx=stage_idx, q_index=3)
            a1 = net.advance_dpb(fr[0], fr[1], dpb0, stage_idx=stage_idx, q_index=3)
            r2 = net.encode_one_stage(fr[2], fr[3], False, r1["dpb"], output_path=os.path.join(td, "3.bin"), pic_width=W,
                                      pic_height=H, skip_decoding=True, stage_idx=stage_idx, q_index=3)
            a2 = net.advance_dpb(fr[2], fr[3], a1, stage_idx=stage_idx, q_index=3)
            for k in ("mv_feature", "ref_mv_y"):
                assert_same(a1[k], r1["dpb"][k], f"stage {stage_idx} first pair dpb.{k}")
                assert_same(a2[k], r2["dpb"][k], f"stage {stage_idx} chained pair dpb.{k}")


def _cross_decode(net):
    """decode the five files the REAL reference wrote for one 128x128 pair (fixture `dec.file.*`) with the HIP decoder"""
    from pMCTF.utils.stream_helper import decode_p
    g = golden()
    dpb = {"mv_feature": None, "ref_mv_y": None}
    with tempfile.TemporaryDirectory() as td:
        for n in ("1.bin", "1_mv.bin", "1_C_main.bin", "0_main.bin", "0_C_main.bin"):
            open(os.path.join(td, n), "wb").write(g[f"dec.file.{n}"].tobytes())
        _, string = decode_p(os.path.join(td, "1_mv.bin"))
        mv = net.decompress_mv(string, torch.float32, H, W, dpb, stage_idx=0, q_index=3)
        luma = net.decompress_one_stage(os.path.join(td, "1.bin"), True, False, psize=128, q_index=3, stage_idx=0)
        chroma = net.decompress_one_stage(os.path.join(td, "1_C_main.bin"), True, True, psize=128, q_index=3, stage_idx=0)
    got = {"mv_hat": mv["mv_hat"], "mv_feature": mv["mv_feature"], "H_t": luma["H_t"]["x_hat"],
           "L_t": luma["L_t"]["x_hat"], "H_tc": chroma["H_t"]["x_hat"], "L_tc": chroma["L_t"]["x_hat"]}
    return {k: float(np.abs(v.cpu().numpy() - g[f"dec.{k}"]).max()) for k, v in got.items()}


def test_cross_decode_reference_written_files(setup):
    """Streams written by the REAL reference on the CPU (the `dec.file.*` arrays of the 128x128 fixture: one pair, H + L,
    decoder-order LL) decoded by the HIP decoder.  A learned codec's decoder must compute every CDF row exactly as the
    encoder did; across two float implementations (ATen vs PM-F32, last-bit differences of conv sums) that holds only
    as long as no scale lands on a bin boundary.  Here: the motion stream, both H streams and the chroma L stream
    decode in step with the reference's range coder and land on the reference decoder's tensors (fp noise only)."""
    err = _cross_decode(setup[0])
    for k in ("mv_hat", "mv_feature", "H_t", "H_tc", "L_tc"):
        assert err[k] < 2e-3, (k, err[k])      # a desynchronised stream decodes to noise (hundreds of grey levels)


def test_cross_decode_reference_written_luma_L_stream(setup):
    """... and the luma L stream (0_main.bin, 5 985 bytes): it desynchronised (509 grey levels) as long as the entropy
    parameters were PM-F32's own chain sums — one CDF row differed from ATen's; with ATen's summation order in the
    entropy-parameter networks (the default profile) the reference-written stream decodes in step."""
    assert _cross_decode(setup[0])["L_t"] < 2e-3


def _pair_shard_gpu_worker(rank, world, port, q):
    import torch.distributed as dist
    import pmctf_dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    net, _ = product_model(4)
    fr = frames(W, H, 8, device="cuda", seed=11)
    fr2 = frames(W, H, 8, device="cuda", seed=12)
    with tempfile.TemporaryDirectory() as td:
        enc = pmctf_dist.encode_gop_pair_sharded(net, fr, H, W, 3, td, rank, world, dist)
        # and two closed GOPs in flight, their chains in opposite directions (SURVEY 8e's GOP overlap); the second time
        # round the pairs replay captured launch plans, relay and all
        os.makedirs(os.path.join(td, "b"))
        for _ in range(2):
            both = pmctf_dist.encode_gops_pair_sharded_overlapped(net, [fr, fr2], H, W, 3, [td, os.path.join(td, "b")],
                                                                  rank, world, dist)
    torch.cuda.synchronize()
    pack = lambda e: (e["bits"], e["bits_mv"], [[t if t is None else t.cpu().numpy() for t in fc] for fc in e["frames_coded"]],
                      len(e["results"]))
    q.put((rank,) + pack(enc) + (pack(both[0]), pack(both[1])))
    dist.barrier()
    dist.destroy_process_group()


def test_pair_parts_equal_encode_one_stage(setup):
    """pmctf_dist lets the ranks a temporal stage leaves idle take PARTS of its pairs: the motion, and the four spatial
    coder calls that are independent given mv_hat (pMCTF_L.py:398-420,570-592).  encode_pair_motion + encode_pair_part for
    (luma, chroma) x (H, L), each on its own, must write encode_one_stage's files and return its tensors and bit counts —
    for a pair that codes L (all four parts, and the two-part cut) and for one that does not."""
    net, _ = setup
    fr = frames(W, H, 2, device="cuda", seed=53)
    for code_lt in (True, False):
        with tempfile.TemporaryDirectory() as td:
            ref = net.encode_one_stage(ref_frame=fr[0], cur_frame=fr[1], output_path=os.path.join(td, "1.bin"), pic_height=H,
                                       pic_width=W, stage_idx=0, code_lt=code_lt, psize=128, skip_decoding=True,
                                       dpb={"mv_feature": None, "ref_mv_y": None}, q_index=3)
            files_ref = {n: open(os.path.join(td, n), "rb").read() for n in sorted(os.listdir(td))}
        for cut in ((("H",), ("L",)) if code_lt else (("H",),), (("H", "L"),) if code_lt else (("H",),)):
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, "1.bin")
                m = net.encode_pair_motion(fr[0], fr[1], {"mv_feature": None, "ref_mv_y": None}, path, stage_idx=0, q_index=3)
                assert_same(m["mv_hat"], ref["mv_hat"], "mv_hat")
                assert m["bit_ME"] == ref["bit_ME"]
                for k in ("mv_feature", "ref_mv_y"):
                    assert_same(m["dpb"][k], ref["dpb"][k], k)
                got = {}
                for chroma in (False, True):
                    for kinds in cut:
                        r = net.encode_pair_part(fr[0][1 if chroma else 0], fr[1][1 if chroma else 0], m["mv_hat"], chroma,
                                                 kinds, code_lt, path, W, H, stage_idx=0, q_index=3)
                        for k in ("H", "L"):
                            if r[k] is not None:
                                got[k + ("c" if chroma else "")] = r[k]
                        for k, b in r["bits"].items():
                            got["bits_" + k + ("c" if chroma else "")] = b
                files = {n: open(os.path.join(td, n), "rb").read() for n in sorted(os.listdir(td))}
            assert files == files_ref
            assert_same(got["H"], ref["H_t"], "H_t")
            assert_same(got["Hc"], ref["H_tc"], "H_tc")
            assert_same(got["L"], ref["L_t"], "L_t")
            assert_same(got["Lc"], ref["L_tc"], "L_tc")
            assert got["bits_H"] + got["bits_Hc"] == ref["bit_H"]
            if code_lt:
                assert got["bits_L"] + got["bits_Lc"] == ref["bit_L"] and got["bits_Lc"] == ref["bit_Lc"]


def test_pair_sharding_two_ranks_real_codec(cuda):
    """BASELINE configs[4] on what one box allows: two fresh processes share the GPU (gloo carries the relay and the
    gather; on the 8-GPU node the same code runs over RCCL), GOP 8 at 128x128 with four ME stages and the REAL codec.
    Both ranks must end with the subband tree, motion fields and bit counts of the single-process schedule, bit for
    bit."""
    import torch.multiprocessing as mp
    import pmctf_gop
    net, _ = product_model(4)
    fr = frames(W, H, 8, device="cuda", seed=11)
    with tempfile.TemporaryDirectory() as td:
        ref = pmctf_gop.encode_gop(net, fr, H, W, 3, td)
        ref2 = pmctf_gop.encode_gop(net, frames(W, H, 8, device="cuda", seed=12), H, W, 3, td)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_pair_shard_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # 4+2+1 pairs: 2+1 whole pairs per rank; the last pair (the one that codes L) is split: motion + luma coders on rank 0,
    # chroma coders on rank 1 (pmctf_dist.pair_parts)
    assert sorted(r[4] for r in res) == [4, 5]
    for rank, bits, bits_mv, fc, _, o1, o2 in res:
        for (b_, bm_, fc_, _), want in (((bits, bits_mv, fc, None), ref), (o1, ref), (o2, ref2)):
            assert b_ == want["bits"] and bm_ == want["bits_mv"], rank
            for a, b in zip(fc_, want["frames_coded"]):
                for x, y in zip(a, b):
                    assert (x is None and y is None) or np.array_equal(x, y.cpu().numpy()), rank
    assert sorted(r[5][3] + r[6][3] for r in res) == [7, 7]     # overlapped: GOP A 4+3, GOP B (started one rank on) 3+4


def _pair_shard_1080p_worker(rank, world, port, q):
    import hashlib
    import torch.distributed as dist
    import pmctf_dist
    import pmctf_gop
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    net, _ = product_model(4)
    w, h, gop = 1920, 1080, 16
    fr = frames(w, h, gop, device="cuda")
    stats = {}
    with tempfile.TemporaryDirectory() as td:
        enc = pmctf_dist.encode_gop_pair_sharded(net, fr, h, w, 3, td, rank, world, dist, stats=stats)
        files = sorted(os.listdir(td))
    rec = pmctf_gop.decode_gop(net, [list(f) for f in enc["frames_coded"]])     # every rank holds the whole subband tree
    ps = [p["yuv"] for p in pmctf_gop.gop_psnr(rec, fr, h, w)]
    tree = hashlib.sha1()
    for fc in enc["frames_coded"]:
        for t in fc:
            if t is not None:
                tree.update(t.contiguous().cpu().numpy().tobytes())
    torch.cuda.synchronize()
    q.put((rank, enc["bits"], enc["bits_mv"], ps, tree.hexdigest(), len(enc["results"]), files, stats))
    dist.barrier()
    dist.destroy_process_group()


def test_pair_sharding_two_ranks_1080p_gop16_vs_reference(cuda):
    """BASELINE configs[4] at its OWN workload (1920x1080, GOP 16, q_index 3, four ME stages) on what one box allows: two
    fresh processes share the GPU, pair k of every temporal stage on rank k mod 2, motion context relayed rank to rank,
    one all-gather of the subband tree per stage (gloo here, RCCL on the 8-GPU node).  Both ranks must end with the bits of
    every frame the REAL reference's CPU run produced, its PSNR within 1e-4 dB after the temporal synthesis of the gathered
    tree, and the same tree; every bitstream file is written by exactly one rank."""
    import torch.multiprocessing as mp
    g = np.load(_digest_path(16, 3))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    procs = [ctx.Process(target=_pair_shard_1080p_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(2)]
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    # 8+4+2+1 pairs: 4+2+1 whole pairs per rank, the last pair split (motion + luma coders / chroma coders)
    assert sorted(r[5] for r in res) == [8, 9]
    assert res[0][4] == res[1][4]                              # the same subband tree on both ranks
    for rank, bits, bits_mv, ps, _, _, files, stats in res:
        assert bits == g["gop.bits"].tolist() and bits_mv == g["gop.bits_mv"].tolist(), rank
        assert np.abs(np.array(ps) - g["gop.psnr_yuv"]).max() < 1e-4
        assert len(stats["gather_bytes_per_stage"]) == 4 and stats["relay_bytes_per_hop"] == 4 * 64 * (288 * 480 + 72 * 120)
    assert sum(r[7]["relay_hops"] for r in res) == 7 + 3 + 1


@pytest.mark.parametrize("K", [2, 4])
def test_cross_gop_batched_equals_gop_by_gop(cuda, K):
    """pmctf_gop.encode_gops_batched (stage s of K closed GOPs in one encode_stage_pairs call, motion context restarted
    at every GOP boundary) against coding the GOPs one after the other with the pair-by-pair schedule: every file, bit
    count and tensor of every GOP identical.  GOP 8, four ME stages, 128x128; the post-processing group size is lowered
    so that the plane-group path is exercised too."""
    import pmctf_gop
    net, _ = product_model(4)
    net.engine().post_process_max_px = 3 * 128 * 128
    gops = [frames(W, H, 8, device="cuda", seed=100 + k) for k in range(K)]
    refs, ref_files = [], []
    for fr in gops:
        with tempfile.TemporaryDirectory() as td:
            refs.append(pmctf_gop.encode_gop(net, fr, H, W, 3, td))
            ref_files.append({n: open(os.path.join(td, n), "rb").read() for n in sorted(os.listdir(td))})
    tds = [tempfile.mkdtemp() for _ in range(K)]
    outs = pmctf_gop.encode_gops_batched(net, gops, H, W, 3, tds)
    for k in range(K):
        files = {n: open(os.path.join(tds[k], n), "rb").read() for n in sorted(os.listdir(tds[k]))}
        assert files == ref_files[k], f"GOP {k}: files differ"
        assert outs[k]["bits"] == refs[k]["bits"] and outs[k]["bits_mv"] == refs[k]["bits_mv"], k
        for a, b in zip(outs[k]["frames_coded"], refs[k]["frames_coded"]):
            for x, y in zip(a, b):
                assert (x is None and y is None) or torch.equal(x, y), k


@pytest.mark.parametrize("shape", [(1, 37, 53), (2, 8, 32), (1, 2, 5), (3, 19, 70), (1, 130, 33)])
def test_fused_predict_update_equals_separate_launches(setup, shape):
    """pu_fused.hip (the whole PredictUpdate CNN + the lifting arithmetic in one launch) against the chain of separate
    launches it replaces, on plane sizes that do not divide into its 8x32 tiles, batches, and planes smaller than one
    tile: the temporal predict / update filters and all four iWave lifting steps, forward and backward signs —
    identical bits.  (Both paths are pinned to the oracle by test_mctf_bitexact / test_dwt_postprocess_bitexact.)"""
    net, orc = setup
    eng = net.engine()
    n, h, w = shape
    rng = np.random.default_rng(h * 1000 + w)
    x = torch.from_numpy(rng.uniform(0, 255, (n, 1, h, w)).astype(np.float32)).cuda()
    o = torch.from_numpy(rng.normal(0, 40, (n, 1, h, w)).astype(np.float32)).cuda()
    wt = "hp_coder.wavelet_transform.lift_h"
    got = {}
    keep_max = eng.pu_fused_max_px
    eng.pu_fused_max_px = 1 << 40
    for fused in (True, False):
        eng.pu_fused = fused
        r = {"P": eng.predict_filter(0, x), "U": eng.update_filter(0, o)}
        for cn, pn in (("conv_P1", "P_1"), ("conv_U1", "U_1"), ("conv_P2", "P_2"), ("conv_U2", "U_2")):
            r[pn + "+"] = eng.lift_step(wt, cn, pn, x, o, 1.0)
            r[pn + "-"] = eng.lift_step(wt, cn, pn, o, x, -1.0)
        got[fused] = r
    eng.pu_fused, eng.pu_fused_max_px = True, keep_max
    for k in got[True]:
        assert_same(got[True][k], got[False][k], f"{shape} {k}")
    # and against the oracle directly
    assert_same(got[True]["P"], orc.predict_filter(0, x.cpu()), "predict_filter vs oracle")
    assert_same(got[True]["U"], orc.update_filter(0, o.cpu()), "update_filter vs oracle")
    if h >= 2:
        assert_same(got[True]["P_1+"], o.cpu() + orc.lift_branch(wt, "conv_P1", "P_1", x.cpu()), "lift step vs oracle")


class _OracleCA:
    """the oracle behind the model API the content-adaptive driver calls (encode_one_stage with or without a bitstream)"""

    def __init__(self, orc):
        self.orc = orc
        self.num_me_stages = orc.num_me_stages

    @staticmethod
    def get_qp_num():
        return 21

    def inverse_MCTF(self, *a, **k):
        return self.orc.inverse_MCTF(*a, **k)

    def encode_one_stage(self, ref_frame, cur_frame, code_lt, dpb, output_path=None, pic_width=None, pic_height=None,
                         psize=128, skip_decoding=True, stage_idx=0, q_index=0, me_downsample=1):
        if output_path is None:
            return self.orc.estimate_one_stage(ref_frame, cur_frame, code_lt, dpb, stage_idx, q_index, me_downsample)
        return self.orc.encode_one_stage(ref_frame, cur_frame, code_lt, dpb, output_path=output_path,
                                         pic_width=pic_width, pic_height=pic_height, psize=psize,
                                         skip_decoding=skip_decoding, stage_idx=stage_idx, q_index=q_index,
                                         me_downsample=me_downsample)


@pytest.mark.parametrize("write_stream", [True, False])
def test_content_adaptive_driver_matches_oracle(setup, write_stream):
    """the GOP-size x motion-resolution RD search of test_pMCTF_CA.py:341-414 (pmctf_ca.search_gop) over
    the HIP product and over the oracle: the same options tried in the same order, the same costs (write mode: bits are
    file sizes, identical; estimate mode — encode_one_stage(output_path=None), the branch that is broken upstream —
    within 1e-6 relative), the same choice.  8 frames of 128x128, GOP sizes {8, 4}, motion at full and half resolution."""
    import pmctf_ca
    net, orc = setup
    fr = frames(W, H, 8, seed=31)
    frd = [[y.cuda(), c.cuda()] for y, c in fr]
    with tempfile.TemporaryDirectory() as td, tempfile.TemporaryDirectory() as td2:
        a = pmctf_ca.search_gop(net, frd, H, W, 3, td, write_stream=write_stream, ds_factors=(1, 2))
        b = pmctf_ca.search_gop(_OracleCA(orc), fr, H, W, 3, td2, write_stream=write_stream, ds_factors=(1, 2))
    assert [(s, d) for s, d, _ in a["trials"]] == [(s, d) for s, d, _ in b["trials"]]
    assert (a["gop_choice"], a["ds_choice"], a["tested_opts"]) == (b["gop_choice"], b["ds_choice"], b["tested_opts"])
    assert a["gop_choice"] in (8, 4) and a["ds_choice"] in (1, 2) and 2 <= a["tested_opts"] <= 3
    for (_, _, ra), (_, _, rb) in zip(a["trials"], b["trials"]):
        assert abs(ra - rb) <= 1e-6 * abs(rb)
    if write_stream:
        assert a["logs"]["bits"] == b["logs"]["bits"]
        # reconstructions are bit-identical; the harness-side PSNR is a torch mean evaluated on the device the frames
        # live on, and the GPU / CPU reductions sum in different orders (1e-6 dB)
        assert np.abs(np.array(a["logs"]["psnrs"]) - np.array(b["logs"]["psnrs"])).max() < 1e-5
    else:
        assert np.allclose(a["logs"]["bits"], b["logs"]["bits"], rtol=1e-6)
    assert len(a["logs"]["bits"]) == 8 and a["logs"]["frame_types"].count(0) == 8 // a["gop_choice"]


def test_content_adaptive_search_vs_reference_script(cuda):
    """pmctf_ca.search_gop over the HIP product against what the REAL script produced (tests/golden/reference_ca_*.npz,
    written by tools/make_golden.py --ca from the unmodified test_pMCTF_CA.run_test in write mode): the same options in
    the same order (GOP 8 / 4 at full-resolution motion, then GOP 8 with motion at 1/2 and 1/4 resolution — padded to
    128, 128 and 256), the same bits per frame in every trial, RD costs within 1e-6 relative, the same choices."""
    import pmctf_ca
    g = golden("reference_ca_128x128_gop8_q3.npz")
    w, h, G, q, me, seed = (int(v) for v in g["ca.meta"])
    net, _ = product_model(me)
    fr8 = frames(w, h, G, seed=seed)
    frd = [[y[:, :, :h, :w].cuda(), c[:, :, :h // 2, :w // 2].cuda()] for y, c in fr8]
    with tempfile.TemporaryDirectory() as td:
        r = pmctf_ca.search_gop(net, frd, h, w, q, td, write_stream=True)
    assert [(s_, d) for s_, d, _ in r["trials"]] == [tuple(t) for t in g["ca.trials"].tolist()]
    assert (r["gop_choice"], r["ds_choice"], r["tested_opts"]) == (int(g["ca.gop_choice"][0]), int(g["ca.ds_choice"][0]),
                                                                  int(g["ca.tested_opts"][0]))
    for (_, _, rd), ref in zip(r["trials"], g["ca.trial_rd"].tolist()):
        assert abs(rd - ref) <= 1e-6 * abs(ref), (rd, ref)
    assert np.allclose(np.array(r["logs"]["bpps"]), g["ca.frame_bpp"], rtol=0, atol=0)
    assert np.abs(np.array(r["logs"]["psnrs"]) - g["ca.frame_psnr"]).max() < 1e-4


def test_estimate_only_branch_of_encode_one_stage(setup):
    """encode_one_stage(output_path=None) (pMCTF_L.py:530-551): luma + chroma forward_one_stage, the dictionary of the
    write branch with estimated bits; tensors bit-exact against the oracle's restatement, bits to 1e-6, and the motion
    context usable by a following pair."""
    net, orc = setup
    fr = frames(W, H, 4, seed=23)
    frd = [[y.cuda(), c.cuda()] for y, c in fr]
    dpb = {"mv_feature": None, "ref_mv_y": None}
    r = net.encode_one_stage(frd[0], frd[1], True, dpb, output_path=None, pic_width=W, pic_height=H, stage_idx=0, q_index=5)
    o = orc.estimate_one_stage(fr[0], fr[1], True, dpb, 0, 5)
    for k in ("L_t", "H_t", "L_tc", "H_tc", "mv_hat"):
        assert_same(r[k], o[k], k)
    for k in ("bit_L", "bit_H", "bit_Lc", "bit_Hc", "bit_ME"):
        assert abs(float(r[k]) - float(o[k])) <= 1e-6 * abs(float(o[k])), k
    r2 = net.encode_one_stage(frd[2], frd[3], False, r["dpb"], output_path=None, pic_width=W, pic_height=H, q_index=5)
    o2 = orc.estimate_one_stage(fr[2], fr[3], False, o["dpb"], 0, 5)
    assert r2["bit_L"] is None and r2["bit_Lc"] is None
    assert_same(r2["H_t"], o2["H_t"], "chained pair H_t")
    assert abs(float(r2["bit_ME"]) - float(o2["bit_ME"])) <= 1e-6 * abs(float(o2["bit_ME"]))


def test_deferred_stage_batching_behind_the_drop_in_api(cuda):
    """The product's OPT-IN mode (lazy_stages / PMCTF_LAZY=1): encode_one_stage hands back deferred results, the pairs a
    caller passes one by one are collected per temporal stage and coded as ONE batch when a value is first needed
    (pMCTF.hip.deferred).  A loop that only STORES the results (pmctf_gop.encode_gop(store_only=True) — the harness loop
    without its two prints) must give exactly the eager results — files, bit counts, subband tree, reconstruction —
    with one encode_stage_pairs call per stage; the harness's own loop, whose prints look at every pair's bit count
    (test_pMCTF_flex.py:240,248), codes pair by pair; and looking at a result early codes what has been collected."""
    import pmctf_gop
    net, _ = product_model(4, lazy=False)
    fr = frames(W, H, 8, device="cuda", seed=41)
    with tempfile.TemporaryDirectory() as td:
        ref = pmctf_gop.encode_gop(net, fr, H, W, 3, td)
        ref_files = {n: open(os.path.join(td, n), "rb").read() for n in sorted(os.listdir(td))}
        ref_rec = pmctf_gop.decode_gop(net, [list(f) for f in ref["frames_coded"]])
    net.lazy_stages = True
    calls = []
    orig = net.encode_stage_pairs
    net.encode_stage_pairs = lambda pairs, *a, **k: (calls.append(len(pairs)), orig(pairs, *a, **k))[1]
    with tempfile.TemporaryDirectory() as td:
        loud = pmctf_gop.encode_gop(net, fr, H, W, 3, td)     # the harness's loop with its prints: every pair on its own
        assert calls == [1] * 7 and loud["bits"] == ref["bits"], calls
        del calls[:]
    with tempfile.TemporaryDirectory() as td:
        lazy = pmctf_gop.encode_gop(net, fr, H, W, 3, td, store_only=True)
        assert calls == [4, 2, 1], calls                     # one batch per temporal stage
        files = {n: open(os.path.join(td, n), "rb").read() for n in sorted(os.listdir(td))}
        rec = pmctf_gop.decode_gop(net, [list(f) for f in lazy["frames_coded"]])
    assert files == ref_files and lazy["bits"] == ref["bits"] and lazy["bits_mv"] == ref["bits_mv"]
    for a, b in zip(lazy["frames_coded"], ref["frames_coded"]):
        for x, y in zip(a, b):
            assert (x is None and y is None) or torch.equal(x, y)
    for a, b in zip(rec, ref_rec):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    # a caller that looks at every result right away gets the pair-by-pair schedule, and its files exist at that moment
    del calls[:]
    dpb = {"mv_feature": None, "ref_mv_y": None}
    with tempfile.TemporaryDirectory() as td:
        r1 = net.encode_one_stage(fr[0], fr[1], False, dpb, output_path=os.path.join(td, "1.bin"), pic_width=W,
                                  pic_height=H, skip_decoding=True, stage_idx=0, q_index=3)
        assert os.listdir(td) == [] and calls == []          # nothing has been needed yet
        bits = r1["bit_H"] + r1["bit_ME"]                     # arithmetic keeps deferring (test_pMCTF_flex.py:236)
        assert calls == []
        r2 = net.encode_one_stage(fr[2], fr[3], False, r1["dpb"], output_path=os.path.join(td, "3.bin"), pic_width=W,
                                  pic_height=H, skip_decoding=True, stage_idx=0, q_index=3)
        assert float(bits) == ref["bits"][1] and calls == [2]           # first use: both collected pairs are coded
        assert sorted(os.listdir(td)) == ["1.bin", "1_C_main.bin", "1_mv.bin", "3.bin", "3_C_main.bin", "3_mv.bin"]
        assert float(r2["bit_H"] + r2["bit_ME"]) == ref["bits"][3]
        assert torch.equal(torch.round(r2["H_t"]), torch.round(ref["frames_coded"][3][0]))     # torch functions force
    # motion at reduced resolution (the content-adaptive harness's calls) is deferred and batched the same way
    net.lazy_stages = False
    with tempfile.TemporaryDirectory() as td:
        ref2 = pmctf_gop.encode_gop(net, fr[:4], H, W, 3, td, me_downsample=2)
        ref2_files = {n: open(os.path.join(td, n), "rb").read() for n in sorted(os.listdir(td))}
    net.lazy_stages = True
    del calls[:]
    with tempfile.TemporaryDirectory() as td:
        lazy2 = pmctf_gop.encode_gop(net, fr[:4], H, W, 3, td, me_downsample=2, store_only=True)
        assert calls == [2, 1]
        assert {n: open(os.path.join(td, n), "rb").read() for n in sorted(os.listdir(td))} == ref2_files
    assert lazy2["bits"] == ref2["bits"]
    net.encode_stage_pairs = orig


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_reduced_precision_profile_is_self_consistent(cuda, precision):
    """The AUXILIARY arithmetic profiles (HipEngine(precision=...): the dense 3x3 convolutions on bf16 MFMA with split
    operands) earn no parity claim, but they must be usable: deterministic, decodable by the same build bit for bit
    (decoder in the loop == encoder's reconstruction), close to the exact profile (bits within 1 % / 10 %, PSNR within
    0.01 / 1 dB), and the exact profile must be untouched by their existence."""
    import pmctf_gop
    from pMCTF.hip import ops
    w, h = 448, 256
    fr = frames(w, h, 4, device="cuda", seed=77)
    exact_net, _ = product_model(1)
    with tempfile.TemporaryDirectory() as td:
        exact = pmctf_gop.encode_gop(exact_net, fr, h, w, 3, td)
        exact_ps = pmctf_gop.gop_psnr(pmctf_gop.decode_gop(exact_net, [list(f) for f in exact["frames_coded"]]), fr, h, w)
    net, _ = product_model(1)
    net.precision = precision
    old = ops.SPLIT_MIN_PX
    ops.SPLIT_MIN_PX = 4096          # small test planes: let the level-0/1 subbands take the split kernel
    try:
        assert net.engine().precision == precision and net.engine().nsplit == {"bf16x3": 3, "bf16": 1}[precision]
        with tempfile.TemporaryDirectory() as td:
            a = pmctf_gop.encode_gop(net, fr, h, w, 3, td)
            files_a = {n: open(os.path.join(td, n), "rb").read() for n in sorted(os.listdir(td))}
            b = pmctf_gop.encode_gop(net, fr, h, w, 3, td)
            files_b = {n: open(os.path.join(td, n), "rb").read() for n in sorted(os.listdir(td))}
            assert files_a == files_b, "reduced-precision encode is not deterministic"
            ps = pmctf_gop.gop_psnr(pmctf_gop.decode_gop(net, [list(f) for f in a["frames_coded"]]), fr, h, w)
            # the real decoder of the same build reproduces the encoder's reconstruction from the files
            dpb = {"mv_feature": None, "ref_mv_y": None}
            e = net.encode_one_stage(fr[0], fr[1], True, dpb, output_path=os.path.join(td, "1.bin"), pic_width=w,
                                     pic_height=h, skip_decoding=True, stage_idx=0, q_index=3)
            d = net.encode_one_stage(fr[0], fr[1], True, dpb, output_path=os.path.join(td, "1.bin"), pic_width=w,
                                     pic_height=h, skip_decoding=False, stage_idx=0, q_index=3)
            for k in ("L_t", "H_t", "L_tc", "H_tc", "mv_hat"):
                assert_same(d[k], e[k], f"{precision}: decoded {k} vs encoder reconstruction")
    finally:
        ops.SPLIT_MIN_PX = old
    rel_bits = abs(sum(a["bits"]) - sum(exact["bits"])) / sum(exact["bits"])
    dpsnr = max(abs(p["yuv"] - q["yuv"]) for p, q in zip(ps, exact_ps))
    print(f"{precision}: total bits {sum(a['bits']):.0f} vs exact {sum(exact['bits']):.0f} ({rel_bits:.2e}), max |dPSNR| {dpsnr:.2e} dB")
    assert rel_bits < (0.01 if precision == "bf16x3" else 0.10) and dpsnr < (0.01 if precision == "bf16x3" else 1.0)
