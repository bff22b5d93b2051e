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


def test_full_size_1080p_properties(cuda):
    """At the benchmark's full size (1920x1080, 4 ME stages) the oracle is too slow, so check what must hold at any
    size: (1) the decoder reproduces the encoder's reconstruction bit for bit from the written files, (2) the files
    account for the reported bits, (3) the lifting is inverted by inverse_MCTF up to rounding, (4) coding is
    deterministic (same bytes twice)."""
    import os
    net, _ = product_model(4)
    w, h = 1920, 1080
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
    assert e["H_t"].shape == (1, 1, 1152, 1920) and e["H_tc"].shape == (2, 1, 576, 960)
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
        for name, data in cases.items():
            if name == "truncated":      # keep the header's length field consistent with what is there
                data = data[:12] + struct.pack(">I", len(data) - 16) + data[16:]
            open(fn, "wb").write(data)
            try:
                out = coder.decompress(fn, padding=64, q_index=3)["x_hat"]
                torch.cuda.synchronize()
                assert out.shape == (1, 1, 128, 128), name
            except (ValueError, RuntimeError):
                pass
        open(fn, "wb").write(good)       # and the coder still works afterwards
        again = coder.decompress(fn, padding=64, q_index=3)["x_hat"]
        assert torch.isfinite(again).all()


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
    assert same >= 10 and diff <= 1, (same, diff)


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
    assert same >= 9 and diff <= 2, (same, diff)


def _headline_fixtures():
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return [q for q in (3, 20, 0)
            if os.path.exists(os.path.join(d, "reference_1920x1080_gop16_me4%s_digest.npz" % ("" if q == 3 else f"_q{q}")))]


@pytest.mark.parametrize("q_index", _headline_fixtures())
def test_headline_config_1080p_gop16_vs_reference(cuda, q_index):
    """BASELINE's headline configuration itself — 1920x1080, GOP 16, q_index 3, four ME stages, full encode with
    bitstream write and PSNR — against digests of what the REAL reference produced for the same synthetic sequence and
    weights on the CPU (tools/make_golden.py --width 1920 --height 1080 --gop_only --gop 16 --me_stages 4, about half an
    hour of CPU per rate point; q_index 3 is the benchmark's point, 20 and 0 the ends of the RD sweep of configs[3]):
    bits of every frame identical (bpp bit-exact), PSNR within 1e-4 dB."""
    import hashlib
    import pmctf_gop
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                        "reference_1920x1080_gop16_me4%s_digest.npz" % ("" if q_index == 3 else f"_q{q_index}"))
    g = np.load(path)
    net, _ = product_model(4)
    net.engine().keep_streams = True
    w, h = 1920, 1080
    fr = frames(w, h, 16, device="cuda")
    with tempfile.TemporaryDirectory() as td:
        enc = pmctf_gop.encode_gop(net, fr, h, w, q_index, td)
        rec = pmctf_gop.decode_gop(net, enc["frames_coded"])
        ps = pmctf_gop.gop_psnr(rec, fr, h, w)
    bpp = sum(enc["bits"]) / (16 * w * h)
    bpp_ref = float(g["gop.bits"].sum()) / (16 * w * h)
    psnr_err = np.abs(np.array([p["yuv"] for p in ps]) - g["gop.psnr_yuv"]).max()
    same = diff = 0
    for i, r in enumerate(enc["results"]):
        cur = int(g[f"gop.pair{i}.meta"][2])
        for name, key in (("mv", f"{cur}_mv.bin"), ("H", f"{cur}.bin"), ("Hc", f"{cur}_C_main.bin"),
                          ("L", "0_main.bin"), ("Lc", "0_C_main.bin")):
            k = f"gop.pair{i}.filesha1.{key}"
            if name in r["files"] and k in g.files:
                if hashlib.sha1(r["files"][name]).digest() == g[k].tobytes():
                    same += 1
                else:
                    diff += 1
    print(f"1080p GOP-16 q_index {q_index}: bpp {bpp:.6f} (reference {bpp_ref:.6f}), max PSNR error {psnr_err:.2e} dB, "
          f"{same} of {same + diff} files byte-identical")
    if q_index == 3:
        # the benchmark's rate point: the north star's bar, exactly
        assert enc["bits"] == g["gop.bits"].tolist(), "bits per frame differ from the reference"
        assert enc["bits_mv"] == g["gop.bits_mv"].tolist()
        assert psnr_err < 1e-4
        # every file has the reference's length; where PM-F32 and ATen round a conv sum differently and a tie flips a
        # symbol, the bytes inside differ (13 of 47 files on this sequence) — reported above, and bounded here
        assert same + diff == 47 and same >= 30
    else:
        # ends of the RD sweep: the same effect can move a stream by one 32-bit rANS word (observed: q_index 20, two frames
        # of 16 are 32 bits longer out of 98.5 Mbit, PSNR off by 1.3e-4 dB).  Stated as measured, bounded here.
        dbits = np.abs(np.array(enc["bits"]) - g["gop.bits"])
        print("   per-frame bit differences:", dbits.tolist())
        assert dbits.max() <= 64 and dbits.sum() <= 1e-6 * g["gop.bits"].sum()
        assert psnr_err < 5e-4
        assert same + diff == 47


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


def test_headline_config_stage_batched_vs_reference(cuda):
    """The schedule bench.py measures by default (pairs of a temporal stage as one batch) on the headline configuration,
    against the real reference's digests: bits of every frame identical, PSNR within 1e-4 dB, and the very same files
    as the pair-by-pair schedule."""
    import hashlib
    import pmctf_gop
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                             "reference_1920x1080_gop16_me4_digest.npz"))
    net, _ = product_model(4)
    net.engine().keep_streams = True
    w, h = 1920, 1080
    fr = frames(w, h, 16, device="cuda")
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
    assert batched == paired and len(batched) == 47


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
