"""The oracle restatement against fixtures generated from the REAL reference (tools/make_golden.py):
ATen back-end within fp noise of the fixtures (bit-exact on the generating machine), PM-F32 back-end within the
stated tolerances and with identical bitstream sizes."""
import tempfile

import numpy as np
import pytest
import torch

from helpers import frames, golden, synth_sd_cpu

W = H = 128


@pytest.fixture(scope="module")
def sd():
    return synth_sd_cpu(1)


def _close(a, b, tol):
    a = a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return np.abs(a.astype(np.float64) - b.astype(np.float64)).max() <= tol


@pytest.mark.parametrize("backend,tol", [("torch", 1e-4), ("cdef", 2e-3)])
def test_units_and_pwave_stream(sd, backend, tol):
    import pmctf_synth
    from pmctf_oracle.model import Oracle, get_curr_q
    g = golden()
    o = Oracle(sd, 1, backend)
    (Y0, C0), (Y1, C1) = frames(W, H, 2)
    flow = torch.from_numpy(pmctf_synth.hashed_normal("golden.flow", (1, 2, H, W), 3.0))
    assert np.array_equal(flow.numpy(), g["unit.flow"])
    with torch.no_grad():
        assert _close(o.K.flow_warp(Y0, flow), g["unit.warp"], tol)
        assert _close(o.predict_filter(0, Y0), g["unit.predict_filter"], tol)
        L, Ht, _, _ = o.forward_MCTF(Y0, Y1, flow)
        assert _close(L, g["unit.mctf.L"], tol) and _close(Ht, g["unit.mctf.H"], tol)
        r, c = o.inverse_MCTF(L, Ht, flow)
        assert _close(r, g["unit.imctf.ref"], tol) and _close(c, g["unit.imctf.cur"], tol)
        assert _close(o.spynet(Y1.tile((1, 3, 1, 1)) / 255, Y0.tile((1, 3, 1, 1)) / 255), g["unit.spynet"], 1e-5)
        Hg = torch.from_numpy(g["unit.mctf.H"])
        sb = o.forward_lift_2d("hp_coder", Hg)
        for k in ("ll", "lh", "hl", "hh"):
            assert _close(sb[k].contiguous(), g[f"unit.dwt.{k}"], tol)
        x_hat, data, trace = o.pwave_compress("hp_coder", Hg, [1, 1, H, W], 3, get_curr_q(o.sd["hp_q_scale.0"], 3))
        sym = np.concatenate([t[0] for t in trace])
        idx = np.concatenate([t[1] for t in trace])
        assert [t[0].size for t in trace] == g["pwave.push_sizes"].tolist()      # 1 LL push + 12 subbands x 4 steps
        assert np.array_equal(sym, g["pwave.symbols"]) and np.array_equal(idx, g["pwave.indexes"])
        assert data == g["pwave.file"].tobytes()
        assert _close(x_hat, g["pwave.x_hat"], tol)


def test_gop4_bits_and_psnr_torch_backend(sd):
    """Full GOP-4 through the harness loop: per-frame bits identical to the reference's file sizes, PSNR within 1e-4 dB."""
    import pmctf_gop
    from pmctf_oracle.model import Oracle
    g = golden()
    o = Oracle(sd, 1, "torch")
    fr = frames(W, H, 4)
    with tempfile.TemporaryDirectory() as td, torch.no_grad():
        enc = pmctf_gop.encode_gop(o, fr, H, W, 3, td)
        rec = pmctf_gop.decode_gop(o, enc["frames_coded"])
        ps = pmctf_gop.gop_psnr(rec, fr, H, W)
    assert enc["bits"] == g["gop.bits"].tolist()
    assert enc["bits_mv"] == g["gop.bits_mv"].tolist()
    assert np.abs(np.array([p["yuv"] for p in ps]) - g["gop.psnr_yuv"]).max() < 1e-4
    for i, r in enumerate(enc["results"]):
        assert _close(r["mv_hat"], g[f"gop.pair{i}.mv_hat"], 1e-5)
        assert _close(r["H_t"], g[f"gop.pair{i}.H_t"], 1e-3)
        cur = int(g[f"gop.pair{i}.meta"][2])
        for name, key in (("mv", f"{cur}_mv.bin"), ("H", f"{cur}.bin"), ("Hc", f"{cur}_C_main.bin")):
            assert r["files"][name] == g[f"gop.pair{i}.file.{key}"].tobytes()


def test_sequence_driver_reproduces_the_reference_run(sd):
    """pmctf_gop.encode_sequence (the build's own driver of a whole sequence: .yuv reader, padding, one encode_one_stage
    call per pair with both report lines, synthesis, PSNR, log record) with the oracle as the codec reproduces what the
    REAL script's loop produced for the golden GOP-4: per-frame bits, motion bits and PSNR — so the GPU test that drives
    the product through the same function compares like with like.  The contract the script relies on is asserted on
    the way: per-frame tables of plain numbers, one L picture per GOP, the record's keys, JSON text."""
    import json
    import os
    import pmctf_gop
    import pmctf_synth
    from pmctf_oracle.model import Oracle
    g = golden()
    o = Oracle(sd, 1, "torch")
    with tempfile.TemporaryDirectory() as td:
        yuv = os.path.join(td, "seq.yuv")
        pmctf_gop.write_yuv(yuv, pmctf_synth.synth_yuv420(W, H, 4))
        assert os.path.getsize(yuv) == 4 * W * H * 3 // 2
        bins = os.path.join(td, "bin")
        os.makedirs(bins)
        out = pmctf_gop.encode_sequence(o, yuv, W, H, 4, 4, 3, bins, "cpu")
        assert sorted(os.listdir(bins)) == sorted(["0_main.bin", "0_C_main.bin"] + [f"{k}{s}" for k in (1, 2, 3)
                                                  for s in (".bin", "_mv.bin", "_C_main.bin")])
    assert out["bits"] == g["gop.bits"].tolist()
    assert np.allclose(np.array(out["bpp_mv"]) * (W * H), g["gop.bits_mv"], rtol=0, atol=1e-9)
    assert np.abs(np.array(out["psnr"]) - g["gop.psnr_yuv"]).max() < 1e-4
    assert out["frame_types"] == [0, 1, 1, 1]
    assert all(type(v) is float for k in ("bits", "bpp_mv", "psnr", "psnr_rgb") for v in out[k])
    parsed = json.loads(out["json"])
    assert {"frame_pixel_num", "i_frame_num", "p_frame_num", "ave_i_frame_bpp", "ave_p_frame_bpp", "ave_all_frame_bpp",
            "ave_all_frame_psnr", "frame_bpp", "frame_psnr", "frame_type", "test_time"} <= set(parsed)
    assert parsed["i_frame_num"] == 1 and parsed["p_frame_num"] == 3
    assert abs(parsed["ave_all_frame_bpp"] - float(g["gop.bits"].sum()) / (4 * W * H)) < 1e-6
    assert sum(l.startswith("percentage MV") for l in out["lines"]) == 3
    assert sum(l.startswith("Frame ") and l.endswith(" bpp") for l in out["lines"]) == 3


def test_content_adaptive_search_reproduces_the_reference_script():
    """The build's own driver of the content-adaptive RD search (pmctf_ca.search_gop, a restatement of
    test_pMCTF_CA.py:341-414) over the oracle against what the REAL script's run_test did on the same 8 frames
    (tools/make_golden.py --ca, write mode): every trial in order with its per-frame bits and RD cost, the number of
    options tested, the chosen GOP size and motion resolution, the per-frame log of the chosen option."""
    import pmctf_ca
    from pmctf_oracle.model import Oracle
    g = golden("reference_ca_128x128_gop8_q3.npz")
    w, h, G, q, me, seed = (int(v) for v in g["ca.meta"])
    o = Oracle(synth_sd_cpu(me), me, "torch")
    o.get_qp_num = lambda: 21
    fr = [[y[:, :, :h, :w], c[:, :, :h // 2, :w // 2]] for y, c in frames(w, h, G, seed=seed)]
    seen = []
    with tempfile.TemporaryDirectory() as td, torch.no_grad():
        r = pmctf_ca.search_gop(o, fr, h, w, q, td, write_stream=True,
                                on_trial=lambda size, ds, logs: seen.append((size, ds, list(logs["bits"]), list(logs["psnrs"]))))
    assert [(s_, d) for s_, d, _ in r["trials"]] == [tuple(t) for t in g["ca.trials"].tolist()]
    assert r["tested_opts"] == int(g["ca.tested_opts"][0])
    assert (r["gop_choice"], r["ds_choice"]) == (int(g["ca.gop_choice"][0]), int(g["ca.ds_choice"][0]))
    for i, (_, _, bits, psnrs) in enumerate(seen):
        assert bits == g["ca.trial_bits"][i].tolist(), i
        assert np.abs(np.array(psnrs) - g["ca.trial_psnr_yuv"][i]).max() < 1e-4
    assert np.allclose([t[2] for t in r["trials"]], g["ca.trial_rd"], rtol=1e-9)
    assert np.array_equal(np.array(r["logs"]["bpps"]), g["ca.frame_bpp"])
    assert abs(pmctf_ca.get_cur_lamda(q) - float(g["ca.lamda"][0])) < 1e-15


@pytest.mark.parametrize("backend,rtol,ttol", [("torch", 1e-6, 1e-4), ("cdef", 2e-5, 2e-3)])
def test_estimate_mode_forward(sd, backend, rtol, ttol):
    """forward_one_stage (bit ESTIMATES instead of range coding, pMCTF_L.py:332-379) against the real reference's
    outputs: luma pair with motion estimation and L coding, chroma pair driven by the luma motion, and a pair without
    L coding whose MV codec uses the dpb of a previous pair.  Scalars relative, tensors absolute."""
    from pmctf_oracle.model import Oracle
    g = golden()
    o = Oracle(sd, 1, backend)
    (Y0, C0), (Y1, C1) = frames(W, H, 2)
    with torch.no_grad():
        dpb = {"mv_feature": None, "ref_mv_y": None}
        ry = o.forward_one_stage(Y0, Y1, 3, True, dpb)
        rc = o.forward_one_stage(C0, C1, 3, True, dpb, mv_hat=ry["mv_hat"])
        rn = o.forward_one_stage(Y0, Y1, 12, False, ry["dpb"], stage_idx=0)
    checked = 0
    for tag, d in (("y", ry), ("c", rc), ("n", rn)):
        for k, v in d.items():
            if k == "dpb":
                for kk, vv in v.items():
                    if vv is not None:
                        assert _close(vv, g[f"est.{tag}.dpb.{kk}"], ttol), (tag, kk)
                continue
            key = f"est.{tag}.{k}"
            if v is None:
                assert key not in g.files, key
            elif isinstance(v, torch.Tensor):
                assert _close(v, g[key], ttol), key
            else:
                ref = float(g[key])
                assert abs(v - ref) <= rtol * max(1.0, abs(ref)), (key, v, ref)
                checked += 1
    assert checked == sum(1 for k in g.files if k.startswith("est.") and g[k].shape == ()) == 34


@pytest.mark.parametrize("backend,ttol", [("torch", 1e-4), ("cdef", 2e-3)])
def test_half_resolution_motion(sd, backend, ttol):
    """me_downsample=2 (pMCTF_L.py:456-458,475-476,516-517): encode, standalone decompress_mv and the estimate-mode
    forward against the real reference's files / tensors."""
    from pmctf_oracle.model import Oracle, decode_p_bytes
    g = golden()
    o = Oracle(sd, 1, backend)
    fr = frames(W, H, 2)
    dpb = {"mv_feature": None, "ref_mv_y": None}
    with torch.no_grad():
        r = o.encode_one_stage(fr[0], fr[1], False, dpb, pic_width=W, pic_height=H, q_index=3, me_downsample=2)
        names = {"mv": "1_mv.bin", "H": "1.bin", "Hc": "1_C_main.bin"}
        for k, n in names.items():
            assert r["files"][k] == g[f"ds2.file.{n}"].tobytes(), n
        for k in ("H_t", "H_tc", "mv_hat"):
            assert _close(r[k], g[f"ds2.{k}"], ttol), k
        _, string = decode_p_bytes(r["files"]["mv"])
        d = o.decompress_mv(string, H // 2, W // 2, dpb, 0, 3, me_downsample=2)
        assert _close(d["mv_hat"], g["ds2.dec.mv_hat"], ttol) and _close(d["mv_feature"], g["ds2.dec.mv_feature"], ttol)
        e = o.forward_one_stage(fr[0][0], fr[1][0], 3, False, dpb, me_downsample=2)
        for k in ("bpp_mv_y", "bpp_mv_z", "bpp", "bit_H", "me_mse"):
            ref = float(g[f"ds2.est.{k}"])
            assert abs(e[k] - ref) <= 2e-5 * max(1.0, abs(ref)), (k, e[k], ref)
        assert _close(e["mv_hat"], g["ds2.est.mv_hat"], ttol) and _close(e["H_t"], g["ds2.est.H_t"], ttol)


def test_gop4_448x256_torch_backend(sd):
    """A second size against the real reference (padded 512x256 planes, GOP-4 through the harness loop): per-frame
    bits, every bitstream file and PSNR identical."""
    import pmctf_gop
    from helpers import golden_448
    from pmctf_oracle.model import Oracle
    g = golden_448()
    o = Oracle(sd, 1, "torch")
    w, h = 448, 256
    fr = frames(w, h, 4)
    with tempfile.TemporaryDirectory() as td, torch.no_grad():
        enc = pmctf_gop.encode_gop(o, fr, h, w, 3, td)
        rec = pmctf_gop.decode_gop(o, enc["frames_coded"])
        ps = pmctf_gop.gop_psnr(rec, fr, h, w)
    assert enc["bits"] == g["gop.bits"].tolist() and enc["bits_mv"] == g["gop.bits_mv"].tolist()
    assert np.abs(np.array([p["yuv"] for p in ps]) - g["gop.psnr_yuv"]).max() < 1e-4
    n = 0
    for i, r in enumerate(enc["results"]):
        cur = int(g[f"gop.pair{i}.meta"][2])
        for name, key in (("mv", f"{cur}_mv.bin"), ("H", f"{cur}.bin"), ("Hc", f"{cur}_C_main.bin"),
                          ("L", "0_main.bin"), ("Lc", "0_C_main.bin")):
            if name in r["files"] and f"gop.pair{i}.file.{key}" in g.files:
                assert r["files"][name] == g[f"gop.pair{i}.file.{key}"].tobytes(), (i, name)
                n += 1
    assert n >= 11
