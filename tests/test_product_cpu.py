"""CPU-side checks of the product: parameter tree, C-ABI exports, host range coder vs the oracle, harness
support code, multi-process scheduling (gloo).  No kernel is launched here."""
import ctypes
import glob
import io
import json
import os
import re
import struct
import tempfile

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n", [1, 2])
def test_state_dict_layout_matches_reference(n):
    from pMCTF.models.video.pMCTF_L import pMCTF
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", f"state_dict_keys_me{n}.json")))
    mine = {k: list(v.shape) for k, v in pMCTF(num_me_stages=n).state_dict().items()}
    assert mine.keys() == ref.keys()
    assert all(mine[k] == ref[k] for k in ref)
    sd = pMCTF(num_me_stages=n).state_dict()
    assert torch.equal(sd["hp_coder.wavelet_transform.lift_h.conv_P1.weight"],
                       sd["hp_coder.wavelet_transform.lift_v.conv_P1.weight"])          # lift_v aliases lift_h
    m = sd["hp_coder.context_fusion.3.ll.maskedConv1.mask"][0, 0]
    assert m.tolist() == [[1, 1, 1], [1, 0, 0], [0, 0, 0]]                               # mask type A
    m = sd["hp_coder.context_fusion.3.ll.maskedConv2.mask"][0, 0]
    assert m.tolist() == [[1, 1, 1], [1, 1, 0], [0, 0, 0]]                               # mask type B


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pmctf_\w+)\s*\(", text)))


def test_c_abi_exports_every_declared_symbol():
    from pMCTF.hip import lib
    hip_syms, rans_syms = _declared("pmctf_hip.h"), _declared("pmctf_rans.h")
    assert len(hip_syms) >= 22 and len(rans_syms) >= 13
    H = ctypes.CDLL(lib.HIP_SO)
    R = ctypes.CDLL(lib.RANS_SO)
    for s in hip_syms:
        assert hasattr(H, s), f"libpmctf_hip.so does not export {s}"
    for s in rans_syms:
        assert hasattr(R, s), f"libpmctf_rans.so does not export {s}"
    assert set(lib.exported_symbols()) == set(hip_syms)
    assert set(lib.rans_exported_symbols()) == set(rans_syms)


def test_invalid_arguments_are_rejected_without_a_gpu():
    from pMCTF.hip import lib
    L = lib.hip()
    assert L.pmctf_conv2d_nhwc_f32(None, None, None, None, None, None, 1, 8, 8, 16, 16, 3, 3, 1, 1, 1, 0, 0.0, None) == -1
    assert L.pmctf_flow_warp_f32(None, None, None, None, None, 1, 1, 8, 8, 1, 1.0, None) == -1
    assert L.pmctf_conv2d_packed_size(112, 112, 3, 3) == 7 * 9 * 7 * 256
    assert L.pmctf_conv2d_packed_bias_size(112) == 112


def test_product_path_has_no_cpu_fallback():
    import pmctf_synth
    from pMCTF.models.video.pMCTF_L import pMCTF
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    net = pMCTF(num_me_stages=1)
    net.update(force=True)
    fr = [list(pmctf_synth.frames_to_tensors(f)) for f in pmctf_synth.synth_yuv420(128, 128, 2)]
    with pytest.raises(RuntimeError):
        net.encode_one_stage(fr[0], fr[1], False, {"mv_feature": None, "ref_mv_y": None}, output_path="/tmp/x.bin",
                             pic_width=128, pic_height=128, skip_decoding=True)
    with pytest.raises(NotImplementedError):
        net.encode_one_stage(fr[0], fr[1], False, {"mv_feature": None, "ref_mv_y": None}, output_path=None)
    with pytest.raises(RuntimeError):       # estimate mode and the decoder need the GPU as well
        net.eval().forward_one_stage(fr[0][0], fr[1][0], 3, True, {"mv_feature": None, "ref_mv_y": None})
    with pytest.raises(RuntimeError):
        net.lp_coder.compress(fr[0][0], [1, 1, 128, 128], "/tmp/x.bin", q_index=3)


def test_product_tables_match_reference_kat():
    import hashlib
    from pMCTF.entropy_models.entropy_models import GaussianEncoder, EntropyCoder, pmf_to_quantized_cdf
    assert pmf_to_quantized_cdf([.1, .2, .3, .35, .05]).tolist() == [0, 6553, 19659, 39319, 62256, 65536]
    g = GaussianEncoder()
    g.update(force=True, entropy_coder=EntropyCoder())
    cdf, ln, off = g.get_cdf_info()
    assert hashlib.sha1(np.ascontiguousarray(cdf, np.int32).tobytes()).hexdigest() == "2e0d0570db9b9fd62dfab006b1ac57759e55eb72"
    assert hashlib.sha1(ln.tobytes()).hexdigest() == "e4fe70191acf4a1d41056176c2599c5d24ac842e"
    assert hashlib.sha1(off.tobytes()).hexdigest() == "93de49f7943bafffbc0cc66c53a21ff2b31c7b6b"
    sc = torch.tensor([0, .5, .5, 1, 2, .01, .02, 4, 4, 8, 8, 64, 100, 1e-9, .3, .3])
    idx = g.build_indexes(sc)
    sym = torch.tensor([0, 1, -1, 2, -3, 0, 0, 5, -7, 40, -60, 0, 1, 0, 0, -1])
    ec = g.entropy_coder
    ec.reset(); g.encode(sym, sc); ec.flush()
    assert ec.get_encoded_stream().hex() == "01f6e0e56434010000eb9e2b66326159b4"
    ec.set_stream(ec.get_encoded_stream())
    assert g.decode_stream(sc, torch.float32, "cpu").int().tolist() == sym.tolist()


@pytest.mark.parametrize("parts,n", [(1, 200003), (2, 200003), (3, 200003), (16, 200003), (2, 1500003)])
def test_host_range_coder_matches_oracle_and_round_trips(parts, n):
    """n = 1 500 003 in two parts makes the first part longer than 65 535 bytes: 4-byte part sizes (flag nibble 0)"""
    from pmctf_oracle import clib, entropy
    from pMCTF.entropy_models.entropy_models import EntropyCoder
    tab = entropy.GaussianTables()
    cdf, ln, off = tab.cdf_info()
    rng = np.random.default_rng(parts)
    idx = rng.integers(0, 256, n).astype(np.int16)
    idx[rng.random(n) < 0.5] = 0
    sym = np.round(rng.laplace(0, 1, n) * tab.scale_table.numpy()[idx] * 1.5).astype(np.int64)
    sym[rng.random(n) < 0.001] = 30000          # long bypass runs
    sym[rng.random(n) < 0.001] = -30000
    sym = np.clip(sym, -30000, 30000).astype(np.int16)
    ec = EntropyCoder(False, parts)
    ec.reset()
    for a, b in ((0, n // 3), (n // 3, n)):      # two pushes
        ec.encode_with_indexes(sym[a:b], idx[a:b], cdf, ln, off)
    ec.flush()
    s = ec.get_encoded_stream()
    assert s[0] == ((parts - 1) << 4) + (0 if n > 1000000 else 1)
    # the oracle's restatement of the reference's N-part container (py_rans.cpp:29-119) writes the same bytes, and
    # decodes what the product wrote
    o = clib.RansEncoder() if parts == 1 else clib.RansEncoderParts(parts)
    o.reset()
    for a, b in ((0, n // 3), (n // 3, n)):
        o.encode_with_indexes(sym[a:b], idx[a:b], cdf, ln, off)
    o.flush()
    assert o.get_encoded_stream().tobytes() == s
    od = clib.RansDecoderParts(parts)
    od.set_stream(s)
    assert np.array_equal(np.concatenate([od.decode_stream(idx[a:b], cdf, ln, off) for a, b in ((0, n // 3), (n // 3, n))]),
                          sym)
    # every part count decodes (one decode_stream per push, as the pushes were split; py_rans.cpp:29-52,174-196)
    ec.set_stream(s)
    out = np.concatenate([ec.decode_stream(torch.from_numpy(idx[a:b]), cdf, ln, off).numpy().astype(np.int16)
                          for a, b in ((0, n // 3), (n // 3, n))])
    assert np.array_equal(out, sym)
    # the threaded encoder (flush in the background, parts in parallel) writes the same bytes
    et = EntropyCoder(True, parts)
    for _ in range(2):                              # twice: reset() after a background flush
        et.reset()
        for a, b in ((0, n // 3), (n // 3, n)):
            et.encode_with_indexes(sym[a:b], idx[a:b], cdf, ln, off)
        et.flush()
        assert et.get_encoded_stream() == s
    with tempfile.TemporaryDirectory() as td:
        from pMCTF.hip import lib
        path = os.path.join(td, "s.bin")
        hdr = b"HEAD"
        size = lib.rans().pmctf_rans_encoder_write_file(ec.encoder, hdr, len(hdr), path.encode())
        assert size == len(hdr) + len(s) == os.path.getsize(path)
        assert open(path, "rb").read() == hdr + s
        # borrow mode (the product's writer threads): the caller's arrays are read at flush(), same bytes
        R = lib.rans()
        eb = R.pmctf_rans_encoder_create(0, parts)
        assert R.pmctf_rans_encoder_set_borrow(eb, 1) == 0
        cdf_n, ln_n, off_n = (np.ascontiguousarray(np.asarray(a), dtype=np.int32) for a in (cdf, ln, off))
        for a, b in ((0, n // 3), (n // 3, n)):
            assert R.pmctf_rans_encoder_encode_with_indexes(eb, sym[a:].ctypes.data, idx[a:].ctypes.data, b - a,
                                                            cdf_n.ctypes.data, cdf_n.shape[0], cdf_n.shape[1],
                                                            ln_n.ctypes.data, off_n.ctypes.data) == 0
        assert R.pmctf_rans_encoder_flush(eb) == 0
        buf = np.empty(R.pmctf_rans_encoder_stream_size(eb), np.uint8)
        assert R.pmctf_rans_encoder_get_encoded_stream(eb, buf.ctypes.data, buf.size) == 0 and buf.tobytes() == s
        bad = np.full(4, cdf_n.shape[0], np.int16)            # a CDF row out of range is refused at the push
        assert R.pmctf_rans_encoder_encode_with_indexes(eb, sym.ctypes.data, bad.ctypes.data, 4, cdf_n.ctypes.data,
                                                        cdf_n.shape[0], cdf_n.shape[1], ln_n.ctypes.data,
                                                        off_n.ctypes.data) == -1
        R.pmctf_rans_encoder_destroy(eb)


def test_stream_framing_and_helpers():
    from pMCTF.utils import stream_helper as sh
    assert sh.get_padding_size(1080, 1920, 128) == (0, 0, 0, 72)
    assert sh.get_downsampled_shape(1152, 1920, 64) == (18, 30)
    assert sh.get_rounded_q(np.array([[[[1.2345]]]])) == (1.23, 123)
    assert sh.get_rounded_q(0.001) == (0.01, 1)
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "a.bin")
        sh.encode_p(b"abc", 0, p)
        assert open(p, "rb").read() == struct.pack(">H", 0) + struct.pack(">I", 3) + b"abc"
        assert sh.decode_p(p) == (0, b"abc")
        sh.encode_image(1080, 1920, 1, b"xyz!", p)
        assert open(p, "rb").read() == struct.pack(">III", 1080, 1920, 1) + struct.pack(">I", 4) + b"xyz!"
        assert sh.decode_image(p) == (1080, 1920, 1, b"xyz!")
    assert sh.image_header(540, 960, 2, 7) == struct.pack(">IIII", 540, 960, 2, 7)


def test_yuv_reader_and_eval_utils():
    import pmctf_synth
    from pMCTF.utils.yuv_reader import YUVReader
    from pMCTF.utils.util import ycbcr2rgb, yuv_420_to_444
    from pMCTF.utils.video_eval_utils import dump_json, generate_log_json, str2bool
    fr = pmctf_synth.synth_yuv420(64, 48, 3)
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "v.yuv")
        with open(p, "wb") as f:
            for y, u, v in fr:
                f.write(y.tobytes()); f.write(u.tobytes()); f.write(v.tobytes())
        r = YUVReader(p, 64, 48)
        for y, u, v in fr:
            Y, Cb, Cr = r.read_one_frame()
            assert np.array_equal(Y, y) and np.array_equal(Cb, u) and np.array_equal(Cr, v)
    t = yuv_420_to_444((torch.rand(1, 1, 8, 8), torch.rand(1, 1, 4, 4), torch.rand(1, 1, 4, 4)))
    assert t.shape == (1, 3, 8, 8)
    rgb = ycbcr2rgb(torch.tensor([[[[100.]], [[128.]], [[128.]]]]))
    assert torch.allclose(rgb, torch.full((1, 3, 1, 1), 100.))
    assert str2bool("1") and not str2bool("no")
    log = generate_log_json(4, [0, 1, 1, 1], [800., 400., 400., 400.], [0, .01, .01, .01], [30., 31., 32., 33.],
                            [29.] * 4, [0.9] * 4, 100, 1.5)
    assert log["i_frame_num"] == 1 and log["p_frame_num"] == 3
    assert abs(log["ave_all_frame_bpp"] - 2000. / 400) < 1e-12 and abs(log["ave_all_frame_psnr"] - 31.5) < 1e-12
    f = io.StringIO(); dump_json(log, f, float_digits=6, indent=2)
    assert json.loads(f.getvalue())["ave_i_frame_bpp"] == 8.0


def _dist_worker(rank, world, port, q):
    import torch.distributed as dist
    import pmctf_dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    n_gops, gop = 5, 4
    mine = pmctf_dist.shard_gops(n_gops, rank, world)
    local = {g: ([1000.0 * g + i for i in range(gop)], [30.0 + g + 0.1 * i for i in range(gop)]) for g in mine}
    bits, psnr = pmctf_dist.gather_gop_metrics(local, n_gops, gop, dist)
    q.put((rank, mine, bits.tolist(), psnr.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_gop_sharding_world_size_2_gloo():
    """N>1 path: every GOP coded by exactly one rank, metrics reassembled in GOP order on all ranks."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dist_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    shards = sorted(sum((r[1] for r in res), []))
    assert shards == [0, 1, 2, 3, 4]
    expect_bits = [[1000.0 * g + i for i in range(4)] for g in range(5)]
    for _, _, bits, psnr in res:
        assert bits == expect_bits
        assert abs(psnr[3][2] - 33.2) < 1e-12


def test_bitstream_inspector_reads_reference_files():
    """tools/inspect_bitstream.py parses files written by the real reference (fixtures) with the standard library only"""
    import subprocess, sys
    from helpers import golden
    g = golden()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        paths = []
        for k in ("dec.file.1.bin", "dec.file.1_mv.bin", "dec.file.1_C_main.bin", "dec.file.0_main.bin"):
            p = os.path.join(td, k[len("dec.file."):])
            open(p, "wb").write(g[k].tobytes())
            paths.append(p)
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "inspect_bitstream.py")] + paths,
                             capture_output=True, text=True, check=True).stdout
    assert "image file: 128x128 (Y)" in out and "image file: 64x64 (UV)" in out and "motion file: mv_y_q_index 0" in out
    assert out.count("1 part(s)") == 4 and "(!)" not in out


class _ChainCodec:
    """A cheap deterministic stand-in with the codec API of pmctf_gop / pmctf_dist: its outputs depend on the inputs AND
    on the motion-context chain, so a wrong schedule, a broken chain or a mis-gathered tensor changes the result."""
    num_me_stages = 4

    @staticmethod
    def _chain(ref, cur, dpb, stage_idx, q_index):
        prev = 0.0 if dpb["mv_feature"] is None else float(dpb["mv_feature"].reshape(-1)[0])
        v = 0.5 * prev + float((cur[0] - ref[0]).mean()) + 0.01 * stage_idx + 0.001 * q_index
        return {"mv_feature": torch.full((1, 2, 2, 2), v), "ref_mv_y": torch.full((1, 1, 1, 1), v * 2)}, v

    def advance_dpb(self, ref_frame, cur_frame, dpb, stage_idx=0, q_index=0):
        return self._chain(ref_frame, cur_frame, dpb, stage_idx, q_index)[0]

    @staticmethod
    def dpb_shapes(height, width):
        return [(1, 2, 2, 2), (1, 1, 1, 1)]

    def encode_one_stage(self, ref_frame, cur_frame, code_lt, dpb, output_path=None, pic_width=None, pic_height=None,
                         psize=128, skip_decoding=True, stage_idx=0, q_index=0, on_dpb=None):
        if callable(dpb):           # the relay delivers the context once the motion has been estimated
            dpb = dpb()
        new, v = self._chain(ref_frame, cur_frame, dpb, stage_idx, q_index)
        if on_dpb is not None:
            on_dpb(new)
        (ry, rc), (cy, cc) = ref_frame, cur_frame
        return {"L_t": (ry + cy) / 2 + v, "L_tc": (rc + cc) / 2 - v, "H_t": cy - ry + v, "H_tc": cc - rc + 2 * v,
                "mv_hat": torch.full((1, 2) + tuple(ry.shape[2:]), v), "dpb": new,
                "bit_H": 1000.0 + round(v * 1e6), "bit_ME": 10.0 + stage_idx, "bit_L": 77.0 if code_lt else None}

    # the pair in parts (pmctf_dist's split of a pair over the ranks a stage leaves idle): same numbers, piecewise
    def encode_pair_motion(self, ref_frame, cur_frame, dpb, output_path, stage_idx=0, q_index=0, me_downsample=1,
                           on_dpb=None):
        if callable(dpb):
            dpb = dpb()
        new, v = self._chain(ref_frame, cur_frame, dpb, stage_idx, q_index)
        if on_dpb is not None:
            on_dpb(new)
        return {"mv_hat": torch.full((1, 2) + tuple(ref_frame[0].shape[2:]), v), "dpb": new, "bit_ME": 10.0 + stage_idx}

    def encode_pair_part(self, ref_planes, cur_planes, mv_hat, chroma, kinds, code_lt, output_path, pic_width, pic_height,
                         stage_idx=0, q_index=0):
        v = float(mv_hat.reshape(-1)[0])
        low = (ref_planes + cur_planes) / 2 + (-v if chroma else v)
        high = cur_planes - ref_planes + (2 * v if chroma else v)
        bits = {}
        if "H" in kinds:
            bits["H"] = 400.0 if chroma else 600.0 + round(v * 1e6)
        if "L" in kinds:
            bits["L"] = 27.0 if chroma else 50.0
        return {"H": high if "H" in kinds else None, "L": low if ("L" in kinds or not code_lt) else None, "bits": bits}


def _pair_worker(rank, world, port, q):
    import torch.distributed as dist
    import pmctf_dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(5)
    frames = [[torch.rand(1, 1, 8, 12, generator=g), torch.rand(2, 1, 4, 6, generator=g)] for _ in range(16)]
    with tempfile.TemporaryDirectory() as td:
        enc = pmctf_dist.encode_gop_pair_sharded(_ChainCodec(), frames, 8, 12, 3, td, rank, world, dist)
    stats = {}
    with tempfile.TemporaryDirectory() as td:       # a second GOP through the same call: results must not depend on history
        enc2 = pmctf_dist.encode_gop_pair_sharded(_ChainCodec(), frames, 8, 12, 3, td, rank, world, dist, stats=stats)
    assert enc2["bits"] == enc["bits"]
    work = [(r.get("pair"), r.get("part")) if "part" in r else ("whole",) for r in enc["results"]]
    q.put((rank, enc["bits"], enc["bits_mv"], [[t if t is None else t.numpy() for t in fc] for fc in enc["frames_coded"]],
           work, stats))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_pair_sharding_inside_a_gop_gloo(world):
    """BASELINE configs[4] layout: the pairs of every temporal stage spread over the ranks, the motion context relayed
    from pair to pair (send/recv), one all-gather of fixed-size pair records per stage.
    Every rank must end with exactly the subband tree and bit counts of the single-process schedule."""
    import torch.multiprocessing as mp
    import pmctf_gop
    g = torch.Generator().manual_seed(5)
    frames = [[torch.rand(1, 1, 8, 12, generator=g), torch.rand(2, 1, 4, 6, generator=g)] for _ in range(16)]
    with tempfile.TemporaryDirectory() as td:
        ref = pmctf_gop.encode_gop(_ChainCodec(), frames, 8, 12, 3, td)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() + world) % 2000
    import pmctf_dist
    procs = [ctx.Process(target=_pair_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # every pair coded exactly once: as a whole by one rank, or — in the stages with at most half as many pairs as ranks —
    # as one motion part plus the parts of pmctf_dist.pair_parts spread over the pair's group of ranks
    whole = sum(sum(1 for w in r[4] if w == ("whole",)) for r in res)
    split_stages = [n for n in (8, 4, 2, 1) if world // n >= 2]
    assert whole == sum(n for n in (8, 4, 2, 1) if world // n < 2)
    parts = sorted(w for r in res for w in r[4] if w != ("whole",))
    expect = []
    for n in split_stages:
        for p in range(n):
            expect.append((p, "motion"))
            expect += [(p, pl + "".join(k)) for pl, k, _ in pmctf_dist.pair_parts(n == 1, world // n)]
    assert parts == sorted(expect)
    if world == 8:      # the last stage: H and L coders of luma and of chroma on four ranks
        who = {w[1]: r[0] for r in res for w in r[4] if w != ("whole",) and w[1] in ("YL", "CL")}
        assert who == {"YL": 1, "CL": 3}            # the L coders run beside the H coders (ranks 0 and 2), not after them
        # only live data moves: per stage exactly (pairs of the stage) x (record of one pair)
        rec = 4 * (2 * 8 * 12 + 2 * 2 * 4 * 6 + 2 * 8 * 12)
        assert all(abs(b - n * rec) <= 64 * n for b, n in zip(res[0][5]["gather_bytes_per_stage"], (8, 4, 2, 1)))
    for rank, bits, bits_mv, fc, _, _ in res:
        assert bits == ref["bits"] and bits_mv == ref["bits_mv"], rank
        for a, b in zip(fc, ref["frames_coded"]):
            for x, y in zip(a, b):
                assert (x is None and y is None) or np.array_equal(x, y.numpy()), rank


def test_range_coder_random_tables_property():
    """Property test over random CDF tables (ragged row lengths, tiny and huge frequencies) and random symbols, in and out
    of the table range (bypass digits): the product's host coder writes the oracle's bytes and decodes them back."""
    from hypothesis import given, settings, strategies as st
    from pmctf_oracle import clib
    from pMCTF.entropy_models.entropy_models import EntropyCoder

    @st.composite
    def case(draw):
        rows = draw(st.integers(1, 6))
        cols = draw(st.integers(4, 40))
        seed = draw(st.integers(0, 2 ** 31 - 1))
        n = draw(st.integers(0, 400))
        return rows, cols, seed, n

    @settings(max_examples=60, deadline=None)
    @given(case())
    def run(c):
        rows, cols, seed, n = c
        r = np.random.default_rng(seed)
        cdf = np.zeros((rows, cols), np.int32)
        sizes = np.zeros(rows, np.int32)
        offs = r.integers(-20, 5, rows).astype(np.int32)
        for i in range(rows):
            k = int(r.integers(2, cols))              # number of coded symbols incl. the escape symbol; size = k + 1
            w = r.random(k) ** 4 + 1e-4               # skewed: some near-minimal frequencies
            pmf = w / w.sum()
            q = clib.pmf_to_quantized_cdf(pmf.astype(np.float32).tolist(), 16)
            cdf[i, :k + 1] = q
            sizes[i] = k + 1
        idx = r.integers(0, rows, n).astype(np.int16)
        span = (sizes[idx] - 2).astype(np.int64)
        sym = (-offs[idx] + r.integers(-3, 3, n) + (r.random(n) < 0.7) * r.integers(0, 1 << 30, n) % np.maximum(span, 1))
        far = r.random(n) < 0.05
        sym = np.where(far, r.integers(-3000, 3000, n), sym).astype(np.int16)
        ec = EntropyCoder(False, 1)
        ec.reset(); ec.encode_with_indexes(sym, idx, cdf, sizes, offs); ec.flush()
        s = ec.get_encoded_stream()
        o = clib.RansEncoder(); o.reset(); o.encode_with_indexes(sym, idx, cdf, sizes, offs); o.flush()
        assert o.get_encoded_stream().tobytes() == s
        ec.set_stream(s)
        out = ec.decode_stream(torch.from_numpy(idx), cdf, sizes, offs).numpy().astype(np.int16)
        assert np.array_equal(out, sym)

    run()


def test_deferred_values_semantics():
    """pMCTF.hip.deferred (pure Python): what the drop-in path hands back instead of finished results.  Arithmetic on a
    deferred number keeps deferring (the harness's `curr_bits = r["bit_H"] + r["bit_ME"]`, `curr_bits / pixels`);
    anything that needs the value forces it exactly once; torch functions and attribute access force a deferred tensor."""
    from pMCTF.hip.deferred import Deferred, DeferredTensor, is_pending, unwrap
    calls = []

    def make(v):
        return lambda: (calls.append(v), v)[1]
    a, b = Deferred(make(1000.0)), Deferred(make(24.0))
    bits = a + b
    bpp = bits / 512
    total = 0.0 + bits                      # generate_log_json-style accumulation
    assert calls == [] and is_pending(a) and isinstance(bpp, Deferred) and isinstance(total, Deferred)
    assert float(bpp) == 2.0 and calls == [1000.0, 24.0]
    assert float(total) == 1024.0 and calls == [1000.0, 24.0]          # forced once
    assert not is_pending(a) and f"{bits:.1f}" == "1024.0" and bits > 1000 and bits == 1024.0
    assert not isinstance(bits, torch.Tensor)
    made = []
    t = DeferredTensor(lambda: (made.append(1), torch.arange(6.0).reshape(1, 1, 2, 3))[1], ready=lambda: bool(made))
    assert is_pending(t)
    assert torch.round(t).shape == (1, 1, 2, 3) and made == [1]        # a torch function forces it
    assert t.shape == (1, 1, 2, 3) and float(t[0, 0, 1, 2]) == 5.0 and float((t + 1).sum()) == 21.0
    assert unwrap({"x": [t, None, 3]})["x"][0] is t.force()
    assert np.asarray(t).shape == (1, 1, 2, 3)
    d = Deferred(lambda: {"H": b"abc"})
    assert d["H"] == b"abc" and "H" in d and len(d) == 1


def _overlap_worker(rank, world, port, q, n_gops):
    import torch.distributed as dist
    import pmctf_dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(7)
    gops = [[[torch.rand(1, 1, 8, 12, generator=g), torch.rand(2, 1, 4, 6, generator=g)] for _ in range(16)]
            for _ in range(n_gops)]
    stats = {}
    ws = pmctf_dist.PairShardWorkspace()
    with tempfile.TemporaryDirectory() as td:
        folders = [os.path.join(td, str(j)) for j in range(n_gops)]
        for f in folders:
            os.makedirs(f)
        for _ in range(2):          # twice through the same workspace: the buffers are reused, the results the same
            encs = pmctf_dist.encode_gops_pair_sharded_overlapped(_ChainCodec(), gops, 8, 12, 3, folders, rank, world, dist,
                                                                  stats=stats, workspace=ws)
    q.put((rank, [(e["bits"], e["bits_mv"], [[t if t is None else t.numpy() for t in fc] for fc in e["frames_coded"]],
                   len(e["results"])) for e in encs], stats))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_gops", [(4, 2), (4, 3), (8, 4)])
def test_pair_sharding_overlapped_gops_gloo(world, n_gops):
    """SURVEY 8e's GOP overlap (pmctf_dist.encode_gops_pair_sharded_overlapped): several closed GOPs in flight over the
    same ranks, their relay chains running in opposite directions so that the late stages land on different ranks.
    Every rank ends with every GOP's subband tree and bit counts exactly as the single-process schedule produces them;
    the relay hop and gather counts are those of the layout."""
    import torch.multiprocessing as mp
    import pmctf_dist
    import pmctf_gop
    # the owner map: every chain runs rank r -> r+1, GOP j of G starts j*N/G ranks further on
    assert [pmctf_dist.pair_owner(p, 8, 0, 2) for p in range(8)] == list(range(8))
    assert [pmctf_dist.pair_owner(p, 8, 1, 2) for p in range(8)] == [4, 5, 6, 7, 0, 1, 2, 3]
    assert [pmctf_dist.pair_owner(p, 8, 1, 4) for p in range(4)] == [2, 3, 4, 5]
    assert [pmctf_dist.pair_owner(p, 8, 3, 4) for p in range(4)] == [6, 7, 0, 1]
    late = {pmctf_dist.pair_owner(p, 8, j, 4) for j in range(4) for p in range(2)}   # the 2-pair stage of four GOPs
    assert len(late) == 8
    for j in range(4):                                  # one direction: the next pair of a chain is on the next rank
        assert all(pmctf_dist.pair_owner(p + 1, 8, j, 4) == (pmctf_dist.pair_owner(p, 8, j, 4) + 1) % 8 for p in range(8))
    g = torch.Generator().manual_seed(7)
    gops = [[[torch.rand(1, 1, 8, 12, generator=g), torch.rand(2, 1, 4, 6, generator=g)] for _ in range(16)]
            for _ in range(n_gops)]
    refs = []
    for fr in gops:
        with tempfile.TemporaryDirectory() as td:
            refs.append(pmctf_gop.encode_gop(_ChainCodec(), fr, 8, 12, 3, td))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() + 7 * world + n_gops) % 2000
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, q, n_gops)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for j in range(n_gops):
        assert sum(r[1][j][3] for r in res) == 15                 # each pair of each GOP coded exactly once
    for rank, encs, stats in res:
        for j, (bits, bits_mv, fc, _) in enumerate(encs):
            assert bits == refs[j]["bits"] and bits_mv == refs[j]["bits_mv"], (rank, j)
            for a, b in zip(fc, refs[j]["frames_coded"]):
                for x, y in zip(a, b):
                    assert (x is None and y is None) or np.array_equal(x, y.numpy()), (rank, j)
        assert len(stats["gather_bytes_per_stage"]) == 4 * n_gops       # one all-gather per GOP and stage
    assert sum(r[2]["relay_hops"] for r in res) == n_gops * (7 + 3 + 1)     # every chain link of every stage, once


def test_capture_gate_excludes_other_threads_while_recording():
    """The gate around model entry points (pMCTF.hip.engine.CaptureGate): any number of threads hold it shared; a thread
    that records a launch plan gives up its own shared hold, waits for the others to leave, holds it exclusively — during
    which nobody can enter — and is back to shared afterwards; two recorders cannot wait for each other."""
    import threading
    import time
    from pMCTF.hip.engine import CaptureGate
    gate = CaptureGate()
    log, lock = [], threading.Lock()

    def say(x):
        with lock:
            log.append(x)

    def worker(name, record):
        for _ in range(3):
            with gate.shared():
                say((name, "in"))
                time.sleep(0.005)
                if record:
                    with gate.exclusive_from_shared():
                        say((name, "rec+"))
                        assert gate.writer and gate.readers == 0
                        time.sleep(0.01)
                        say((name, "rec-"))
                    assert gate.readers >= 1 and not gate.writer
                say((name, "out"))
    ts = [threading.Thread(target=worker, args=(n, r)) for n, r in (("a", True), ("b", True), ("c", False))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(30)
        assert not t.is_alive()                 # no deadlock with two recorders
    assert gate.readers == 0 and not gate.writer
    inside = None                               # nothing of another thread between a recorder's rec+ and rec-
    for name, what in log:
        if what == "rec+":
            assert inside is None
            inside = name
        elif what == "rec-":
            assert inside == name
            inside = None
        else:
            assert inside is None or inside == name, log
    assert sum(1 for _, w in log if w == "rec+") == 6

