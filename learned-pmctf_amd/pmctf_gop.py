"""GOP driver: the temporal-decomposition loop of the reference's evaluation harness
(test_pMCTF_flex.py:run_test, lines 131-327) restated as a reusable function over any codec object
exposing the reference model API (`num_me_stages`, `encode_one_stage`, `inverse_MCTF`).

Used by bench.py (HIP product), the parity tests (oracle vs product) and tools/make_golden.py (the
real reference, in the build container) so that all three run exactly the same schedule:
stage s codes pairs (2k*2^s, 2k*2^s + 2^s); dpb is reset per stage; the last stage also codes L;
then the inverse MCTF runs the stages backwards and PSNR is taken on the un-padded crop.
"""
import math
import os

import torch


def psnr(a, b):
    """test_pMCTF_flex.py:81-84"""
    mse = torch.mean((a - b) ** 2)
    return (20 * torch.log10(255.0 / torch.sqrt(mse))).item()


def ca_psize(me_downsample):
    """padding granularity the content-adaptive harness uses for a motion down-sampling factor (test_pMCTF_CA.py:121-123)"""
    psize = 256 if me_downsample > 2 else 128
    return psize * 2 if me_downsample > 4 else psize


def encode_gop_batched(codec, frames, pic_height, pic_width, q_index, bin_folder, psize=128):
    """The schedule of encode_gop with the pairs of each temporal stage handed to the codec in ONE call
    (codec.encode_stage_pairs): identical files, bits and tensors, larger launches.  skip_decoding=True only."""
    gop = len(frames)
    stages = int(round(math.log2(gop)))
    assert 2 ** stages == gop and gop >= 2
    frames_coded = [None] * gop
    bits = [None] * gop
    bits_mv = [None] * gop
    results = []
    num_frames = gop
    for stage_idx in range(stages):
        num_frames //= 2
        step = 2 ** stage_idx
        code_lt = (stage_idx + 1) == stages
        me_num = min(codec.num_me_stages - 1, stage_idx)
        idx = [(g * 2 * step, g * 2 * step + step) for g in range(num_frames)]
        if stage_idx == 0:
            pairs = [(frames[a], frames[b]) for a, b in idx]
        else:
            pairs = [(frames_coded[a][:2], frames_coded[b][:2]) for a, b in idx]
        paths = [os.path.join(bin_folder, f"{b}.bin") for _, b in idx]
        rs, _ = codec.encode_stage_pairs(pairs, code_lt, {"mv_feature": None, "ref_mv_y": None}, paths,
                                         pic_width=pic_width, pic_height=pic_height, psize=psize, stage_idx=me_num,
                                         q_index=q_index)
        for (i_ref, i_cur), r in zip(idx, rs):
            frames_coded[i_ref] = [r["L_t"], r["L_tc"], None]
            frames_coded[i_cur] = [r["H_t"], r["H_tc"], r["mv_hat"]]
            bits[i_cur] = float(r["bit_H"] + r["bit_ME"])
            bits_mv[i_cur] = float(r["bit_ME"])
            if code_lt:
                bits[i_ref] = float(r["bit_L"])
                bits_mv[i_ref] = 0.0
            results.append(r)
    return {"bits": bits, "bits_mv": bits_mv, "frames_coded": frames_coded, "results": results, "stages": stages}


def encode_gops_batched(codec, gops, pic_height, pic_width, q_index, bin_folders, psize=128):
    """K closed GOPs at once: stage s of ALL of them goes to the codec in one encode_stage_pairs call (the GOPs are
    independent units, test_pMCTF_flex.py:131-141, and use the same coder weights), so the late stages — 2 and 1 pairs
    per GOP — are batched K-fold as well.  The motion context restarts at every GOP boundary (chain_reset); each GOP's
    files go to its own folder.  Returns one encode_gop-style dict per GOP; files, bits and tensors are identical to
    coding the GOPs one after the other."""
    K = len(gops)
    gop = len(gops[0])
    stages = int(round(math.log2(gop)))
    assert 2 ** stages == gop and gop >= 2 and all(len(g) == gop for g in gops) and len(bin_folders) == K
    outs = [{"bits": [None] * gop, "bits_mv": [None] * gop, "frames_coded": [None] * gop, "results": [],
             "stages": stages} for _ in range(K)]
    num_frames = gop
    for stage_idx in range(stages):
        num_frames //= 2
        step = 2 ** stage_idx
        code_lt = (stage_idx + 1) == stages
        me_num = min(codec.num_me_stages - 1, stage_idx)
        idx = [(g * 2 * step, g * 2 * step + step) for g in range(num_frames)]
        pairs, paths = [], []
        for k in range(K):
            src = gops[k] if stage_idx == 0 else [f[:2] for f in outs[k]["frames_coded"]]
            pairs += [(src[a], src[b]) for a, b in idx]
            paths += [os.path.join(bin_folders[k], f"{b}.bin") for _, b in idx]
        rs, _ = codec.encode_stage_pairs(pairs, code_lt, {"mv_feature": None, "ref_mv_y": None}, paths,
                                         pic_width=pic_width, pic_height=pic_height, psize=psize, stage_idx=me_num,
                                         q_index=q_index, chain_reset=[k * num_frames for k in range(1, K)])
        for k in range(K):
            o = outs[k]
            for (i_ref, i_cur), r in zip(idx, rs[k * num_frames:(k + 1) * num_frames]):
                o["frames_coded"][i_ref] = [r["L_t"], r["L_tc"], None]
                o["frames_coded"][i_cur] = [r["H_t"], r["H_tc"], r["mv_hat"]]
                o["bits"][i_cur] = float(r["bit_H"] + r["bit_ME"])
                o["bits_mv"][i_cur] = float(r["bit_ME"])
                if code_lt:
                    o["bits"][i_ref] = float(r["bit_L"])
                    o["bits_mv"][i_ref] = 0.0
                o["results"].append(r)
    return outs


def encode_gop(codec, frames, pic_height, pic_width, q_index, bin_folder, skip_decoding=True, psize=128,
               on_pair=None, me_downsample=1, store_only=False):
    """frames: list (len = GOP size, power of two) of [Y (1,1,Hp,Wp), UV (2,1,Hp/2,Wp/2)] padded tensors.
    Returns dict(bits[], frames_coded (after forward), results[] per pair in coding order).
    me_downsample > 1: the schedule of test_pMCTF_CA.py:code_one_gop (motion at reduced resolution; pad the frames to
    ca_psize(me_downsample) and pass that as psize).
    store_only=True: a caller that keeps the results of every pair WITHOUT the harness's two prints (not what
    test_pMCTF_flex.py does): with the codec's opt-in deferral (lazy_stages) the pairs of a stage are then coded as one
    batch."""
    gop = len(frames)
    stages = int(round(math.log2(gop)))
    assert 2 ** stages == gop and gop >= 2
    frames_coded = [None] * gop
    bits = [None] * gop
    bits_mv = [None] * gop
    results = []
    log = []
    num_frames = gop
    for stage_idx in range(stages):
        num_frames //= 2
        dpb = {"mv_feature": None, "ref_mv_y": None}
        for group_idx in range(num_frames):
            step = 2 ** stage_idx
            i_ref = group_idx * 2 * step
            i_cur = i_ref + step
            if stage_idx == 0:
                y_ref, c_ref = frames[i_ref]
                y_cur, c_cur = frames[i_cur]
            else:
                y_ref, c_ref, mv_r = frames_coded[i_ref]
                y_cur, c_cur, mv_c = frames_coded[i_cur]
                assert mv_r is None and mv_c is None
            code_lt = (stage_idx + 1) == stages
            me_num = min(codec.num_me_stages - 1, stage_idx)
            # bin_folder None: the estimate-only branch of encode_one_stage (test_pMCTF_CA.py:153-154)
            bin_path = os.path.join(bin_folder, f"{i_cur}.bin") if bin_folder is not None else None
            r = codec.encode_one_stage(ref_frame=[y_ref, c_ref], cur_frame=[y_cur, c_cur], output_path=bin_path,
                                       pic_height=pic_height, pic_width=pic_width, stage_idx=me_num,
                                       code_lt=code_lt, psize=psize, skip_decoding=skip_decoding, dpb=dpb,
                                       q_index=q_index, **({"me_downsample": me_downsample} if me_downsample != 1 else {}))
            frames_coded[i_ref] = [r["L_t"], r["L_tc"], None]
            frames_coded[i_cur] = [r["H_t"], r["H_tc"], r["mv_hat"]]
            dpb = r["dpb"]
            # what the harness does with the numbers of every pair, statement for statement (test_pMCTF_flex.py:236-258):
            # the two f-strings it prints LOOK at the bit counts right here, before the next call
            curr_bits = r["bit_H"] + r["bit_ME"]
            if isinstance(curr_bits, torch.Tensor):
                curr_bits = curr_bits.item()
            tmp = r["bit_ME"] / curr_bits
            if not store_only:
                log.append(f"percentage MV: {tmp*100} %")
            bits[i_cur] = curr_bits
            bit_me = r["bit_ME"].item() if isinstance(r["bit_ME"], torch.Tensor) else r["bit_ME"]
            bits_mv[i_cur] = bit_me
            if not store_only:
                log.append(f"Frame {i_cur}: {curr_bits / (pic_height * pic_width)} bpp")
            if code_lt:
                curr_bits = r["bit_L"]
                if isinstance(curr_bits, torch.Tensor):
                    curr_bits = curr_bits.item()
                bits[i_ref] = curr_bits
                bits_mv[i_ref] = 0.0
            results.append(r)
            if on_pair is not None:
                on_pair(stage_idx, i_ref, i_cur, r)
    bits = [None if b is None else float(b) for b in bits]
    bits_mv = [None if b is None else float(b) for b in bits_mv]
    frames_coded = [[t if t is None or isinstance(t, torch.Tensor) else t.force() for t in fc] for fc in frames_coded]
    return {"bits": bits, "bits_mv": bits_mv, "frames_coded": frames_coded, "results": results, "stages": stages,
            "log": log}


def decode_gop(codec, frames_coded, luma_stage0=False):
    """Temporal synthesis, test_pMCTF_flex.py:268-291.  Modifies and returns frames_coded.
    luma_stage0: the content-adaptive harness's variant, which reconstructs luma with stage 0's lifting filters at every
    stage (inverse_MCTF without stage_idx, test_pMCTF_CA.py:239)."""
    gop = len(frames_coded)
    stages = int(round(math.log2(gop)))
    num_frames = 1
    for stage_idx in reversed(range(stages)):
        if stage_idx != stages - 1:
            num_frames *= 2
        for group_idx in reversed(range(num_frames)):
            step = 2 ** stage_idx
            i_ref = group_idx * 2 * step
            L_t, L_tc, mv_ref = frames_coded[i_ref]
            H_t, H_tc, mv_hat = frames_coded[i_ref + step]
            assert mv_ref is None
            me_num = min(codec.num_me_stages - 1, stage_idx)
            ref, cur = codec.inverse_MCTF(L_t, H_t, mv_hat, stage_idx=0 if luma_stage0 else me_num)
            ref_c, cur_c = codec.inverse_MCTF(L_tc, H_tc, mv_hat, stage_idx=me_num, downscale=True)
            frames_coded[i_ref] = [ref, ref_c, None]
            frames_coded[i_ref + step] = [cur, cur_c, None]
    return frames_coded


def gop_psnr(frames_rec, frames_orig, pic_height, pic_width):
    """YUV-PSNR (6Y+Cb+Cr)/8 per frame on the un-padded crop, test_pMCTF_flex.py:294-325."""
    out = []
    for (rec_y, rec_c, _), (y, c) in zip(frames_rec, frames_orig):
        ry = torch.round(rec_y.clamp(0, 255.0))[:, :, :pic_height, :pic_width]
        rc = torch.round(rec_c.clamp(0, 255.0))[:, :, :pic_height // 2, :pic_width // 2]
        oy = y[:, :, :pic_height, :pic_width]
        oc = c[:, :, :pic_height // 2, :pic_width // 2]
        py = psnr(ry, oy)
        pcb = psnr(rc[0:1], oc[0:1])
        pcr = psnr(rc[1:2], oc[1:2])
        out.append({"y": py, "cb": pcb, "cr": pcr, "yuv": (6.0 * py + pcb + pcr) / 8.0})
    return out


def write_yuv(path, frames_u8):
    """[(Y, Cb, Cr) uint8 arrays] -> planar 8-bit 4:2:0 file, the layout YUVReader / image_import read"""
    with open(path, "wb") as f:
        for planes in frames_u8:
            for p in planes:
                f.write(p.tobytes(order="C"))


def read_gop(reader, gop, device, psize=128):
    """GOP pictures from a YUVReader as the model's inputs: ([Y (1,1,Hp,Wp), UV (2,1,Hp/2,Wp/2)] zero padded right/bottom
    to multiples of psize (chroma psize/2), the un-padded originals, (height, width)).  What the harness does per pair
    at stage 0 (test_pMCTF_flex.py:151-192), done here per GOP."""
    import torch.nn.functional as F
    from pMCTF.utils.stream_helper import get_padding_size
    padded, orig, size = [], [], None
    for _ in range(gop):
        y, cb, cr = (torch.from_numpy(p).float() for p in reader.read_one_frame())
        assert size in (None, tuple(y.shape)), "picture size changes inside the sequence"
        size = tuple(y.shape)
        luma = y[None, None].to(device)
        chroma = torch.stack((cb, cr))[:, None].to(device)
        left, right, top, bottom = get_padding_size(size[0], size[1], p=psize)
        orig.append([luma, chroma])
        padded.append([F.pad(luma, (left, right, top, bottom)),
                       F.pad(chroma, (left // 2, right // 2, top // 2, bottom // 2))])
    return padded, orig, size


def rgb_psnr(rec_y, rec_c, y, c):
    """PSNR of the rounded RGB pictures (chroma bilinearly up-sampled), test_pMCTF_flex.py:312-321"""
    from pMCTF.utils.util import ycbcr2rgb, yuv_420_to_444
    to_rgb = lambda luma, chroma: torch.round(ycbcr2rgb(yuv_420_to_444((luma, chroma[0:1], chroma[1:2]))))
    return psnr(to_rgb(y, c), to_rgb(rec_y, rec_c))


def encode_sequence(codec, yuv_path, width, height, frame_num, gop, q_index, bin_folder, device,
                    skip_decoding=True, psize=128):
    """What the evaluation harness produces for one sequence (test_pMCTF_flex.py:run_test, 86-346) built from this
    module's own pieces: pictures come from a planar .yuv through YUVReader and get_padding_size, every closed GOP goes
    through encode_gop (one encode_one_stage call per pair, both per-pair report lines), decode_gop and gop_psnr, and
    the per-frame tables are folded into the harness's log record by generate_log_json / dump_json.  MS-SSIM is
    reported as 0 (pytorch_msssim is a third-party package the harness imports; not part of the path).
    Returns {"log": record, "json": its text, "bits", "bpp_mv", "psnr", "psnr_rgb", "frame_types", "lines"}."""
    import io
    import time
    from pMCTF.utils.video_eval_utils import dump_json, generate_log_json
    from pMCTF.utils.yuv_reader import YUVReader
    assert frame_num % gop == 0
    reader = YUVReader(yuv_path, width, height, start_index=0)
    tables = {k: [] for k in ("bits", "bpp_mv", "psnr", "psnr_rgb", "frame_types")}
    lines = []
    pairs = 0
    seconds = {"encoding_time": 0.0, "decoding_time": 0.0}
    t0 = time.time()
    with torch.no_grad():
        for _ in range(frame_num // gop):
            padded, orig, (h, w) = read_gop(reader, gop, device, psize)
            enc = encode_gop(codec, padded, h, w, q_index, bin_folder, skip_decoding=skip_decoding, psize=psize)
            for r in enc["results"]:
                pairs += 1
                for k in seconds:
                    seconds[k] += r[k]
            lines += enc["log"]
            rec = decode_gop(codec, enc["frames_coded"])
            quality = gop_psnr(rec, orig, h, w)
            tables["bits"] += enc["bits"]
            tables["bpp_mv"] += [b / (h * w) for b in enc["bits_mv"]]
            tables["psnr"] += [p["yuv"] for p in quality]
            tables["frame_types"] += [0] + [1] * (gop - 1)          # the one coded L picture of a GOP, then its H pictures
            for (ry, rc, _), (y, c) in zip(rec, orig):
                crop_y = torch.round(ry.clamp(0, 255.0))[:, :, :h, :w]
                crop_c = torch.round(rc.clamp(0, 255.0))[:, :, :h // 2, :w // 2]
                tables["psnr_rgb"].append(rgb_psnr(crop_y, crop_c, y, c))
    reader.close()
    for k, label in (("encoding_time", "encoding"), ("decoding_time", "decoding")):
        lines.append(f"{label} {pairs} P frames, average {seconds[k] / pairs * 1000:.0f} ms.")
    record = generate_log_json(frame_num, tables["frame_types"], tables["bits"], tables["bpp_mv"], tables["psnr"],
                               tables["psnr_rgb"], [0] * frame_num, height * width, time.time() - t0)
    text = io.StringIO()
    dump_json(record, text, float_digits=6, indent=2)
    return dict(tables, log=record, json=text.getvalue(), lines=lines)
