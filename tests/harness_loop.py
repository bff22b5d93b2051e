"""What the reference's evaluation script does with a model, restated for the tests (test infrastructure).

`run_sequence` follows `run_test` of test_pMCTF_flex.py (lines 86-346) statement for statement where the statements
touch the model or its results: frames come from a planar .yuv file through `YUVReader`, are padded with
`get_padding_size`, every pair goes through `encode_one_stage` with the very keyword arguments of lines 214-223, and
what comes back is used exactly as lines 225-258 use it — stored in `frames_coded`, `.to(device)` on the next stage,
`isinstance(..., torch.Tensor)` checks, BOTH per-pair f-strings (they look at the bit counts before the next call),
`np.mean` over the stage's bpp list — followed by the temporal synthesis (268-291), the PSNR block (294-325) and
`generate_log_json` / `dump_json` (338-346, video_eval_utils.py).  Only MS-SSIM is left out (`pytorch_msssim` is not
installed here; the harness's own `pic_height > 128` guard is where it would go).

Nothing here is product code: the product must work with the UNMODIFIED script; this is the closest the GPU box (where
/root/reference does not exist) can get to running it.
"""
import io
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from pMCTF.utils.stream_helper import get_padding_size
from pMCTF.utils.util import ycbcr2rgb, yuv_420_to_444
from pMCTF.utils.video_eval_utils import dump_json, generate_log_json
from pMCTF.utils.yuv_reader import YUVReader


def write_yuv(path, frames_u8):
    """frames_u8: [(Y, Cb, Cr) uint8 arrays] -> planar 4:2:0 file as image_import / YUVReader read it"""
    with open(path, "wb") as f:
        for y, cb, cr in frames_u8:
            f.write(np.ascontiguousarray(y).tobytes())
            f.write(np.ascontiguousarray(cb).tobytes())
            f.write(np.ascontiguousarray(cr).tobytes())


def _np_image_to_tensor(img):       # test_pMCTF_flex.py:68-71
    return torch.from_numpy(img).type(torch.FloatTensor).unsqueeze(0)


def _psnr(a, b):                    # test_pMCTF_flex.py:80-83
    mse = torch.mean((a - b) ** 2)
    return (20 * torch.log10(255.0 / torch.sqrt(mse))).item()


def run_sequence(video_net, vid_path, width, height, frame_num, gop_size, q_index, bin_folder, device,
                 skip_decoding=True, prints=None):
    """Returns (log_result, per-frame bits, per-frame YUV-PSNR, printed lines)."""
    out = prints if prints is not None else []
    say = out.append
    num_stages = 1
    while 2 ** num_stages < gop_size:
        num_stages += 1
    assert 2 ** num_stages == gop_size and frame_num % gop_size == 0
    gop_num = frame_num // gop_size
    src_reader = YUVReader(vid_path, width, height, start_index=0)
    frame_types, psnrs, msssims, rgb_psnrs = ([None] * frame_num for _ in range(4))
    bits, bpps, bpp_mv = ([None] * frame_num for _ in range(3))
    frame_pixel_num = 0
    start_time = time.time()
    p_frame_number = 0
    overall_p_decoding_time = overall_p_encoding_time = 0
    with torch.no_grad():
        for gop_idx in range(gop_num):
            frames_coded = [None] * gop_size
            frames_orig = [None] * gop_size
            num_frames = gop_size
            hp_last_stage = 0
            for stage_idx in range(num_stages):
                num_frames = num_frames // 2
                hp_bpp_cur_stage = []
                dpb = {"mv_feature": None, "ref_mv_y": None}
                for group_idx in range(num_frames):
                    group_step = 2 ** stage_idx
                    frame_idx_gop = group_idx * 2 * group_step
                    frame_idx = gop_idx * gop_size + frame_idx_gop
                    if stage_idx == 0:
                        ycbcr_ref = [_np_image_to_tensor(c) for c in src_reader.read_one_frame()]
                        ycbcr_cur = [_np_image_to_tensor(c) for c in src_reader.read_one_frame()]
                        y_ref, cb_ref, cr_ref = ycbcr_ref
                        chroma_ref = torch.cat((cb_ref, cr_ref), dim=0)
                        y_cur, cb_cur, cr_cur = ycbcr_cur
                        chroma_cur = torch.cat((cb_cur, cr_cur), dim=0)
                        y_ref = y_ref.unsqueeze(0).to(device)
                        y_cur = y_cur.unsqueeze(0).to(device)
                        chroma_ref = chroma_ref.unsqueeze(1).to(device)
                        chroma_cur = chroma_cur.unsqueeze(1).to(device)
                        frames_orig[frame_idx_gop] = [y_ref, chroma_ref]
                        frames_orig[frame_idx_gop + group_step] = [y_cur, chroma_cur]
                        pic_height, pic_width = y_ref.shape[2], y_ref.shape[3]
                        if frame_pixel_num == 0:
                            frame_pixel_num = pic_height * pic_width
                        else:
                            assert frame_pixel_num == pic_height * pic_width
                        psize = 128
                        pl, pr, pt, pb = get_padding_size(pic_height, pic_width, p=psize)
                        y_ref_p = F.pad(y_ref, (pl, pr, pt, pb), mode="constant", value=0)
                        y_cur_p = F.pad(y_cur, (pl, pr, pt, pb), mode="constant", value=0)
                        chroma_ref_p = F.pad(chroma_ref, (pl // 2, pr // 2, pt // 2, pb // 2), mode="constant", value=0)
                        chroma_cur_p = F.pad(chroma_cur, (pl // 2, pr // 2, pt // 2, pb // 2), mode="constant", value=0)
                    else:
                        y_ref_p, chroma_ref_p, mv_ref = frames_coded[frame_idx_gop]
                        y_cur_p, chroma_cur_p, mv_cur = frames_coded[frame_idx_gop + group_step]
                        assert mv_ref is None and mv_cur is None
                    bin_path = os.path.join(bin_folder, f"{frame_idx_gop + group_step}.bin") if bin_folder else None
                    code_lt = (stage_idx + 1) == num_stages
                    me_num = min(video_net.num_me_stages - 1, stage_idx)
                    y_ref_p = y_ref_p.to(device)
                    y_cur_p = y_cur_p.to(device)
                    chroma_ref_p = chroma_ref_p.to(device)
                    chroma_cur_p = chroma_cur_p.to(device)
                    result = video_net.encode_one_stage(ref_frame=[y_ref_p, chroma_ref_p],
                                                        cur_frame=[y_cur_p, chroma_cur_p], output_path=bin_path,
                                                        pic_height=pic_height, pic_width=pic_width, stage_idx=me_num,
                                                        code_lt=code_lt, psize=psize, skip_decoding=skip_decoding,
                                                        dpb=dpb, q_index=q_index)
                    frames_coded[frame_idx_gop] = [result["L_t"], result["L_tc"], None]
                    frames_coded[frame_idx_gop + group_step] = [result["H_t"], result["H_tc"], result["mv_hat"]]
                    dpb = result["dpb"]
                    frame_types[frame_idx + group_step] = 1
                    p_frame_number += 1
                    overall_p_decoding_time += result["decoding_time"]
                    overall_p_encoding_time += result["encoding_time"]
                    curr_bits = result["bit_H"] + result["bit_ME"]
                    if isinstance(curr_bits, torch.Tensor):
                        curr_bits = curr_bits.item()
                    tmp = result["bit_ME"] / curr_bits
                    say(f"percentage MV: {tmp*100} %")
                    bpps[frame_idx + group_step] = curr_bits / frame_pixel_num
                    bits[frame_idx + group_step] = curr_bits
                    if isinstance(result["bit_ME"], torch.Tensor):
                        result["bit_ME"] = result["bit_ME"].item()
                    bpp_mv[frame_idx + group_step] = result["bit_ME"] / frame_pixel_num
                    say(f"Frame {frame_idx+group_step}: {bpps[frame_idx+group_step] } bpp")
                    hp_bpp_cur_stage.append(bpps[frame_idx + group_step])
                    if code_lt:
                        frame_types[frame_idx] = 0
                        curr_bits = result["bit_L"]
                        if isinstance(curr_bits, torch.Tensor):
                            curr_bits = curr_bits.item()
                        bpps[frame_idx] = curr_bits / frame_pixel_num
                        bits[frame_idx] = curr_bits
                        bpp_mv[frame_idx] = 0
                say(f"STAGE {stage_idx} completed")
                if hp_last_stage != 0:
                    hp_cur_stage = np.mean(hp_bpp_cur_stage)
                    say(f"Average bpp HP {hp_cur_stage} current stage and {hp_last_stage}  prev stage ")
                hp_last_stage = np.mean(hp_bpp_cur_stage)
            # temporal synthesis, test_pMCTF_flex.py:268-291
            for stage_idx in reversed(range(num_stages)):
                num_frames = 1 if stage_idx == num_stages - 1 else num_frames * 2
                for group_idx in reversed(range(num_frames)):
                    group_step = 2 ** stage_idx
                    frame_idx_gop = group_idx * 2 * group_step
                    L_t, L_tc, mv_ref = frames_coded[frame_idx_gop]
                    H_t, H_tc, mv_hat = frames_coded[frame_idx_gop + group_step]
                    assert mv_ref is None
                    me_num = min(video_net.num_me_stages - 1, stage_idx)
                    ref_frame, cur_frame = video_net.inverse_MCTF(L_t, H_t, mv_hat, stage_idx=me_num)
                    ref_frame_c, cur_frame_c = video_net.inverse_MCTF(L_tc, H_tc, mv_hat, stage_idx=me_num, downscale=True)
                    frames_coded[frame_idx_gop] = [ref_frame, ref_frame_c, None]
                    frames_coded[frame_idx_gop + group_step] = [cur_frame, cur_frame_c, None]
            # PSNR, test_pMCTF_flex.py:294-325
            for frame_idx_gop in range(gop_size):
                frame_idx = gop_idx * gop_size + frame_idx_gop
                cur_frame, cur_frame_c, mv_ref = frames_coded[frame_idx_gop]
                y_cur, chroma_cur = frames_orig[frame_idx_gop]
                assert mv_ref is None
                cur_frame_rec = torch.round(cur_frame.clamp_(0, 255.0))
                cur_frame_c = torch.round(cur_frame_c.clamp_(0, 255.0))
                y_hat_cur = F.pad(cur_frame_rec, (-pl, -pr, -pt, -pb))
                y_psnr_cur = _psnr(y_hat_cur, y_cur)
                c_hat_cur = F.pad(cur_frame_c, (-pl // 2, -pr // 2, -pt // 2, -pb // 2))
                cb_psnr_cur = _psnr(c_hat_cur[0:1], chroma_cur[0:1])
                cr_psnr_cur = _psnr(c_hat_cur[1:2], chroma_cur[1:2])
                ycbcr_444_hat = yuv_420_to_444((y_hat_cur, *c_hat_cur.chunk(2, 0)))
                ycbcr_444_orig = yuv_420_to_444((y_cur, *chroma_cur.chunk(2, 0)))
                x_rgb = torch.round(ycbcr2rgb(ycbcr_444_orig))
                x_hat_rgb = torch.round(ycbcr2rgb(ycbcr_444_hat))
                psnrs[frame_idx] = (6.0 * y_psnr_cur + cb_psnr_cur + cr_psnr_cur) / 8.0
                rgb_psnrs[frame_idx] = _psnr(x_rgb, x_hat_rgb)
                msssims[frame_idx] = 0
    test_time = time.time() - start_time
    say(f"encoding {p_frame_number} P frames, average {overall_p_encoding_time / p_frame_number * 1000:.0f} ms.")
    say(f"decoding {p_frame_number} P frames, average {overall_p_decoding_time / p_frame_number * 1000:.0f} ms.")
    log_result = generate_log_json(frame_num, frame_types, bits, bpp_mv, psnrs, rgb_psnrs, msssims, frame_pixel_num,
                                   test_time)
    buf = io.StringIO()
    dump_json(log_result, buf, float_digits=6, indent=2)           # the script's last step (test_pMCTF_flex.py:476)
    return log_result, bits, psnrs, out, buf.getvalue()
