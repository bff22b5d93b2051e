"""Multi-GPU scheduling of the pMCTF encode path: one process per GPU, closed GOPs are independent units.

Rank r of `world` encodes GOPs r, r+world, r+2*world, ... with no data-path collective (GOP-level data
parallelism, weak scaling).  Only per-frame metrics (bits, PSNR) are gathered at the end, as small
float64 tensors, with torch.distributed (backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in CPU tests).
"""
import torch


def shard_gops(n_gops, rank, world):
    return list(range(rank, n_gops, world))


def gather_gop_metrics(local, n_gops, gop, dist=None, device="cpu"):
    """local: {gop_index: (bits[gop], psnr[gop])} coded by this rank -> on every rank two [n_gops, gop] float64
    tensors in GOP order."""
    bits = torch.zeros(n_gops, gop, dtype=torch.float64, device=device)
    psnr = torch.zeros(n_gops, gop, dtype=torch.float64, device=device)
    for g, (b, p) in local.items():
        bits[g] = torch.as_tensor(b, dtype=torch.float64)
        psnr[g] = torch.as_tensor(p, dtype=torch.float64)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        # each GOP is owned by exactly one rank and zero elsewhere: a sum reassembles the table
        dist.all_reduce(bits, op=dist.ReduceOp.SUM)
        dist.all_reduce(psnr, op=dist.ReduceOp.SUM)
    return bits, psnr


# ----------------------------------------------------------------------------------------------------------------------
# Pair-level sharding inside ONE GOP (the layout of BASELINE configs[4]): stage s has GOP/2^(s+1)
# pairs; pair k of the stage goes to rank k % world.
#
#   * SpyNet of a pair depends on nothing but its two frames: every rank runs it for its own pair at once.
#   * The motion codec's context (`dpb`: mv_feature 1x64xH/4xW/4 + ref_mv_y 1x64xH/16xW/16, 37 MB at 1080p) is the one
#     chain between the pairs of a stage (pMCTF_L.py:448-495): it travels as a RELAY — rank(k) receives it from
#     rank(k-1), runs the motion codec (0.33 TFLOP, ~8 ms on an MI355X), sends it on to rank(k+1) the moment it exists
#     and only then codes the rest of its pair.  Critical path per stage: (pairs-1) hops of (motion codec + 37 MB over
#     one xGMI link) ~ 8-9 ms each; the other 93 % of a pair's work runs concurrently on all ranks.
#   * ONE collective per stage: an all-gather of one fixed-size byte record per pair — L, L chroma, H, H chroma, the
#     motion field (5 * H * W floats = 44 MB at 1080p) and three bit counts as float64 — into a buffer allocated once per
#     stage; `frames_coded` holds views of it (no second copy).  Every rank ends with the complete subband tree.
#
# Results are identical to pmctf_gop.encode_gop on one device.  On the GPU node the backend is "nccl" (= RCCL over xGMI:
# device tensors, all_gather_into_tensor); under "gloo" (CPU tests, one-GPU rehearsal) records are staged through host
# memory.
def pair_owner(pair_idx, world):
    return pair_idx % world


def _is_nccl(dist):
    return dist.get_backend() == "nccl"


class _Relay:
    """Hands the motion codec's context from the owner of pair k-1 to the owner of pair k (point-to-point)."""

    def __init__(self, dist, rank, world, device, shapes):
        self.dist, self.rank, self.world, self.device = dist, rank, world, device
        self.shapes = shapes                     # logical NCHW shapes of (mv_feature, ref_mv_y)
        self.pending = []
        self.comm_dev = device if _is_nccl(dist) else "cpu"

    def recv(self, src):
        out = {}
        for key, (n, c, h, w) in zip(("mv_feature", "ref_mv_y"), self.shapes):
            buf = torch.empty((n, h, w, c), dtype=torch.float32, device=self.comm_dev)   # channels-last storage
            self.dist.recv(buf, src=src)
            out[key] = buf.to(self.device).permute(0, 3, 1, 2)
        return out

    def send(self, dpb, dst):
        for key in ("mv_feature", "ref_mv_y"):
            t = dpb[key].permute(0, 2, 3, 1).contiguous().to(self.comm_dev)
            self.pending.append((self.dist.isend(t, dst=dst), t))

    def drain(self):
        for work, _ in self.pending:
            work.wait()
        self.pending = []


def _record_layout(shapes):
    """byte offsets of the tensors of one pair record: five float32 tensors, then three float64 scalars (8-aligned)"""
    offs, o = [], 0
    for shp in shapes:
        n = 1
        for d in shp:
            n *= d
        offs.append((o, n))
        o += 4 * n
    o = (o + 7) // 8 * 8
    return offs, o, o + 24


def encode_gop_pair_sharded(codec, frames, pic_height, pic_width, q_index, bin_folder, rank=0, world=1, dist=None,
                            psize=128):
    """Same schedule and same return value as pmctf_gop.encode_gop (bits, bits_mv, frames_coded; `results` holds only
    this rank's pairs), with the pairs of every stage spread over the ranks.  Every rank ends up with the complete
    subband tree (frames_coded), so any of them can run pmctf_gop.decode_gop.

    The codec needs, beyond the reference API: `dpb_shapes(height, width)` and the keyword arguments `dpb` (may be a
    callable, evaluated once the motion has been estimated) and `on_dpb` (called with the new context as soon as the
    motion codec has produced it) of encode_one_stage."""
    import math
    import os
    gop = len(frames)
    stages = int(round(math.log2(gop)))
    assert 2 ** stages == gop and gop >= 2
    device = frames[0][0].device
    multi = dist is not None and world > 1
    y0, c0 = frames[0]
    shapes = [tuple(y0.shape), tuple(c0.shape), tuple(y0.shape), tuple(c0.shape), (1, 2) + tuple(y0.shape[2:])]
    offs, sc_off, rec_bytes = _record_layout(shapes)
    relay = _Relay(dist, rank, world, device, codec.dpb_shapes(y0.shape[2], y0.shape[3])) if multi else None
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
        slots = (num_frames + world - 1) // world
        comm_dev = device if (not multi or _is_nccl(dist)) else "cpu"
        mine_buf = torch.empty((slots, rec_bytes), dtype=torch.uint8, device=device)
        last_dpb = {"mv_feature": None, "ref_mv_y": None}
        for p in range(rank, num_frames, world):
            i_ref = p * 2 * step
            i_cur = i_ref + step
            if stage_idx == 0:
                (y_ref, c_ref), (y_cur, c_cur) = frames[i_ref], frames[i_cur]
            else:
                y_ref, c_ref, _ = frames_coded[i_ref]
                y_cur, c_cur, _ = frames_coded[i_cur]
            if p == 0:
                dpb_in = {"mv_feature": None, "ref_mv_y": None}
            elif world == 1:
                dpb_in = last_dpb
            else:
                dpb_in = (lambda src=pair_owner(p - 1, world): relay.recv(src))

            def on_dpb(d, p=p):
                if multi and p + 1 < num_frames:
                    relay.send(d, pair_owner(p + 1, world))

            r = codec.encode_one_stage(ref_frame=[y_ref, c_ref], cur_frame=[y_cur, c_cur],
                                       output_path=os.path.join(bin_folder, f"{i_cur}.bin"), pic_height=pic_height,
                                       pic_width=pic_width, stage_idx=me_num, code_lt=code_lt, psize=psize,
                                       skip_decoding=True, dpb=dpb_in, q_index=q_index, on_dpb=on_dpb)
            last_dpb = r["dpb"]
            results.append(r)
            rec = mine_buf[p // world]
            for (o, n), t in zip(offs, (r["L_t"], r["L_tc"], r["H_t"], r["H_tc"], r["mv_hat"])):
                rec[o:o + 4 * n].view(torch.float32).copy_(t.reshape(-1))
            sc = torch.tensor([float(r["bit_H"]), float(r["bit_ME"]), float(r["bit_L"]) if code_lt else 0.0],
                              dtype=torch.float64)
            rec[sc_off:sc_off + 24].copy_(sc.view(torch.uint8))
        if relay is not None:
            relay.drain()
        # ---- the one collective of the stage
        if multi:
            everything = torch.empty((world, slots, rec_bytes), dtype=torch.uint8, device=comm_dev)
            if _is_nccl(dist):
                dist.all_gather_into_tensor(everything.view(-1), mine_buf.view(-1))
            else:
                dist.all_gather(list(everything.unbind(0)), mine_buf.to(comm_dev))
                everything = everything.to(device)
        else:
            everything = mine_buf.unsqueeze(0)
        scalars = everything[:, :, sc_off:sc_off + 24].contiguous().cpu().view(torch.float64)     # (world, slots, 3)
        for p in range(num_frames):
            i_ref = p * 2 * step
            i_cur = i_ref + step
            rec = everything[pair_owner(p, world), p // world]
            L_t, L_tc, H_t, H_tc, mv_hat = [rec[o:o + 4 * n].view(torch.float32).view(shp)
                                            for (o, n), shp in zip(offs, shapes)]
            bit_H, bit_ME, bit_L = scalars[pair_owner(p, world), p // world].tolist()
            frames_coded[i_ref] = [L_t, L_tc, None]
            frames_coded[i_cur] = [H_t, H_tc, mv_hat]
            bits[i_cur] = bit_H + bit_ME
            bits_mv[i_cur] = bit_ME
            if code_lt:
                bits[i_ref] = bit_L
                bits_mv[i_ref] = 0.0
    return {"bits": bits, "bits_mv": bits_mv, "frames_coded": frames_coded, "results": results, "stages": stages}
