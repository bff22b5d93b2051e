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
#   * ONE collective per stage while it has at least as many pairs as ranks: an all-gather of one fixed-size byte record
#     per pair — L, L chroma, H, H chroma, the motion field (5 * H * W floats = 44 MB at 1080p) and three bit counts as
#     float64 — into a buffer allocated once per stage; `frames_coded` holds views of it (no second copy).  Every rank
#     ends with the complete subband tree.
#   * A stage with at most half as many pairs as ranks (the late stages: 4, 2, 1 pairs) is shared in PARTS: the four spatial
#     coder calls of a pair are independent given mv_hat, so the ranks the stage would leave idle take the chroma coders
#     (and, in the stage that codes L, the L coders) of its pairs; the tree is then reassembled by broadcasts of exactly
#     the tensors that were produced (pair_parts, _encode_stage_split).
#
# Results are identical to pmctf_gop.encode_gop on one device.  On the GPU node the backend is "nccl" (= RCCL over xGMI:
# device tensors, all_gather_into_tensor); under "gloo" (CPU tests, one-GPU rehearsal) records are staged through host
# memory.
def pair_owner(pair_idx, world, gop_idx=0, n_gops=1):
    """rank that codes pair `pair_idx` of a stage.  Closed GOPs coded together (encode_gops_pair_sharded_overlapped) all
    run their chains in the SAME direction (rank r hands the motion context to rank r+1) from shifted starts — GOP j of G:
    pair k on rank k + j*N/G — so that the late stages of the GOPs (4, 2, 1 pairs) land on different ranks.  One direction
    matters under RCCL: the point-to-point operations between two ranks share one communicator and execute in issue
    order, so contexts travelling r -> r+1 for one GOP and r+1 -> r for another could wait for each other for ever."""
    shift = (gop_idx * world) // max(1, n_gops)
    return (pair_idx + shift) % world


def _is_nccl(dist):
    return dist.get_backend() == "nccl"


class _Relay:
    """Hands the motion codec's context from the owner of pair k-1 to the owner of pair k (point-to-point): ONE message
    per hop — both tensors (mv_feature, ref_mv_y; channels-last storage) travel in one contiguous buffer.  A receive
    buffer and a send buffer are allocated once.  Aliasing contract: what recv() returns are VIEWS of the receive
    buffer, valid until this rank's next recv(); the motion codec consumes the context (copies it into its own
    storage, in stream order) before the rank asks for the next one.  Send buffers come from a small pool that grows to
    the number of sends a rank ever has in flight (a send never waits for an earlier one: no new wait edges between the
    chains of GOPs in flight) and is reused from then on."""

    def __init__(self, dist, rank, world, device, shapes):
        self.dist, self.rank, self.world, self.device = dist, rank, world, device
        self.shapes = shapes                     # logical NCHW shapes of (mv_feature, ref_mv_y)
        self.pool = []                           # [send buffer, its last isend or None]
        self.comm_dev = device if _is_nccl(dist) else "cpu"
        self.numel = [n * c * h * w for (n, c, h, w) in shapes]
        total = sum(self.numel)
        self.rbuf = torch.empty(total, dtype=torch.float32, device=self.comm_dev)
        self.hops = 0
        self.bytes_per_hop = 4 * total

    def _views(self, buf):
        out, o = {}, 0
        for key, (n, c, h, w), cnt in zip(("mv_feature", "ref_mv_y"), self.shapes, self.numel):
            out[key] = buf[o:o + cnt].view(n, h, w, c)
            o += cnt
        return out

    def recv(self, src):
        self.dist.recv(self.rbuf, src=src)
        local = self.rbuf if self.rbuf.device == torch.device(self.device) else self.rbuf.to(self.device)
        return {k: v.permute(0, 3, 1, 2) for k, v in self._views(local).items()}

    def send(self, dpb, dst):
        self.hops += 1
        slot = next((s for s in self.pool if s[1] is None or s[1].is_completed()), None)
        if slot is None:
            slot = [torch.empty(sum(self.numel), dtype=torch.float32, device=self.comm_dev), None]
            self.pool.append(slot)
        for key, v in self._views(slot[0]).items():
            v.copy_(dpb[key].permute(0, 2, 3, 1))
        slot[1] = self.dist.isend(slot[0], dst=dst)

    def drain(self):
        for slot in self.pool:
            if slot[1] is not None:
                slot[1].wait()
                slot[1] = None


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


class PairShardWorkspace:
    """The gather buffers of encode_gop[s]_pair_sharded, allocated ONCE and reused GOP after GOP (a caller that codes a
    sequence passes the same workspace to every call; without one, every call allocates its own).  frames_coded of a
    call are views of these buffers: they are valid until the next call that uses the workspace."""

    def __init__(self):
        self.bufs = {}

    def get(self, key, shape, device):
        t = self.bufs.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.device != torch.device(device):
            t = self.bufs[key] = torch.empty(shape, dtype=torch.uint8, device=device)
        return t


def pair_parts(code_lt, group):
    """How one pair is cut when `group` ranks share it (a stage with fewer pairs than ranks): a list of
    (planes, kinds, offset) — planes "Y" (luma) or "C" (chroma), kinds the spatial coders of those planes this part runs,
    offset the rank inside the group (0 = the pair's owner, which also estimates and codes the motion).  The four coder
    calls of a pair are independent given mv_hat (pMCTF_L.py:398-420,570-592): luma / chroma on two ranks, and in the
    stage that codes L the H and L coders of each on two more."""
    if group >= 4 and code_lt:
        return [("Y", ("H",), 0), ("Y", ("L",), 1), ("C", ("H",), 2), ("C", ("L",), 3)]
    kinds = ("H", "L") if code_lt else ("H",)
    return [("Y", kinds, 0), ("C", kinds, 1)]


def encode_gop_pair_sharded(codec, frames, pic_height, pic_width, q_index, bin_folder, rank=0, world=1, dist=None,
                            psize=128, stats=None, workspace=None):
    """Same schedule and same return value as pmctf_gop.encode_gop (bits, bits_mv, frames_coded; `results` holds only
    this rank's pairs), with the pairs of every stage spread over the ranks.  Every rank ends up with the complete
    subband tree (frames_coded), so any of them can run pmctf_gop.decode_gop.

    The codec needs, beyond the reference API: `dpb_shapes(height, width)` and the keyword arguments `dpb` (may be a
    callable, evaluated once the motion has been estimated) and `on_dpb` (called with the new context as soon as the
    motion codec has produced it) of encode_one_stage.
    stats: dict that receives gather_bytes_per_stage, relay_hops, relay_bytes_per_hop of this rank."""
    return encode_gops_pair_sharded_overlapped(codec, [frames], pic_height, pic_width, q_index, [bin_folder], rank, world,
                                               dist, psize, stats, workspace)[0]


def encode_gops_pair_sharded_overlapped(codec, gops, pic_height, pic_width, q_index, bin_folders, rank=0, world=1,
                                        dist=None, psize=128, stats=None, workspace=None):
    """Several closed GOPs in flight over the same ranks (SURVEY 8e's GOP overlap): stage s of ALL of them is coded before
    stage s+1 of any, and the chains of the GOPs start on DIFFERENT ranks but all run in the same direction, rank r to rank
    r+1 (pair_owner: one direction is what keeps the chains deadlock-free under RCCL), so the ranks one GOP leaves idle
    in its late stages (4, 2, 1 pairs) code the other's.  With 8 ranks and two GOP-16s the critical path is
    2 + 1 + 1 + 1 pair-times for 32 frames instead of 2 x 4.  Per GOP-stage: one relay chain and ONE all-gather, exactly as
    in encode_gop_pair_sharded; a rank works through its pairs of a stage in the order of their position in their chain
    (lowest first: every wait is for a pair with a lower position, so the chains cannot deadlock, and the sends r -> r+1 of
    the GOPs are issued in the order rank r+1 issues its receives).  With fewer than three ranks the two neighbours of a
    rank are the same rank and one communicator would carry both directions: under RCCL the GOPs are then coded one
    after the other.  Returns one encode_gop-style dict per GOP, identical to coding them one after the other."""
    import math
    import os
    G = len(gops)
    gop = len(gops[0])
    stages = int(round(math.log2(gop)))
    assert 2 ** stages == gop and gop >= 2 and all(len(g) == gop for g in gops) and len(bin_folders) == G
    if world > 1 and dist is None:
        raise ValueError("pair sharding over world > 1 ranks needs an initialised torch.distributed")
    device = gops[0][0][0].device
    multi = dist is not None and world > 1
    if multi and G > 1 and world < 3 and _is_nccl(dist):
        return [encode_gops_pair_sharded_overlapped(codec, [g], pic_height, pic_width, q_index, [f], rank, world, dist, psize,
                                                    stats, workspace)[0] for g, f in zip(gops, bin_folders)]
    y0, c0 = gops[0][0]
    shapes = [tuple(y0.shape), tuple(c0.shape), tuple(y0.shape), tuple(c0.shape), (1, 2) + tuple(y0.shape[2:])]
    offs, sc_off, rec_bytes = _record_layout(shapes)
    relay = _Relay(dist, rank, world, device, codec.dpb_shapes(y0.shape[2], y0.shape[3])) if multi else None
    ws = workspace if workspace is not None else PairShardWorkspace()
    outs = [{"bits": [None] * gop, "bits_mv": [None] * gop, "frames_coded": [None] * gop, "results": [],
             "stages": stages} for _ in range(G)]
    gather_bytes = []
    num_frames = gop
    comm_dev = device if (not multi or _is_nccl(dist)) else "cpu"
    for stage_idx in range(stages):
        num_frames //= 2
        step = 2 ** stage_idx
        code_lt = (stage_idx + 1) == stages
        me_num = min(codec.num_me_stages - 1, stage_idx)
        if multi and G == 1 and world // num_frames >= 2 and hasattr(codec, "encode_pair_part"):
            # fewer pairs than ranks: the ranks the stage would leave idle take PARTS of its pairs
            _encode_stage_split(codec, gops[0], outs[0], stage_idx, num_frames, code_lt, me_num, pic_height, pic_width,
                                q_index, bin_folders[0], psize, rank, world, dist, relay, ws, shapes, offs, rec_bytes,
                                comm_dev, device, gather_bytes)
            continue
        slots = (num_frames + world - 1) // world
        mine = [ws.get(("mine", j, stage_idx), (slots, rec_bytes), device) for j in range(G)]
        last_dpb = [{"mv_feature": None, "ref_mv_y": None} for _ in range(G)]
        # this rank's pairs of the stage, lowest chain position first
        tasks = sorted((p, j) for j in range(G) for p in range(num_frames) if pair_owner(p, world, j, G) == rank)
        for p, j in tasks:
            o = outs[j]
            i_ref = p * 2 * step
            i_cur = i_ref + step
            if stage_idx == 0:
                (y_ref, c_ref), (y_cur, c_cur) = gops[j][i_ref], gops[j][i_cur]
            else:
                y_ref, c_ref, _ = o["frames_coded"][i_ref]
                y_cur, c_cur, _ = o["frames_coded"][i_cur]
            if p == 0:
                dpb_in = {"mv_feature": None, "ref_mv_y": None}
            elif world == 1:
                dpb_in = last_dpb[j]
            else:
                dpb_in = (lambda src=pair_owner(p - 1, world, j, G): relay.recv(src))

            def on_dpb(d, p=p, j=j):
                if multi and p + 1 < num_frames:
                    relay.send(d, pair_owner(p + 1, world, j, G))

            r = codec.encode_one_stage(ref_frame=[y_ref, c_ref], cur_frame=[y_cur, c_cur],
                                       output_path=os.path.join(bin_folders[j], f"{i_cur}.bin"), pic_height=pic_height,
                                       pic_width=pic_width, stage_idx=me_num, code_lt=code_lt, psize=psize,
                                       skip_decoding=True, dpb=dpb_in, q_index=q_index, on_dpb=on_dpb)
            last_dpb[j] = r["dpb"]
            o["results"].append(r)
            rec = mine[j][p // world]
            for (o_, n), t in zip(offs, (r["L_t"], r["L_tc"], r["H_t"], r["H_tc"], r["mv_hat"])):
                rec[o_:o_ + 4 * n].view(torch.float32).copy_(t.reshape(-1))
            sc = torch.tensor([float(r["bit_H"]), float(r["bit_ME"]), float(r["bit_L"]) if code_lt else 0.0],
                              dtype=torch.float64)
            rec[sc_off:sc_off + 24].copy_(sc.view(torch.uint8))
        if relay is not None:
            relay.drain()
        # ---- the one collective per GOP of the stage (same order on every rank)
        for j in range(G):
            o = outs[j]
            if multi and num_frames < world:
                # fewer live records than ranks (several GOPs in flight, or a codec without the part API): every record
                # is broadcast by its owner — the all-gather below would move world x slots records for num_frames live ones
                everything = ws.get(("all", j, stage_idx), (world, slots, rec_bytes), comm_dev)
                for p in range(num_frames):
                    own = pair_owner(p, world, j, G)
                    if own == rank:
                        everything[own, 0].copy_(mine[j][0])
                    dist.broadcast(everything[own, 0], src=own)
                everything = everything.to(device)
                gather_bytes.append(num_frames * rec_bytes)
            elif multi:
                everything = ws.get(("all", j, stage_idx), (world, slots, rec_bytes), comm_dev)
                if _is_nccl(dist):
                    dist.all_gather_into_tensor(everything.view(-1), mine[j].view(-1))
                else:
                    dist.all_gather(list(everything.unbind(0)), mine[j].to(comm_dev))
                    everything = everything.to(device)
                gather_bytes.append(world * slots * rec_bytes)
            else:
                everything = mine[j].unsqueeze(0)
            scalars = everything[:, :, sc_off:sc_off + 24].contiguous().cpu().view(torch.float64)     # (world, slots, 3)
            for p in range(num_frames):
                i_ref = p * 2 * step
                i_cur = i_ref + step
                own = pair_owner(p, world, j, G)
                rec = everything[own, p // world]
                L_t, L_tc, H_t, H_tc, mv_hat = [rec[o_:o_ + 4 * n].view(torch.float32).view(shp)
                                                for (o_, n), shp in zip(offs, shapes)]
                bit_H, bit_ME, bit_L = scalars[own, p // world].tolist()
                o["frames_coded"][i_ref] = [L_t, L_tc, None]
                o["frames_coded"][i_cur] = [H_t, H_tc, mv_hat]
                o["bits"][i_cur] = bit_H + bit_ME
                o["bits_mv"][i_cur] = bit_ME
                if code_lt:
                    o["bits"][i_ref] = bit_L
                    o["bits_mv"][i_ref] = 0.0
    if stats is not None:
        stats["gather_bytes_per_stage"] = gather_bytes
        stats["relay_hops"] = relay.hops if relay is not None else 0
        stats["relay_bytes_per_hop"] = relay.bytes_per_hop if relay is not None else 0
    return outs


def _encode_stage_split(codec, frames, out, stage_idx, num_frames, code_lt, me_num, pic_height, pic_width, q_index,
                        bin_folder, psize, rank, world, dist, relay, ws, shapes, offs, rec_bytes, comm_dev, device,
                        gather_bytes):
    """One temporal stage of ONE GOP with fewer pairs than ranks: pair p belongs to the group of `world // num_frames`
    ranks starting at p * group.  The group's first rank (the owner) estimates and codes the motion — the motion context
    still travels owner to owner — and hands mv_hat to the others; every part (pair_parts) runs forward_MCTF for its planes
    and its spatial coder(s) and writes its files.  The subband tree is reassembled by one broadcast per produced tensor
    (only live data moves: 5 x H x W floats per pair in total, as in the all-gather record) and one all-reduce of the bit
    counts.  Results are those of encode_one_stage per pair."""
    import os
    group = world // num_frames
    step = 2 ** stage_idx
    parts = pair_parts(code_lt, group)
    nccl = _is_nccl(dist)
    recs = ws.get(("split", stage_idx), (num_frames, rec_bytes), comm_dev)
    views = lambda p: [recs[p, o_:o_ + 4 * n].view(torch.float32).view(shp) for (o_, n), shp in zip(offs, shapes)]
    bits = torch.zeros(num_frames, 5, dtype=torch.float64)                # H luma, H chroma, motion, L luma, L chroma
    to_comm = lambda t: t if t.device == torch.device(comm_dev) else t.to(comm_dev)
    pending = []
    my_parts = [(p, part) for p in range(num_frames) for part in parts if p * group + part[2] == rank]
    for p, (planes, kinds, off) in my_parts:
        owner = p * group
        i_ref = p * 2 * step
        i_cur = i_ref + step
        ref = frames[i_ref] if stage_idx == 0 else out["frames_coded"][i_ref][:2]
        cur = frames[i_cur] if stage_idx == 0 else out["frames_coded"][i_cur][:2]
        path = os.path.join(bin_folder, f"{i_cur}.bin")
        L_t, L_tc, H_t, H_tc, mv_rec = views(p)
        if off == 0:
            dpb_in = {"mv_feature": None, "ref_mv_y": None} if p == 0 else (lambda src=owner - group: relay.recv(src))

            def on_dpb(d, p=p):
                if p + 1 < num_frames:
                    relay.send(d, (p + 1) * group)
            m = codec.encode_pair_motion([ref[0], ref[1]], [cur[0], cur[1]], dpb_in, path, stage_idx=me_num, q_index=q_index,
                                         on_dpb=on_dpb)
            mv_hat = m["mv_hat"]
            mv_rec.copy_(mv_hat)
            bits[p, 2] = float(m["bit_ME"])
            for other in sorted({o for _, _, o in parts if o != 0}):
                pending.append(dist.isend(mv_rec if nccl else mv_rec.contiguous(), dst=owner + other))
            out["results"].append({"pair": p, "part": "motion", "dpb": m["dpb"]})
        else:
            dist.recv(mv_rec, src=owner)
            mv_hat = mv_rec.to(device)
        chroma = planes == "C"
        r = codec.encode_pair_part(ref[1] if chroma else ref[0], cur[1] if chroma else cur[0], mv_hat, chroma, kinds,
                                   code_lt, path, pic_width, pic_height, stage_idx=me_num, q_index=q_index)
        if r["H"] is not None:
            (H_tc if chroma else H_t).copy_(r["H"])
            bits[p, 1 if chroma else 0] = float(r["bits"]["H"])
        if r["L"] is not None:
            (L_tc if chroma else L_t).copy_(r["L"])
            if "L" in kinds:
                bits[p, 4 if chroma else 3] = float(r["bits"]["L"])
        out["results"].append({"pair": p, "part": planes + "".join(kinds)})
    for w in pending:
        w.wait()
    if relay is not None:
        relay.drain()
    # ---- reassemble: who produced what (the same table on every rank), one broadcast per tensor, live data only
    moved = 0
    for p in range(num_frames):
        owner = p * group
        producer = {"mv": owner}
        for planes, kinds, off in parts:
            c = planes == "C"
            if "H" in kinds:
                producer["Hc" if c else "H"] = owner + off
            if "L" in kinds or not code_lt:
                producer["Lc" if c else "L"] = owner + off
        for name, t in zip(("L", "Lc", "H", "Hc", "mv"), views(p)):
            dist.broadcast(t, src=producer[name])
            moved += 4 * t.numel()
    bits_c = bits.to(comm_dev) if nccl else bits
    dist.all_reduce(bits_c, op=dist.ReduceOp.SUM)
    bits = bits_c.cpu()
    gather_bytes.append(moved)
    local = recs if recs.device == torch.device(device) else recs.to(device)
    for p in range(num_frames):
        i_ref = p * 2 * step
        i_cur = i_ref + step
        L_t, L_tc, H_t, H_tc, mv_hat = [local[p, o_:o_ + 4 * n].view(torch.float32).view(shp)
                                        for (o_, n), shp in zip(offs, shapes)]
        out["frames_coded"][i_ref] = [L_t, L_tc, None]
        out["frames_coded"][i_cur] = [H_t, H_tc, mv_hat]
        b = bits[p].tolist()
        out["bits"][i_cur] = b[0] + b[1] + b[2]
        out["bits_mv"][i_cur] = b[2]
        if code_lt:
            out["bits"][i_ref] = b[3] + b[4]
            out["bits_mv"][i_ref] = 0.0
