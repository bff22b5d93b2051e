"""HipEngine — the pMCTF encode hot path on MI355X.

Sequences the gfx950 kernels of libpmctf_hip.so (through pMCTF.hip.ops) for every function of
SURVEY.md §8(a): SpyNet, flow warp, temporal predict/update lifting, the MV codec, the learned
spatial DWT, the four-step context-fusion entropy model with its conv-LSTM context, the post-processing
CNN and the symbol hand-off to the host range coder.  Python only allocates tensors and orders
launches; no numeric torch op runs on this path.

Data layout: single-channel planes are (N,1,H,W) dense; feature maps are NHWC (N,H,W,C) dense.
Weights are packed once per prefix (first use) from the owning module's state_dict.
Arithmetic follows the PM-F32 spec, which makes results bit-identical to the oracle's
C restatement (oracle/, test infrastructure) and byte-identical bitstreams.
"""
import contextlib
import math
import threading
from concurrent.futures import ThreadPoolExecutor
import ctypes as C
import os

import numpy as np
import torch

from . import aten_rules
from . import lib as _lib
from . import ops
from .ops import (ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_TANH, EW_ADD, EW_ADD_MULS2, EW_ADD_MULS_MULS, EW_CLAMP_MULS,
                  EW_COPY, EW_DIVS, EW_MULS, EW_ROUND_CLAMP_MULS, EW_SUB, EW_SUB_MULS2, EW_TANH, as_nchw, ew)

QP_NUM = 21
SCALE_L = 1.149604398860241     # iWave1D scale_l / scale_h, lifting_1d.py:62,100-101
SCALE_H = 0.869864451624781


def get_curr_q(q_scale, q_index):
    """pMCTF_L.py:195-209 / pWave.py:209-225 — host scalar arithmetic on the CPU copy of the parameter."""
    min_q = q_scale[0:1]
    max_q = q_scale[1:2]
    step = (torch.log(max_q) - torch.log(min_q)) / (QP_NUM - 1)
    return torch.exp(torch.log(min_q) + step * q_index)


def get_rounded_q(q_scale):
    """stream_helper.py:41-45"""
    q_scale = float(np.clip(np.asarray(q_scale, dtype=np.float64).reshape(-1)[0], 0.01, 655.))
    q_index = int(np.round(q_scale * 100))
    return q_index / 100, q_index


class SymbolStream:
    """Device-side int16 (symbol, CDF row) buffers of one bitstream, filled push by push in coding order."""

    def __init__(self, total, device, merge=True):
        self.sym = torch.empty(total, dtype=torch.int16, device=device)
        self.idx = torch.empty(total, dtype=torch.int16, device=device)
        self.off = 0
        self.total = total
        self.segments = []          # (offset, count, table_name)
        self.merge = merge          # False: keep one segment per push (needed to slice a batched stream per plane)

    def take(self, n, table):
        off = self.off
        assert off + n <= self.total, "symbol stream overflow"
        self.off += n
        if self.merge and self.segments and self.segments[-1][2] == table and \
                self.segments[-1][0] + self.segments[-1][1] == off:
            o, c, t = self.segments[-1]
            self.segments[-1] = (o, c + n, t)
        else:
            self.segments.append((off, n, table))
        return off

    def to_host(self):
        """Asynchronous D2H into pinned memory; returns (sym, idx, event)."""
        hs = torch.empty(self.total, dtype=torch.int16, pin_memory=True)
        hi = torch.empty(self.total, dtype=torch.int16, pin_memory=True)
        self.copy_to(hs, hi)
        ev = torch.cuda.Event()
        ev.record()
        return hs, hi, ev

    def copy_to(self, hs, hi):
        """the two asynchronous D2H copies alone, into the caller's pinned buffers (a captured launch plan records
        them as copy nodes and marks completion with an event behind the whole segment)"""
        assert self.off == self.total == hs.numel() == hi.numel(), (self.off, self.total, hs.numel())
        hs.copy_(self.sym, non_blocking=True)
        hi.copy_(self.idx, non_blocking=True)


class BitSink:
    """Estimate mode: where a SymbolStream collects symbols for the range coder, a BitSink collects the Laplace bit
    estimates of the same elements (device float64, one accumulator per plane)."""

    def __init__(self, planes, device):
        self.bits = torch.zeros(planes, dtype=torch.float64, device=device)


class RangeCoderPool:
    """Host range coding overlapped with GPU work: one job = one bitstream file.  ctypes releases the GIL
    inside libpmctf_rans.so, so streams are coded in parallel on worker threads."""

    def __init__(self, workers=4):
        self.pool = ThreadPoolExecutor(max_workers=workers)
        self.local = threading.local()

    def _encoder(self):
        e = getattr(self.local, "enc", None)
        if e is None:
            e = _lib.rans().pmctf_rans_encoder_create(0, 1)
            if not e:
                raise RuntimeError("pmctf_rans_encoder_create failed")
            _lib.rans().pmctf_rans_encoder_set_borrow(e, 1)     # the pinned symbol buffers live as long as the job
            self.local.enc = e
        return e

    def _run(self, hs, hi, ev, segments, tables, header, path, keep):
        R = _lib.rans()
        ev.synchronize()
        enc = self._encoder()
        _lib.check(R.pmctf_rans_encoder_reset(enc), "rans reset")
        s_np, i_np = hs.numpy(), hi.numpy()
        for off, cnt, tname in segments:
            cdf, sizes, offsets = tables[tname]
            _lib.check(R.pmctf_rans_encoder_encode_with_indexes(
                enc, s_np[off:].ctypes.data, i_np[off:].ctypes.data, cnt, cdf.ctypes.data, cdf.shape[0], cdf.shape[1],
                sizes.ctypes.data, offsets.ctypes.data), "rans encode_with_indexes")
        _lib.check(R.pmctf_rans_encoder_flush(enc), "rans flush")
        hdr = header(R.pmctf_rans_encoder_stream_size(enc))
        if path is not None:
            size = R.pmctf_rans_encoder_write_file(enc, hdr, len(hdr), path.encode())
            if size < 0:
                raise IOError(f"writing {path} failed ({size})")
        else:
            size = len(hdr) + R.pmctf_rans_encoder_stream_size(enc)
        data = None
        if keep:
            n = R.pmctf_rans_encoder_stream_size(enc)
            buf = np.empty(n, np.uint8)
            _lib.check(R.pmctf_rans_encoder_get_encoded_stream(enc, buf.ctypes.data, n), "rans get stream")
            data = hdr + buf.tobytes()
        if keep:      # the symbols / CDF rows of exactly this bitstream, in coding order
            trace = (np.concatenate([s_np[o:o + c] for o, c, _ in segments]) if segments else s_np[:0],
                     np.concatenate([i_np[o:o + c] for o, c, _ in segments]) if segments else i_np[:0])
            return size, data, trace
        return size, data, None

    def submit(self, stream, tables, header, path, keep=False):
        hs, hi, ev = stream.to_host()
        return self.pool.submit(self._run, hs, hi, ev, list(stream.segments), tables, header, path, keep)

    def submit_host(self, hs, hi, ev, segments, tables, header, path, keep=False):
        """the symbols are (being) copied into the pinned buffers hs / hi; `ev` marks the end of those copies"""
        return self.pool.submit(self._run, hs, hi, ev, segments, tables, header, path, keep)

    def submit_planes(self, stream, nplanes, groups, tables, keep=False):
        """One device stream holds the symbols of `nplanes` independently coded planes (every push is plane-major, as
        the reference flattens NCHW).  groups: [(first_plane, n_planes, header_fn, path)] -> one bitstream per group,
        made of that group's slice of every push.  One D2H copy serves all groups.  Returns a list of futures."""
        hs, hi, ev = stream.to_host()
        futures = []
        for lo, cnt_planes, header, path in groups:
            segs = []
            for off, cnt, tname in stream.segments:
                assert cnt % nplanes == 0
                per = cnt // nplanes
                segs.append((off + lo * per, cnt_planes * per, tname))
            futures.append(self.pool.submit(self._run, hs, hi, ev, segs, tables, header, path, keep))
        return futures


class CaptureGate:
    """Stream capture tolerates no device synchronisation, pinned allocation or allocator growth anywhere in the PROCESS
    while it lasts, so a host thread that records a launch plan needs the device to itself: the entry points of every
    model on that device hold ONE gate shared (CaptureGate.of(device)), a recording holds it exclusively (it gives up its
    own shared hold first, so two recorders cannot wait for each other).  With one host thread — the reference harness,
    also when it runs several models one after the other — the gate is never contended; with one model per host thread
    (an RD sweep run in parallel) a recording waits for the other models' calls in flight and they wait for it."""
    _per_device = {}
    _per_device_lock = threading.Lock()

    @classmethod
    def of(cls, device):
        key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
        with cls._per_device_lock:
            gate = cls._per_device.get(key)
            if gate is None:
                gate = cls._per_device[key] = cls()
            return gate

    def __init__(self):
        self.cond = threading.Condition()
        self.readers = 0
        self.writer = False

    @contextlib.contextmanager
    def shared(self):
        with self.cond:
            while self.writer:
                self.cond.wait()
            self.readers += 1
        try:
            yield
        finally:
            with self.cond:
                self.readers -= 1
                self.cond.notify_all()

    @contextlib.contextmanager
    def exclusive_from_shared(self):
        with self.cond:
            self.readers -= 1
            self.cond.notify_all()
            while self.writer or self.readers > 0:
                self.cond.wait()
            self.writer = True
        try:
            yield
        finally:
            with self.cond:
                self.writer = False
                self.readers += 1
                self.cond.notify_all()


class HipEngine:
    PRECISIONS = {"f32": 0, "f32-chain": 0, "bf16x3": 3, "bf16x2": 2, "bf16": 1}

    def __init__(self, state_dict, num_me_stages, device, gaussian_tables, bit_est_tables, decomp_levels=4,
                 coder_threads=4, precision="f32"):
        """precision: "f32" = PM-F32, the product's arithmetic: bit-exact against the oracle, and ATen's summation order
        in EVERY layer (sum_rule) plus ATen's thread split of torch.sigmoid — what the reference's CPU path computes, to
        the bit, on frames whose planes ATen evaluates through oneDNN or its small-plane sgemm path: the written files
        equal the reference's byte for byte and its files decode.
        "f32-chain": the entropy-parameter networks (context fusion, LL network, conv-LSTM) keep one chain from the bias
        instead (their dominant kernel then needs no second accumulator set: ~5 % faster); coefficients and motion are
        still the reference's, but a CDF row flips now and then, so a few files differ in bytes and, rarely, a stream
        moves by a 32-bit word.  No parity claim beyond that.
        "bf16x3" / "bf16x2" / "bf16": the AUXILIARY reduced-precision profile — the dense 3x3 convolutions with 64 / 112
        couts on planes of at least ops.SPLIT_MIN_PX pixels run on bf16 MFMA with operands split into 3 / 2 / 1 planes
        (conv_split.hip; the other layers as in "f32-chain").  Encoder and decoder built with the same profile agree bit
        for bit; results differ from PM-F32 in the last bits, so no parity claim is made for it."""
        if precision not in self.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(self.PRECISIONS)}")
        self.precision = precision
        self.nsplit = self.PRECISIONS[precision]
        self.aten_all = precision == "f32"
        # ... including ATen's split of torch.sigmoid over its intra-op threads (the scalar tails of the threads' slices go
        # through libm's expf): 8 threads, the machine the fixtures under tests/golden were generated on
        self.aten_threads = int(os.environ.get("PMCTF_ATEN_THREADS", "8"))
        if not torch.cuda.is_available():
            raise RuntimeError("pMCTF HIP engine needs a GPU: the product path has no CPU fallback")
        _lib.hip()
        _lib.rans()
        self.dev = torch.device(device)
        self.num_me_stages = num_me_stages
        self.L = decomp_levels
        sd = {k: v.detach().to("cpu", torch.float32) for k, v in state_dict.items()}
        for k in list(sd.keys()):                       # MaskedConv2d: weight *= mask (layers/layers.py:49-51)
            if k.endswith(".mask"):
                sd[k[:-5] + ".weight"] = sd[k[:-5] + ".weight"] * sd[k]
        self.sd = sd
        self._convs = {}
        self._dw = {}
        self._lin = {}
        self._dev_tab = {}      # CDF tables on the device (decoder)
        self._llw = {}          # packed masked weights of the sequential LL network, per coder (decoder)
        self._bitparm = {}      # per-channel constants of the factorized prior (estimate mode)
        # entropy tables: name -> (cdf int32 [R,C], sizes int32 [R], offsets int32 [R]) host arrays
        self.tables = {"gauss": tuple(np.ascontiguousarray(a, dtype=np.int32) for a in gaussian_tables["cdf_info"])}
        for i, t in enumerate(bit_est_tables):
            self.tables[f"z{i}"] = tuple(np.ascontiguousarray(a, dtype=np.int32) for a in t)
        self.lmin = gaussian_tables["log_scale_min"]
        self.lstep = gaussian_tables["log_scale_step"]
        self.coder = RangeCoderPool(coder_threads)
        self.keep_streams = False       # tests: keep bytes + symbol traces of each stream
        # luma and chroma of a pair are independent once mv_hat exists: they are coded on two side streams so that
        # the small-plane kernels of one fill the tails of the other (per-stream LSTM state keeps them apart)
        self.side_streams = [torch.cuda.Stream(device=self.dev), torch.cuda.Stream(device=self.dev)]
        self.multi_stream = os.environ.get("PMCTF_MULTI_STREAM", "0") == "1"   # luma/chroma on two streams
        self.multi_stream_max_pairs = int(os.environ.get("PMCTF_MULTI_STREAM_MAX_PAIRS", "2"))
        # motion chain of a stage on its own stream, overlapping the previous stage's coding (pMCTF.encode_stage_pairs)
        self.motion_overlap = os.environ.get("PMCTF_MOTION_OVERLAP", "1") != "0"
        self.motion_stream = torch.cuda.Stream(device=self.dev)
        self.lt_event, self.lt_tensor = None, None       # event behind the last batched luma lifting + that L_t tensor
        self.lt_stream = None                            # ... and the stream it was recorded on
        self.pu_fused = os.environ.get("PMCTF_PU_FUSED", "1") != "0"     # one launch per PredictUpdate + lifting step
        self.pu_fused_max_px = int(os.environ.get("PMCTF_PU_FUSED_MAX_PX", "400000"))
        self.post_process_max_px = 8 * 1152 * 1920      # pixels per post-processing launch group (4.5 GB per 64-ch map)
        # encode_one_stage replays captured launch plans (HIP graphs, pMCTF.hip.pair_plan) from the second pair of a
        # configuration on: luma and chroma coders on two streams, no per-launch host work
        self.use_graphs = os.environ.get("PMCTF_GRAPHS", "1") != "0"
        self._dec_pool = None
        self.pair_plans = {}
        self._plan_ctx = {}             # per host thread: graph memory pools and streams of its launch plans
        self.plan_timing = None         # a list: launch plans append (start, motion, luma, chroma, luma syn, chroma syn) events
        self._ref_tls = threading.local()   # .planes: planes per REFERENCE tensor while a stacked batch is being coded
        self.gate = CaptureGate.of(self.dev)
        self.host_threads = set()
        # launch-shape options the plans' convolutions carry per launch (see pair_plan.PairPlan.__init__)
        self.plan_launch_opts = ops.ConvLaunchOpts(int(os.environ.get("PMCTF_PLAN_SPLIT", "0")),
                                                   int(os.environ.get("PMCTF_PLAN_MSPLIT_PX", "40000")))
        from . import pair_plan
        pair_plan.RESOURCES.drain()     # a safe point: what dropped models left behind goes now
        self.syn_after_analysis = os.environ.get("PMCTF_SYN_AFTER_ANALYSIS", "1") != "0"
        self.stats = {"enqueue_s": 0.0, "gpu_done_s": 0.0, "pair_s": 0.0, "pairs": 0}
        self.profile_host = False

    def plan_context(self):
        """Graph memory pools (main + luma chain, chroma chain) and the two coder streams of the CALLING host thread's
        launch plans.  Plans of one thread run one after the other and may share scratch memory; plans of different
        threads (bench.py --inflight) replay concurrently and must not.  luma carries the longest symbol stream: its
        coder gets the high-priority queue so that the stream reaches the host range coder early (measured: 5.41 vs
        5.38 frames/s on the 1080p GOP-16 encode)."""
        tid = threading.get_ident()
        ctx = self._plan_ctx.get(tid)
        if ctx is None:
            prio = int(os.environ.get("PMCTF_LUMA_PRIORITY", "-1"))
            ctx = self._plan_ctx[tid] = {
                "pools": (torch.cuda.graph_pool_handle(), torch.cuda.graph_pool_handle()),
                "streams": (torch.cuda.Stream(device=self.dev, priority=prio),
                            torch.cuda.Stream(device=self.dev, priority=int(os.environ.get("PMCTF_CHROMA_PRIORITY", "0")))),
                "capture": torch.cuda.Stream(device=self.dev)}
            from . import pair_plan
            pair_plan.RESOURCES.adopt(self, dict(ctx))       # the streams outlive the engine until the next safe point
        return ctx

    def release(self):
        """drop the captured launch plans (their graphs hold device memory pools) — called when the model replaces the
        engine or goes away.  May run at ANY moment (a finaliser): it only lets go of the plans; their device objects
        are destroyed by pair_plan.RESOURCES at the next point where no capture can be open — here, if that is now."""
        from . import pair_plan
        self.pair_plans.clear()
        self._plan_ctx.clear()
        pair_plan.RESOURCES.drain()

    # ------------------------------------------------------------------ packed layers
    SIGNAL_PATH = ("optic_flow.", "mv_", "temporal_filtering.")

    def sum_rule(self, p, weight, stride=1):
        """Summation rule of the convolution with parameter key p (include/pmctf_hip.h PMCTF_SUM_*).
        KH*KW > 1 layers of the SIGNAL path — motion estimation, motion codec, temporal lifting, the spatial lifting DWT
        and its inverse: everything a coefficient value or the motion field is computed by — and of the post-processing
        CNN (64 channels: the rule costs its kernels nothing, and it is what the reconstruction's PSNR sees) add their
        products the way ATen's CPU convolution does (per 16-channel block from zero, block sums in turn, bias after the
        first): measured, the last bits of exactly these layers decide the symbols that differed from the reference's
        (profiles/round4_flip_attribution.md).  1x1 layers follow the chain from the bias, which IS ATen's order — except
        where ATen blocks a 1x1 layer's reduction or leaves oneDNN (aten_rules).
        The entropy-parameter networks follow ATen's order like the signal path; with precision "f32-chain" (and the
        reduced-precision profiles) they keep the single chain from the bias (their last bits move a CDF row now and then,
        not a coefficient)."""
        cout, cin, kh, kw = (int(v) for v in weight.shape)
        if not (self.aten_all or p.startswith(self.SIGNAL_PATH) or ".wavelet_transform." in p or ".dequantModule." in p):
            return ops.SUM_CHAIN
        tls = self._ref_tls
        if kh * kw > 1:
            if self.aten_all and stride == 1 and ((1 < cin <= 4 and cout > 1) or (cin == 1 and cout == 1 and kh == 3 and kw == 3)):
                # small-cin layers on small planes: ATen's im2col + sgemm order where it leaves oneDNN (aten_rules)
                return lambda n, h, w: aten_rules.conv_kxk_sum_rule(cin, cout, kh, kw, getattr(tls, "planes", None) or n, h, w)
            return ops.SUM_BLOCKS
        if stride == 1 and (cin >= 112 or self.aten_all):
            # a 1x1 layer: ATen's jit_1x1 kernel cuts the reduction of wide layers into blocks on some plane sizes
            # (e.g. 256 -> 64 at 576x960: blocks of 96 channels); the rule follows from the shape of the REFERENCE's call
            # (its tensors hold one luma or two chroma planes, whatever batch this build has stacked)
            return lambda n, h, w: aten_rules.conv1x1_sum_rule(cin, cout, getattr(tls, "planes", None) or n, h, w)
        return ops.SUM_CHAIN

    @contextlib.contextmanager
    def reference_planes(self, n):
        """inside the block the calling thread codes a batch of stacked reference tensors of n planes each (the
        stage-batched schedule): shape-dependent choices the reference makes per tensor follow n, not the stacked batch"""
        prev = getattr(self._ref_tls, "planes", None)
        self._ref_tls.planes = n
        try:
            yield
        finally:
            self._ref_tls.planes = prev

    @staticmethod
    def lift_skip_rule(ref_planes, H, W):
        """Rule of a lifting step's 3x1 filter on planes of H x W (one input channel = one block).  ATen evaluates the
        layer through oneDNN (bias last) unless its `use_mkldnn` test fails — a reference tensor of ONE plane whose
        reflect-padded input has at most 20 480 elements — in which case im2col + gemv starts from the bias.
        ref_planes: planes in the REFERENCE's tensor (1 luma, 2 chroma), whatever batch this build has stacked."""
        return ops.SUM_CHAIN if ref_planes == 1 and (H + 2) * W <= 20480 else ops.SUM_BLOCKS

    def conv(self, p, stride=1, padding=0):
        key = (p, stride, padding)
        c = self._convs.get(key)
        if c is None:
            w = self.sd[p + ".weight"]
            c = ops.Conv2d(w, self.sd.get(p + ".bias"), stride, (padding, padding), self.dev,
                           split=self.nsplit, rule=self.sum_rule(p, w, stride))
            torch.cuda.synchronize(self.dev)      # packed weights are used from several streams later
            self._convs[key] = c
        return c

    def dwconv(self, p):
        c = self._dw.get(p)
        if c is None:
            c = ops.DepthwiseConv2d(self.sd[p + ".weight"], self.sd.get(p + ".bias"), self.dev)
            torch.cuda.synchronize(self.dev)
            self._dw[p] = c
        return c

    def lin_tables(self, H, W):
        """the cached linspace(-1,1,.) grids of video_net.py:33-40 (built once per shape, on the host)"""
        t = self._lin.get((H, W))
        if t is None:
            t = (torch.linspace(-1.0, 1.0, W).to(self.dev), torch.linspace(-1.0, 1.0, H).to(self.dev))
            torch.cuda.synchronize(self.dev)
            self._lin[(H, W)] = t
        return t

    def warp(self, im, flow, sign=1.0):
        lx, ly = self.lin_tables(im.shape[2], im.shape[3])
        return ops.flow_warp(im, flow, lx, ly, sign)

    # ------------------------------------------------------------------ a6 PredictUpdate (lifting_1d.py:36-49)
    def predict_update(self, p, x):
        """x: plane (N,1,H,W) -> plane"""
        N, _, H, W = x.shape
        xin = x.view(N, H, W, 1)
        c1, t = ops.conv3x3_cin1_dual(self.conv(p + ".conv1", 1, 1), xin, ACT_TANH)
        t = self.conv(p + ".conv2", 1, 1)(t, act=ACT_TANH)
        t = self.conv(p + ".conv3", 1, 1)(t, res1=c1)
        out = self.conv(p + ".conv4", 1, 1)(t)
        return out.view(N, 1, H, W)

    def use_pu_fused(self, x):
        """The one-launch PredictUpdate (pu_fused.hip) wins wherever the chain of eight launches is latency-bound
        (tools/bench_pu.py: 23 vs 66-103 us on the level-2/3 planes, 79 vs 89-101 us at 288x960); on the largest planes
        its halo re-computation (1.5x the matrix work, 3.8 tanh per output value) costs more than the HBM traffic it
        saves, so those keep the separate launches.  Same bits either way."""
        return self.pu_fused and x.numel() <= self.pu_fused_max_px

    def pu_convs(self, p):
        return tuple(self.conv(f"{p}.conv{k}", 1, 1) for k in (1, 2, 3, 4))

    def predict_filter(self, stage, x):
        """wavelet_transform_temporal_mctf.py:27-35: (x + 0.1*P(x)) * (1/sqrt2)"""
        p = f"temporal_filtering.{stage}.P_t"
        if self.use_pu_fused(x):
            return ops.predict_update_fused(x, None, self.pu_convs(p), 0, c=1 / math.sqrt(2))
        pu = self.predict_update(p, x)
        return ew(EW_ADD_MULS_MULS, x, pu, 0.1, 1 / math.sqrt(2))

    def update_filter(self, stage, x):
        p = f"temporal_filtering.{stage}.U_t"
        if self.use_pu_fused(x):
            return ops.predict_update_fused(x, None, self.pu_convs(p), 0, c=0.5)
        pu = self.predict_update(p, x)
        return ew(EW_ADD_MULS_MULS, x, pu, 0.1, 0.5)

    # ------------------------------------------------------------------ a5/a7 (pMCTF_L.py:297-330)
    def forward_MCTF(self, ref, cur, mv_hat, stage_idx=0):
        me = min(self.num_me_stages - 1, stage_idx)
        pred = self.predict_filter(me, self.warp(ref, mv_hat))
        H_t = ew(EW_SUB, cur, pred)
        inv = self.update_filter(me, self.warp(H_t, mv_hat, -1.0))
        L_t = ew(EW_ADD, ref, inv)
        return L_t, H_t, pred, inv

    def inverse_MCTF(self, L_t, H_t, mv_hat, downscale=False, stage_idx=0):
        me = min(self.num_me_stages - 1, stage_idx)
        if downscale:
            mv_hat = ops.bilinear_down2(mv_hat, 2.0)
        inv = self.update_filter(me, self.warp(H_t, mv_hat, -1.0))
        ref = ew(EW_SUB, L_t, inv)
        pred = self.predict_filter(me, self.warp(ref, mv_hat))
        cur = ew(EW_ADD, H_t, pred)
        return ref, cur

    # ------------------------------------------------------------------ a3 SpyNet (video_net.py:93-121)
    def spynet(self, cur_y, ref_y, levels=6, me_downsample=1):
        im1 = [ew(EW_DIVS, cur_y, alpha=255.0)]
        im2 = [ew(EW_DIVS, ref_y, alpha=255.0)]
        if me_downsample > 1:       # the planes are scaled to 0..1 first, then reduced (pMCTF_L.py:452-458)
            im1 = [ops.bilinear_down2(im1[0], 1.0, me_downsample)]
            im2 = [ops.bilinear_down2(im2[0], 1.0, me_downsample)]
        for l in range(levels - 1):
            im1.append(ops.avgpool2(im1[l]))
            im2.append(ops.avgpool2(im2[l]))
        hs, ws = im2[levels - 1].shape[2] // 2, im2[levels - 1].shape[3] // 2
        flow = torch.zeros((1, 2, hs, ws), dtype=torch.float32, device=self.dev)
        for level in range(levels):
            flow_up = ops.bilinear_up2(flow, 2.0)
            i = levels - 1 - level
            warped = self.warp(im2[i], flow_up)
            x = ops.spynet_pack8(im1[i], warped, flow_up)
            p = f"optic_flow.moduleBasic.{level}"
            for k in (1, 2, 3, 4):
                x = self.conv(f"{p}.conv{k}", 1, 3)(x, act=ACT_RELU)
            y5 = self.conv(f"{p}.conv5", 1, 3)(x)
            flow = ew(EW_ADD, flow_up, as_nchw(y5))
        return flow

    # ------------------------------------------------------------------ video/layers.py blocks (NHWC)
    def res_block_stride(self, p, x, stride=2):
        out = self.conv(p + ".conv1", stride, 1)(x, act=ACT_LEAKY, slope=0.01)
        identity = self.conv(p + ".downsample", stride, 0)(x) if stride != 1 else x
        return self.conv(p + ".conv2", 1, 1)(out, act=ACT_LEAKY, slope=0.1, res1=identity)

    def res_block_up(self, p, x):
        out = ops.pixel_shuffle2(self.conv(p + ".subpel_conv.0")(x), ACT_LEAKY, 0.01)
        identity = ops.pixel_shuffle2(self.conv(p + ".upsample.0")(x))
        return self.conv(p + ".conv", 1, 1)(out, act=ACT_LEAKY, slope=0.1, res1=identity)

    def depth_conv(self, p, x):
        identity = self.conv(p + ".adaptor")(x) if (p + ".adaptor.weight") in self.sd else x
        t = self.conv(p + ".conv1.0")(x, act=ACT_LEAKY, slope=0.01)
        t = self.dwconv(p + ".depth_conv")(t)
        return self.conv(p + ".conv2")(t, res1=identity)

    def depth_conv_block(self, p, x):
        x = self.depth_conv(p + ".block.0", x)
        t = self.conv(p + ".block.1.conv.0")(x, act=ACT_LEAKY, slope=0.1)
        return self.conv(p + ".block.1.conv.2")(t, act=ACT_LEAKY, slope=0.1, res1=x)

    def depth_conv_block4(self, p, x):
        x = self.depth_conv(p + ".block.0", x)
        t = ops.ffn3_mix(self.conv(p + ".block.1.conv")(x))
        return self.conv(p + ".block.1.conv_out")(t, res1=x)

    def cat_channels(self, a, b):
        N, H, W, Ca = a.shape
        Cb = b.shape[3]
        out = ops.empty_nhwc(N, H, W, Ca + Cb, self.dev)
        ew(EW_COPY, as_nchw(a), out=as_nchw(out)[:, :Ca])
        ew(EW_COPY, as_nchw(b), out=as_nchw(out)[:, Ca:])
        return out

    # ------------------------------------------------------------------ MV codec (video_net.py:124-191)
    def mv_enc(self, s, est_mv, context, q_enc):
        p = f"mv_encoder.{s}"
        N, _, H, W = est_mv.shape
        x = ops.empty_nhwc(N, H, W, 2, self.dev)
        ew(EW_COPY, est_mv, out=as_nchw(x))
        out = self.res_block_stride(p + ".enc_1.0", x)
        out = self.depth_conv_block(p + ".enc_1.1", out)
        out = as_nhwc_ew(EW_MULS, out, alpha=q_enc)
        out = self.res_block_stride(p + ".enc_2", out)
        if context is None:
            out = self.depth_conv_block(p + ".adaptor_0", out)
        else:
            out = self.depth_conv_block(p + ".adaptor_1", self.cat_channels(out, context))
        out = self.res_block_stride(p + ".enc_3.0", out)
        out = self.depth_conv_block(p + ".enc_3.1", out)
        return self.conv(p + ".enc_3.2", 2, 1)(out)

    def mv_dec(self, s, y_hat, q_dec):
        p = f"mv_decoder.{s}"
        f = self.depth_conv_block(p + ".dec_1.0", y_hat)
        f = self.res_block_up(p + ".dec_1.1", f)
        f = self.depth_conv_block(p + ".dec_1.2", f)
        f = self.res_block_up(p + ".dec_1.3", f)
        feature = self.depth_conv_block(p + ".dec_1.4", f)
        out = self.res_block_up(p + ".dec_2", feature)
        out = as_nhwc_ew(EW_MULS, out, alpha=q_dec)
        out = self.depth_conv_block(p + ".dec_3.0", out)
        mv = ops.pixel_shuffle2(self.conv(p + ".dec_3.1.0")(out))       # (1,H,W,2)
        N, H, W, _ = mv.shape
        mv_planar = ops.empty_planar(N, 2, H, W, self.dev)
        ew(EW_COPY, as_nchw(mv), out=mv_planar)
        return mv_planar, feature

    def mv_hyper_enc(self, s, x):
        p = f"mv_hyper_prior_encoder.{s}"
        x = self.depth_conv_block4(p + ".0", x)
        x = self.conv(p + ".1", 2, 1)(x, act=ACT_LEAKY, slope=0.01)
        return self.conv(p + ".3", 2, 1)(x)

    def mv_hyper_dec(self, s, x):
        p = f"mv_hyper_prior_decoder.{s}"
        x = self.res_block_up(p + ".0", x)
        x = self.res_block_up(p + ".1", x)
        return self.depth_conv_block4(p + ".2", x)

    def mv_prior_param_decoder(self, z_hat, ref_mv_y, s):
        params = self.mv_hyper_dec(s, z_hat)
        if ref_mv_y is None:
            params = self.depth_conv_block(f"mv_y_prior_fusion_adaptor_0.{s}", params)
        else:
            params = self.depth_conv_block(f"mv_y_prior_fusion_adaptor_1.{s}", self.cat_channels(params, ref_mv_y))
        params = self.depth_conv_block(f"mv_y_prior_fusion.{s}.0", params)
        return self.depth_conv_block(f"mv_y_prior_fusion.{s}.1", params)

    def get_mv_y_q(self, q_index, s, inference=True):
        """pMCTF_L.py:221-231; the estimate-mode forward does not round the steps to two decimals"""
        enc = get_curr_q(self.sd[f"mv_y_q_scale_enc.{s}"], q_index)
        dec = get_curr_q(self.sd[f"mv_y_q_scale_dec.{s}"], q_index)
        if not inference:
            return float(enc), float(dec)
        return get_rounded_q(enc.numpy())[0], get_rounded_q(dec.numpy())[0]

    def to_nhwc_input(self, t):
        """dpb tensors come back from the caller as logical NCHW; accept channels-last or planar storage."""
        if t is None:
            return None
        v = t.permute(0, 2, 3, 1)
        if v.is_contiguous():
            return v
        N, Cc, H, W = t.shape
        out = ops.empty_nhwc(N, H, W, Cc, self.dev)
        ew(EW_COPY, t, out=as_nchw(out))
        return out

    # ------------------------------------------------------------------ a2 compress_mv (pMCTF_L.py:448-495)
    def bitparm_consts(self, s):
        """per-channel constants of the factorized prior (entropy_models.py:72-77), evaluated on the host like the
        other scalar parameters: rows softplus(h) f1..f4, b f1..f4, tanh(a) f1..f3"""
        c = self._bitparm.get(s)
        if c is None:
            import torch.nn.functional as F
            g = lambda i, n: self.sd[f"mv_bit_est.{s}.f{i}.{n}"].reshape(-1)
            rows = [F.softplus(g(i, "h")) for i in (1, 2, 3, 4)] + [g(i, "b") for i in (1, 2, 3, 4)] + \
                   [torch.tanh(g(i, "a")) for i in (1, 2, 3)]
            c = torch.stack(rows).contiguous().to(self.dev)
            torch.cuda.synchronize(self.dev)
            self._bitparm[s] = c
        return c

    def compress_mv(self, ref_y, cur_y, dpb, stage_idx=0, q_index=0, estimate=False, me_downsample=1):
        """estimate=True: compute_and_code_motion (pMCTF_L.py:244-292): same networks, unrounded quantisation steps,
        bit estimates (two device float64: y, z) instead of a symbol stream.
        dpb may be a zero-argument callable: it is evaluated after the motion ESTIMATION (which needs no context), so
        a caller that receives the context from another GPU overlaps the wait with SpyNet (pmctf_dist relay)."""
        est_mv = self.spynet(cur_y, ref_y, me_downsample=me_downsample)
        if callable(dpb):
            dpb = dpb()
        return self.motion_code(est_mv, dpb, stage_idx, q_index, estimate, me_downsample)

    def motion_estimate(self, ref_y, cur_y, me_downsample=1):
        """the motion ESTIMATION of compress_mv (pMCTF_L.py:452-460): optic_flow(cur, ref)"""
        return self.spynet(cur_y, ref_y, me_downsample=me_downsample)

    def mv_symbol_count(self, height, width, me_downsample=1):
        """symbols of one motion stream (z + four y parts) for a luma plane of height x width"""
        h, w = height // me_downsample, width // me_downsample
        return (h // 64) * (w // 64) * 64 + 4 * 16 * (h // 16) * (w // 16)

    def motion_code(self, est_mv, dpb, stage_idx=0, q_index=0, estimate=False, me_downsample=1):
        """the motion CODEC of compress_mv (pMCTF_L.py:462-495): everything after the motion estimation"""
        s = min(self.num_me_stages - 1, stage_idx)
        q_enc, q_dec = self.get_mv_y_q(q_index, s, inference=not estimate)
        mv_y = self.mv_enc(s, est_mv, self.to_nhwc_input(dpb["mv_feature"]), q_enc)
        mv_z = self.mv_hyper_enc(s, mv_y)
        _, hy, wy, _ = mv_y.shape
        _, hz, wz, cz = mv_z.shape
        if estimate:
            stream = None
            bits_y, bits_z = BitSink(1, self.dev), BitSink(1, self.dev)
            z_hat = ops.z_estimate(mv_z, self.bitparm_consts(s), bits_z.bits)
        else:
            stream = SymbolStream(hz * wz * cz + 4 * 16 * hy * wy, self.dev)
            z_hat = ops.z_symbols(mv_z, stream.sym, stream.idx, stream.take(hz * wz * cz, f"z{s}"))
        common = self.mv_prior_param_decoder(z_hat, self.to_nhwc_input(dpb["ref_mv_y"]), s)
        so_far = torch.empty_like(mv_y)
        sp = None
        for t in range(4):
            if t > 0:
                x = self.conv(f"mv_y_spatial_prior_adaptor_{t}.{s}")(self.cat_channels(so_far, common))
                for i in range(3):
                    x = self.depth_conv_block(f"mv_y_spatial_prior.{s}.{i}", x)
                sp = x
            if estimate:
                ops.mv_fourpart_estimate(mv_y, common, sp, so_far, t, bits_y.bits)
            else:
                ops.mv_fourpart_step(mv_y, common, sp, so_far, stream.sym, stream.idx,
                                     stream.take(16 * hy * wy, "gauss"), t, self.lmin, self.lstep)
        mv_y_hat = ops.mv_dequant(so_far, common)
        mv_hat, mv_feature = self.mv_dec(s, mv_y_hat, q_dec)
        if me_downsample > 1:       # pMCTF_L.py:274-275,475-476
            mv_hat = ops.bilinear_up2(mv_hat, float(me_downsample), me_downsample)
        out = {"stream": stream, "mv_hat": mv_hat, "mv_feature": mv_feature, "mv_y_hat": mv_y_hat,
               "est_mv": est_mv, "mv_y": mv_y, "z_hat": z_hat, "common": common}
        if estimate:
            out["bits_y"], out["bits_z"] = bits_y.bits, bits_z.bits
        return out

    # ------------------------------------------------------------------ a10 learned lifting DWT
    def lift_branch(self, wt, conv_name, pu_name, x, skip_rule):
        """skip + 0.1 * 256 * PU(skip/256)   (lifting_1d.py:105-110)"""
        w = self.sd[f"{wt}.{conv_name}.weight"].reshape(-1)
        b = self.sd[f"{wt}.{conv_name}.bias"].reshape(-1)
        skip = ops.lift_skip3(x, w.tolist(), float(b[0]), skip_rule)
        pu = self.predict_update(f"{wt}.{pu_name}", ew(EW_DIVS, skip, alpha=256.0))
        return ew(EW_ADD_MULS2, skip, pu, 256.0, 0.1)

    def lift_step(self, wt, conv_name, pu_name, src, other, sign, ref_planes=None):
        """other + sign * branch(src): one lifting step (lifting_1d.py:105-118 forward, :150-163 backward).
        ref_planes: planes per tensor in the reference (None: this tensor's own batch)."""
        if ref_planes is None:
            ref_planes = getattr(self._ref_tls, "planes", None)
        skip_rule = self.lift_skip_rule(src.shape[0] if ref_planes is None else ref_planes, src.shape[2], src.shape[3])
        if self.use_pu_fused(src) and src.shape[2] >= 2:
            w = self.sd[f"{wt}.{conv_name}.weight"].reshape(-1).tolist()
            b = float(self.sd[f"{wt}.{conv_name}.bias"].reshape(-1)[0])
            return ops.predict_update_fused(src, other, self.pu_convs(f"{wt}.{pu_name}"), 1, sign=sign,
                                            lift=(w[0], w[1], w[2], b), skip_rule=skip_rule)
        return ew(EW_ADD if sign > 0 else EW_SUB, other, self.lift_branch(wt, conv_name, pu_name, src, skip_rule))

    def forward_lift(self, wt, x):
        """iWave1D.forward_lift along H (lifting_1d.py:103-145); x plane (N,1,H,W) -> l, h (N,1,H/2,W)"""
        x_e = ew(EW_COPY, x[:, :, ::2, :])
        x_o = ew(EW_COPY, x[:, :, 1::2, :])
        x_o = self.lift_step(wt, "conv_P1", "P_1", x_e, x_o, 1.0)
        x_e = self.lift_step(wt, "conv_U1", "U_1", x_o, x_e, 1.0)
        x_o = self.lift_step(wt, "conv_P2", "P_2", x_e, x_o, 1.0)
        x_e = self.lift_step(wt, "conv_U2", "U_2", x_o, x_e, 1.0)
        return ew(EW_MULS, x_e, alpha=SCALE_L), ew(EW_MULS, x_o, alpha=SCALE_H)

    def backward_lift(self, wt, l, h):
        """iWave1D.backward_lift (lifting_1d.py:147-189)"""
        l = ew(EW_DIVS, l, alpha=SCALE_L)
        h = ew(EW_DIVS, h, alpha=SCALE_H)
        l = self.lift_step(wt, "conv_U2", "U_2", h, l, -1.0)
        h = self.lift_step(wt, "conv_P2", "P_2", l, h, -1.0)
        l = self.lift_step(wt, "conv_U1", "U_1", h, l, -1.0)
        h = self.lift_step(wt, "conv_P1", "P_1", l, h, -1.0)
        N, _, H2, W = l.shape
        x = ops.empty_planar(N, 1, 2 * H2, W, self.dev)
        ew(EW_COPY, l, out=x[:, :, ::2, :])
        ew(EW_COPY, h, out=x[:, :, 1::2, :])
        return x

    def transpose(self, x):
        return ew(EW_COPY, x.permute(0, 1, 3, 2))

    def forward_lift_2d(self, coder, x):
        """LiftingScheme2D.forward_lift_2d (wavelet_transform.py:25-42)"""
        wt = f"{coder}.wavelet_transform.lift_h"
        l, h = self.forward_lift(wt, x)
        ll, lh = self.forward_lift(wt, self.transpose(l))
        hl, hh = self.forward_lift(wt, self.transpose(h))
        return {k: self.transpose(v) for k, v in (("ll", ll), ("lh", lh), ("hl", hl), ("hh", hh))}

    def backward_lift_2d(self, coder, sb):
        wt = f"{coder}.wavelet_transform.lift_h"
        l = self.transpose(self.backward_lift(wt, self.transpose(sb["ll"]), self.transpose(sb["lh"])))
        h = self.transpose(self.backward_lift(wt, self.transpose(sb["hl"]), self.transpose(sb["hh"])))
        return self.backward_lift(wt, l, h)

    # ------------------------------------------------------------------ a11 LL parameters (context_fusion.py:100-128)
    def context_fusion_ll(self, coder, ll):
        p = f"{coder}.context_fusion.{self.L - 1}.ll"
        N, _, H, W = ll.shape
        x = self.conv(p + ".maskedConv1", 1, 1)(ll.view(N, H, W, 1))
        conv1 = x
        for i in range(2):
            q = f"{p}.residualBlocks.{i}"
            o = self.conv(q + ".conv1", 1, 1)(x, act=ACT_LEAKY, slope=0.2)
            if i == 1:
                x = self.conv(q + ".conv2", 1, 1)(o, res1=x, res2=conv1)      # (o + x) + conv1
            else:
                x = self.conv(q + ".conv2", 1, 1)(o, res1=x)
        x = self.conv(p + ".maskedConv2", 1, 1)(x, act=ACT_LEAKY, slope=0.2)
        x = self.conv(p + ".convs.0")(x, act=ACT_LEAKY, slope=0.2)
        x = self.conv(p + ".convs.1")(x, act=ACT_LEAKY, slope=0.2)
        return self.conv(p + ".convs.2")(x)

    # ------------------------------------------------------------------ a12 four-step context fusion
    def context_residual(self, p, x, res2=None):
        o = self.conv(p + ".conv1", 1, 1)(x, act=ACT_LEAKY, slope=0.2)
        return self.conv(p + ".conv2", 1, 1)(o, res1=x, res2=res2)

    def fusion_compress(self, p, x, ctx, prev, stream):
        """ContextFusionFourStep.forward(write=True) (context_fusion_4step.py:139-191).
        x: plane (N,1,h,w); ctx: NHWC (N,h,w,1); prev: plane of the previous level's subband or None."""
        N, _, H, W = x.shape
        if prev is not None:
            up = ops.nearest_up2(prev.view(N, H // 2, W // 2, 1))
            prevc = self.conv(p + ".lower_level_subband.1", 1, 1)(up)
            ctx = self.cat_channels(ctx, prevc)
        c = self.conv(p + ".conv1_context", 1, 1)(ctx)
        c = self.context_residual(p + ".y_hierarchical_prior_enc.0", c)
        c = self.context_residual(p + ".y_hierarchical_prior_enc.1", c)
        params = self.depth_conv_block(p + ".y_hierarchical_prior_out", c)
        so_far = torch.empty_like(x)
        n = N * H * W
        self._fourstep(stream, x, params, so_far, 0, n)
        for step in (1, 2, 3):
            t = self.conv(f"{p}.y_spatial_prior_{step}.0", 1, 1)(so_far.view(N, H, W, 1))
            t = self.context_residual(f"{p}.y_spatial_prior_{step}.1", t, res2=c)      # (.. + x) + context
            t = self.context_residual(f"{p}.y_spatial_prior_{step}_out.0", t)
            if H % 2 == 0 and W % 2 == 0:
                # this step only consumes its parameters where mask `step` is set: evaluate the last 3x3 conv of the
                # residual block, its skip and the 1x1 head at that quarter of the positions (identical sums)
                q = f"{p}.y_spatial_prior_{step}_out.1"
                o = self.conv(q + ".conv1", 1, 1)(t, act=ACT_LEAKY, slope=0.2)
                py, px = step >> 1, step & 1
                tq = ops.empty_nhwc(N, H // 2, W // 2, t.shape[3], self.dev)
                ew(EW_COPY, as_nchw(t)[:, :, py::2, px::2], out=as_nchw(tq))
                tq = ops.conv_at_class(self.conv(q + ".conv2", 1, 1), o, step, res1=tq)
                params = self.conv(f"{p}.y_spatial_prior_{step}_out.2")(tq, rule_hw=(H, W))
            else:
                t = self.context_residual(f"{p}.y_spatial_prior_{step}_out.1", t)
                params = self.conv(f"{p}.y_spatial_prior_{step}_out.2")(t)
            self._fourstep(stream, x, params, so_far, step, n)
        return so_far

    def _fourstep(self, sink, x, params, so_far, step, n):
        if isinstance(sink, BitSink):
            ops.fourstep_estimate(x, params, so_far, step, sink.bits)
        else:
            ops.fourstep_quant(x, params, so_far, sink.sym, sink.idx, sink.take(n, "gauss"), step, self.lmin, self.lstep)

    # ------------------------------------------------------------------ a13 conv-LSTM context (long_context.py)
    def lstm(self, p, x, state):
        a = self.conv(p + ".conv_in", 1, 1)(x)
        xh = self.conv(p + ".conv_hidden", 1, 1)(state[0], res1=a)
        hid, cell = ops.lstm_gates(xh, state[1], getattr(self._ref_tls, "planes", None),
                                   self.aten_threads if self.aten_all else 0)
        return [hid, cell]

    def ctx_init(self, N, H, W):
        z = lambda c: torch.zeros((N, H, W, c), dtype=torch.float32, device=self.dev)
        # init_sequential quirk: LSTM3's cell state starts with 1 channel (long_context.py:163-164)
        return {"l1": [z(32), z(32)], "l2": [z(32), z(32)], "l3": [z(3), z(1)]}

    def ctx_upsample(self, p, x):
        return self.conv(p + ".conv", 1, 1)(ops.nearest_up2(x))

    def ctx_forward_one_subband(self, coder, st, subband, name, lvl):
        p = f"{coder}.context_prediction"
        N, _, H, W = subband.shape
        st["l1"] = self.lstm(p + ".LSTM1", subband.view(N, H, W, 1), st["l1"])
        st["l2"] = self.lstm(p + ".LSTM2", st["l1"][0], st["l2"])
        st["l3"] = self.lstm(p + ".LSTM3", st["l2"][0], st["l3"])
        if name == "hh" and lvl > 0:
            for key, nm in (("l1", "1"), ("l2", "2"), ("l3", "3")):
                st[key][0] = self.ctx_upsample(f"{p}.deconv_h{nm}.{lvl - 1}", st[key][0])
                st[key][1] = self.ctx_upsample(f"{p}.deconv_c{nm}.{lvl - 1}", st[key][1])
        return st["l3"][0]

    # ------------------------------------------------------------------ a14 PostProcess (postprocessing.py:35-44)
    def post_process(self, coder, x, in_div=1.0, out_mul=1.0):
        """returns (x/in_div + net(x/in_div)) * out_mul"""
        p = f"{coder}.dequantModule"
        N, _, H, W = x.shape
        # large batches (stage s of several GOPs at once): 64-channel full-resolution maps are the biggest tensors of
        # the path — bound them by working through the planes in groups (planes are independent; same arithmetic)
        cap = self.post_process_max_px
        if N > 1 and N * H * W > cap:
            per = max(1, cap // (H * W))
            return torch.cat([self.post_process(coder, x[i:i + per], in_div, out_mul) for i in range(0, N, per)], dim=0)
        xs = ew(EW_DIVS, x, alpha=in_div) if in_div != 1.0 else x
        tmp = self.conv(p + ".conv1", 1, 1)(xs.view(N, H, W, 1))
        conv1 = tmp
        for i in range(6):
            q = f"{p}.resBlocks.{i}"
            o = self.conv(q + ".conv1", 1, 1)(tmp, act=ACT_LEAKY, slope=0.2)
            tmp = self.conv(q + ".conv2", 1, 1)(o, res1=tmp)
        tmp = self.conv(p + ".conv2", 1, 1)(tmp, res1=conv1)
        tmp = self.conv(p + ".conv3", 1, 1)(tmp)
        return ew(EW_ADD_MULS_MULS, xs, tmp.view(N, 1, H, W), 1.0, out_mul)

    # ------------------------------------------------------------------ a9 pWave.compress (pWave.py:381-463)
    def pwave_symbol_count(self, N, H, W):
        n = N * (H >> self.L) * (W >> self.L)
        for lvl in range(self.L):
            n += 3 * 4 * N * (H >> (lvl + 1)) * (W >> (lvl + 1))
        return n

    def q_scales(self, coder, q_index, qp_scale=None):
        """quantisation step of a coder's subbands and of its LL subband (pWave.py:383-393,468-478)"""
        if q_index is None:
            q_scale, q_scale_ll = self.sd[f"{coder}.QP"][-1], self.sd[f"{coder}.QP_ll"][-1]
        else:
            q_scale = get_curr_q(self.sd[f"{coder}.QP"], q_index)
            q_scale_ll = get_curr_q(self.sd[f"{coder}.QP_ll"], q_index)
            if qp_scale is not None:
                q_scale = q_scale * qp_scale
                q_scale_ll = q_scale_ll * qp_scale
        return float(q_scale), float(q_scale_ll)

    def pwave_forward(self, coder, x, q_index, qp_scale=None):
        """pWave.forward_one_channel (pWave.py:243-312): the coder's networks with bit ESTIMATES.  Returns a dict with
        x_hat (plane), subbands (quantised), bits (device float64 per plane), sq_err (device float64 scalar)."""
        x_hat, sink, hat = self.pwave_compress(coder, x, q_index, qp_scale, estimate=True)
        sq = torch.zeros(1, dtype=torch.float64, device=self.dev)
        ops.sqdiff_sum(x.contiguous(), x_hat, sq)
        return {"x_hat": x_hat, "subbands": hat, "bits": sink.bits, "sq_err": sq}

    def pwave_compress(self, coder, x, q_index, qp_scale=None, ar_order=False, estimate=False, defer=False,
                       per_push=False):
        """x: plane (N,1,H,W).  Returns (x_hat plane, SymbolStream).  ar_order: LL symbols in the sequential coder's
        order (needed for streams the decoder will read: skip_decoding=False, pWave.py:410-411,531-555).
        estimate=True: returns (x_hat, BitSink, quantised subbands) — see pwave_forward.
        defer=True: returns (synthesis, SymbolStream); synthesis() computes x_hat later.
        per_push=True: the stream keeps one segment per push, so a batch of independently coded planes can be cut into
        one bitstream per plane group (RangeCoderPool.submit_planes)."""
        q_scale, q_scale_ll = self.q_scales(coder, q_index, qp_scale)
        N, _, H, W = x.shape
        clip = 8192.0
        y = {}
        ll = x
        for lvl in range(self.L):
            y[lvl] = self.forward_lift_2d(coder, ll)
            ll = y[lvl]["ll"]
        stream = BitSink(N, self.dev) if estimate else SymbolStream(self.pwave_symbol_count(N, H, W), self.dev,
                                                                    merge=not per_push)
        hat = {lvl: {} for lvl in range(self.L)}
        llq = ew(EW_ROUND_CLAMP_MULS, ll, alpha=q_scale_ll, beta=clip)
        params = self.context_fusion_ll(coder, llq)
        if estimate:        # pWave.py:255-263: the rounded LL itself is kept; bits of the unrounded residual
            ops.ll_estimate(llq, params, stream.bits)
            ll_hat = llq
        else:
            ll_hat = ops.ll_quant(llq, params, stream.sym, stream.idx, stream.take(llq.numel(), "gauss"), self.lmin,
                                  self.lstep, ar_order)
        hat[self.L - 1]["ll"] = ll_hat
        lstm_state = self.ctx_init(N, ll.shape[2], ll.shape[3])
        context = self.ctx_forward_one_subband(coder, lstm_state, ll_hat, "ll", self.L - 1)
        for lvl in range(self.L - 1, -1, -1):
            for sidx, sb in enumerate(("lh", "hl", "hh")):
                h, w = context.shape[1], context.shape[2]
                ctx = ops.empty_nhwc(N, h, w, 1, self.dev)
                ew(EW_COPY, as_nchw(context)[:, sidx:sidx + 1], out=as_nchw(ctx))
                prev = hat[lvl + 1][sb] if lvl < self.L - 1 else None
                s_curr = ew(EW_CLAMP_MULS, y[lvl][sb], alpha=q_scale, beta=clip)
                s_hat = self.fusion_compress(f"{coder}.context_fusion.{lvl}.{sb}", s_curr, ctx, prev, stream)
                hat[lvl][sb] = s_hat
                context = self.ctx_forward_one_subband(coder, lstm_state, s_hat, sb, lvl)
        def synthesis():
            """dequantise, inverse DWT, post-process (pWave.py:446-455): needed for the reconstruction only, not for
            the bitstream, so a caller may run it after the stream has been handed to the range coder"""
            out = None
            rec_ll = ew(EW_DIVS, hat[self.L - 1]["ll"], alpha=q_scale_ll)
            for lvl in range(self.L - 1, -1, -1):
                sbs = {"ll": rec_ll}
                for sb in ("lh", "hl", "hh"):
                    sbs[sb] = ew(EW_DIVS, hat[lvl][sb], alpha=q_scale)
                out = self.backward_lift_2d(coder, sbs)
                rec_ll = out
            return self.post_process(coder, out, 256.0, 256.0)

        if defer:
            return synthesis, stream
        x_hat = synthesis()
        if estimate:
            return x_hat, stream, hat
        return x_hat, stream

    # ------------------------------------------------------------------ estimate-mode stage (pMCTF_L.py:332-379)
    def forward_one_stage(self, ref, cur, q_index, code_lt, dpb, mv_hat=None, stage_idx=0, me_downsample=1):
        """Returns tensors plus a dict `acc` of device float64 accumulators; the caller turns them into the
        reference's bpp / mse scalars with ONE synchronising read."""
        acc = {}
        if mv_hat is not None:
            mv_hat = ops.bilinear_down2(mv_hat, 2.0)
            ref_mv = {"mv_feature": None, "mv_y_hat": None}
        else:
            mv = self.compress_mv(ref[0:1], cur[0:1], dpb, stage_idx=stage_idx, q_index=q_index, estimate=True,
                                  me_downsample=me_downsample)
            mv_hat = mv["mv_hat"]
            ref_mv = mv
            acc["bits_mv_y"], acc["bits_mv_z"] = mv["bits_y"], mv["bits_z"]
        L_t, H_t, pred, inv = self.forward_MCTF(ref, cur, mv_hat, stage_idx)
        qp_scale = get_curr_q(self.sd[f"hp_q_scale.{stage_idx}"], q_index)
        res_H = self.pwave_forward("hp_coder", H_t, q_index, qp_scale)
        acc["bits_H"], acc["sq_H"] = res_H["bits"], res_H["sq_err"]
        acc["sq_me"] = torch.zeros(1, dtype=torch.float64, device=self.dev)
        ops.sqdiff_sum(pred, cur.contiguous(), acc["sq_me"])
        out = {"mv_hat": mv_hat, "ref_mv": ref_mv, "H_t": res_H["x_hat"], "L_t": L_t, "acc": acc}
        if code_lt:
            res_L = self.pwave_forward("lp_coder", L_t, q_index)
            acc["bits_L"], acc["sq_L"] = res_L["bits"], res_L["sq_err"]
            acc["sq_me_inv"] = torch.zeros(1, dtype=torch.float64, device=self.dev)
            ops.sqdiff_sum(inv, ref.contiguous(), acc["sq_me_inv"])
            out["L_t"] = res_L["x_hat"]
        return out

    # ------------------------------------------------------------------ a8 compress_one_stage (pMCTF_L.py:398-420)
    def compress_one_stage(self, ref, cur, code_lt, mv_hat, ischroma, stage_idx=0, q_index=0, ar_order=False,
                           on_stream=None, defer=False, code_h=True):
        """on_stream(kind, SymbolStream): called as soon as a stream's symbols are complete (kind "H" / "L").
        code_h=False: only the temporal lifting and the L coder (the H coder of this plane set runs elsewhere —
        pmctf_dist's split of a pair over ranks; the four spatial coders of a pair are independent given mv_hat,
        pMCTF_L.py:398-420,570-592)."""
        if ischroma:
            mv_hat = ops.bilinear_down2(mv_hat, 2.0)
        L_t, H_t, _, _ = self.forward_MCTF(ref, cur, mv_hat, stage_idx)
        out = {"L_t": L_t, "H_t": H_t, "H_t_hat": None, "H_stream": None, "L_t_hat": None, "L_stream": None}
        H_syn = None
        if code_h:
            qp_scale = get_curr_q(self.sd[f"hp_q_scale.{stage_idx}"], q_index)
            H_syn, out["H_stream"] = self.pwave_compress("hp_coder", H_t, q_index, qp_scale, ar_order, defer=True)
            if on_stream is not None:
                on_stream("H", out["H_stream"])
        L_syn = None
        if code_lt:
            L_syn, out["L_stream"] = self.pwave_compress("lp_coder", L_t, q_index, None, ar_order, defer=True)
            if on_stream is not None:
                on_stream("L", out["L_stream"])

        def finish():
            """the reconstructions (synthesis transforms + post-processing), after every stream of the pair is with
            the range coder: its tail overlaps with this GPU work instead of leaving the GPU idle"""
            if H_syn is not None:
                out["H_t_hat"] = H_syn()
            if L_syn is not None:
                out["L_t_hat"] = L_syn()
            return out
        if defer:
            out["finish"] = finish
            return out
        return finish()


    def compress_stage_batched(self, refs, curs, code_lt, mv_hats, ischroma, stage_idx=0, q_index=0, on_stream=None):
        """compress_one_stage for ALL pairs of a temporal stage at once: the pairs are independent (same coder weights,
        per-plane contexts), so they run as one batch — every kernel launch covers len(refs) x more pixels, which is
        what the latency-bound small subbands need.  refs/curs: per pair a (n,1,H,W) plane tensor (n = 1 luma,
        2 chroma); mv_hats: per pair (1,2,H,W) at luma resolution.  on_stream(kind, SymbolStream, planes_per_pair).
        Per plane the arithmetic is exactly that of compress_one_stage.  Returns dict with batched tensors + finish()."""
        n = refs[0].shape[0]
        ref = torch.cat(refs, dim=0)
        cur = torch.cat(curs, dim=0)
        flows = []
        for mv in mv_hats:
            f = ops.bilinear_down2(mv, 2.0) if ischroma else mv
            flows.extend([f] * n)
        mv_all = torch.cat(flows, dim=0)
        L_t, H_t, _, _ = self.forward_MCTF(ref, cur, mv_all, stage_idx)
        if not ischroma:      # the next stage's motion estimation may start from here (its inputs are slices of L_t)
            ev = torch.cuda.Event()
            ev.record()
            self.lt_event, self.lt_tensor, self.lt_stream = ev, L_t, torch.cuda.current_stream(self.dev).cuda_stream
        qp_scale = get_curr_q(self.sd[f"hp_q_scale.{stage_idx}"], q_index)
        with self.reference_planes(n):
            H_syn, h_stream = self.pwave_compress("hp_coder", H_t, q_index, qp_scale, False, defer=True, per_push=True)
        out = {"L_t": L_t, "H_t": H_t, "H_t_hat": None, "L_t_hat": None}
        if on_stream is not None:
            on_stream("H", h_stream, n)
        L_syn = None
        if code_lt:
            with self.reference_planes(n):
                L_syn, l_stream = self.pwave_compress("lp_coder", L_t, q_index, None, False, defer=True, per_push=True)
            if on_stream is not None:
                on_stream("L", l_stream, n)

        def finish():
            with self.reference_planes(n):
                out["H_t_hat"] = H_syn()
                if L_syn is not None:
                    out["L_t_hat"] = L_syn()
            return out
        out["finish"] = finish
        return out

    # =================================================================================================
    # Decoder: pMCTF.decompress_mv / pWave.decompress on the GPU, entropy decoding on the host
    # (libpmctf_rans.so) except for the sequential LL subband, which is decoded inside one persistent kernel.
    # =================================================================================================
    def _dev_tables(self, name):
        t = self._dev_tab.get(name)
        if t is None:
            cdf, sizes, offsets = self.tables[name]
            t = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(self.dev) for a in (cdf, sizes, offsets))
            torch.cuda.synchronize(self.dev)
            self._dev_tab[name] = t
        return t

    def _ll_weights(self, coder):
        w = self._llw.get(coder)
        if w is None:
            p = f"{coder}.context_fusion.{self.L - 1}.ll"
            g = lambda k: np.ascontiguousarray(self.sd[p + k].numpy(), dtype=np.float32)
            names = [".residualBlocks.0.conv1", ".residualBlocks.0.conv2", ".residualBlocks.1.conv1",
                     ".residualBlocks.1.conv2", ".maskedConv2"]
            wb = [g(n + ".weight") for n in names]
            bb = [g(n + ".bias") for n in names]
            PF = C.POINTER(C.c_float)
            wb_p = (C.c_void_p * 5)(*[a.ctypes.data for a in wb])
            bb_p = (C.c_void_p * 5)(*[a.ctypes.data for a in bb])
            L = _lib.hip()
            out = np.empty(L.pmctf_ll_ar_packed_size(), np.float32)
            keep = [g(".maskedConv1.weight"), g(".maskedConv1.bias"), g(".convs.0.weight"), g(".convs.0.bias"),
                    g(".convs.1.weight"), g(".convs.1.bias"), g(".convs.2.weight"), g(".convs.2.bias")]
            _lib.check(L.pmctf_ll_ar_pack_weights(keep[0].ctypes.data, keep[1].ctypes.data, C.addressof(wb_p),
                                                  C.addressof(bb_p), keep[2].ctypes.data, keep[3].ctypes.data,
                                                  keep[4].ctypes.data, keep[5].ctypes.data, keep[6].ctypes.data,
                                                  keep[7].ctypes.data, out.ctypes.data), "ll_ar_pack_weights")
            w = torch.from_numpy(out).to(self.dev)
            torch.cuda.synchronize(self.dev)
            self._llw[coder] = w
        return w

    def _decode(self, dec, idx_dev, table):
        """device int16 CDF rows -> host range decoder -> device int16 symbols"""
        return torch.from_numpy(dec.decode(idx_dev.cpu().numpy(), table)).to(self.dev)

    def ll_ar_launch(self, coder, dec, N, H, W, stream=None):
        """Enqueue the sequential LL decode (one persistent workgroup) on `stream`; returns a ticket for ll_ar_finish.
        The LL streams of different files are independent, so callers start all of them before waiting for any."""
        if N > 4:
            raise NotImplementedError("the sequential LL decoder handles up to 4 planes per stream (Y / UV / RGB)")
        L = _lib.hip()
        w = self._ll_weights(coder)
        cdf, sizes, offsets = self._dev_tables("gauss")
        x, pos = dec.get_state()
        st = stream if stream is not None else torch.cuda.current_stream(self.dev)
        with torch.cuda.stream(st):
            words = torch.from_numpy(dec.words.copy()).to(self.dev)
            ll = torch.zeros((N, 1, H, W), dtype=torch.float32, device=self.dev)
            scratch = torch.zeros(L.pmctf_ll_ar_scratch_floats(N, H, W), dtype=torch.float32, device=self.dev)
            state = torch.zeros(3, dtype=torch.int64, device=self.dev)
            vp = lambda t: C.c_void_p(t.data_ptr())
            # the rules of the encoder's one-shot network (context_fusion_ll), layer for layer
            p = f"{coder}.context_fusion.{self.L - 1}.ll"
            rules = []
            for name in (".maskedConv2", ".convs.0", ".convs.2"):
                r = self.sum_rule(p + name, self.sd[p + name + ".weight"])
                rules.append(r(N, H, W) if callable(r) else r)
            _lib.check(L.pmctf_ll_ar_decode_rules_f32(vp(w), vp(words), words.numel(), C.c_uint64(x), pos, vp(cdf),
                                                      vp(sizes), vp(offsets), cdf.shape[1], float(self.lmin),
                                                      float(self.lstep), vp(ll), vp(scratch), N, H, W, vp(state),
                                                      rules[0], rules[1], rules[2], C.c_void_p(st.cuda_stream)),
                       "ll_ar_decode")
        return {"dec": dec, "ll": ll, "state": state, "stream": st, "keep": (words, scratch)}

    def ll_ar_finish(self, t):
        t["stream"].synchronize()
        st = t["state"].cpu().numpy()
        if st[2] != 0:
            raise ValueError("LL decode ran past the end of the bitstream")
        t["dec"].set_state(int(np.uint64(st[0])), int(st[1]))
        t["ll"].record_stream(torch.cuda.current_stream(self.dev))
        return t["ll"]

    def ll_ar_decode(self, coder, dec, N, H, W):
        return self.ll_ar_finish(self.ll_ar_launch(coder, dec, N, H, W))

    def fusion_decompress(self, p, ctx, prev, dec, N, H, W):
        """ContextFusionFourStep.decompress (context_fusion_4step.py:196-249)"""
        if prev is not None:
            up = ops.nearest_up2(prev.view(N, H // 2, W // 2, 1))
            prevc = self.conv(p + ".lower_level_subband.1", 1, 1)(up)
            ctx = self.cat_channels(ctx, prevc)
        c = self.conv(p + ".conv1_context", 1, 1)(ctx)
        c = self.context_residual(p + ".y_hierarchical_prior_enc.0", c)
        c = self.context_residual(p + ".y_hierarchical_prior_enc.1", c)
        params = self.depth_conv_block(p + ".y_hierarchical_prior_out", c)
        so_far = torch.empty((N, 1, H, W), dtype=torch.float32, device=self.dev)
        for step in range(4):
            if step > 0:
                t = self.conv(f"{p}.y_spatial_prior_{step}.0", 1, 1)(so_far.view(N, H, W, 1))
                t = self.context_residual(f"{p}.y_spatial_prior_{step}.1", t, res2=c)
                t = self.context_residual(f"{p}.y_spatial_prior_{step}_out.0", t)
                if H % 2 == 0 and W % 2 == 0:
                    q = f"{p}.y_spatial_prior_{step}_out.1"
                    o = self.conv(q + ".conv1", 1, 1)(t, act=ACT_LEAKY, slope=0.2)
                    py, px = step >> 1, step & 1
                    tq = ops.empty_nhwc(N, H // 2, W // 2, t.shape[3], self.dev)
                    ew(EW_COPY, as_nchw(t)[:, :, py::2, px::2], out=as_nchw(tq))
                    tq = ops.conv_at_class(self.conv(q + ".conv2", 1, 1), o, step, res1=tq)
                    params = self.conv(f"{p}.y_spatial_prior_{step}_out.2")(tq, rule_hw=(H, W))
                else:
                    t = self.context_residual(f"{p}.y_spatial_prior_{step}_out.1", t)
                    params = self.conv(f"{p}.y_spatial_prior_{step}_out.2")(t)
            idx = ops.fourstep_indexes(params, N, H, W, step, self.lmin, self.lstep)
            sym = self._decode(dec, idx, "gauss")
            ops.fourstep_dequant(sym, params, so_far, step)
        return so_far

    def pwave_decompress_begin(self, coder, data, padding, q_index, qp_scale=None, stream=None):
        """Parse one bitstream file and start its sequential LL decode; finish with pwave_decompress_end."""
        import struct
        q_scale, q_scale_ll = self.q_scales(coder, q_index, qp_scale)
        height, width, N = struct.unpack(">III", data[:12])
        (n,) = struct.unpack(">I", data[12:16])
        dec = HostDecoder(self.tables, data[16:16 + n])
        new_h = (height + padding - 1) // padding * padding
        new_w = (width + padding - 1) // padding * padding
        sh, sw = new_h >> self.L, new_w >> self.L
        ticket = self.ll_ar_launch(coder, dec, N, sh, sw, stream)
        return {"coder": coder, "dec": dec, "ticket": ticket, "q": (q_scale, q_scale_ll), "N": N,
                "shape": (new_h, new_w, sh, sw)}

    def pwave_decompress(self, coder, data, padding, q_index, qp_scale=None):
        """pWave.decompress (pWave.py:467-529) from the bytes of one bitstream file -> x_hat plane (N,1,Hp,Wp)"""
        return self.pwave_decompress_end(self.pwave_decompress_begin(coder, data, padding, q_index, qp_scale))

    def pwave_decompress_many(self, jobs):
        """jobs: [(coder, data, padding, q_index, qp_scale)].  All LL decodes run concurrently on side streams."""
        return self.pwave_decompress_many_end(self.pwave_decompress_many_begin(jobs))

    def pwave_decompress_many_begin(self, jobs):
        """Parse the files and start their sequential LL decodes (side streams); the caller may do other work — the
        motion stream's decode — before pwave_decompress_many_end."""
        while len(self.side_streams) < len(jobs):
            self.side_streams.append(torch.cuda.Stream(device=self.dev))
        return [self.pwave_decompress_begin(*j, stream=self.side_streams[i]) for i, j in enumerate(jobs)]

    def pwave_decompress_many_end(self, begun):
        if len(begun) == 1:
            return [self.pwave_decompress_end(begun[0])]
        # The files are independent: each is finished by its own host thread on its own stream (the four-step decode
        # alternates GPU network evaluations with host range decoding, so one file alone leaves both sides idle half the
        # time; ctypes and the device waits release the GIL).  Chroma's subbands are decoded while luma's four times
        # longer sequential LL part is still running on its CU.
        main = torch.cuda.current_stream(self.dev)
        ready = torch.cuda.Event()
        ready.record(main)

        def finish(i):
            torch.cuda.set_device(self.dev)
            st = self.side_streams[i]
            st.wait_event(ready)
            with torch.no_grad(), torch.cuda.stream(st):
                return self.pwave_decompress_end(begun[i])
        if self._dec_pool is None:
            self._dec_pool = ThreadPoolExecutor(max_workers=4)
        out = list(self._dec_pool.map(finish, range(len(begun))))
        for i, t in enumerate(out):
            main.wait_stream(self.side_streams[i])
            t.record_stream(main)
        return out

    def pwave_decompress_end(self, job):
        coder, dec, N = job["coder"], job["dec"], job["N"]
        q_scale, q_scale_ll = job["q"]
        new_h, new_w, sh, sw = job["shape"]
        ll_rec = self.ll_ar_finish(job["ticket"])
        hat = {lvl: {} for lvl in range(self.L)}
        hat[self.L - 1]["ll"] = ll_rec
        lstm_state = self.ctx_init(N, sh, sw)
        context = self.ctx_forward_one_subband(coder, lstm_state, ll_rec, "ll", self.L - 1)
        for lvl in range(self.L - 1, -1, -1):
            h, w = new_h >> (lvl + 1), new_w >> (lvl + 1)
            for sidx, sb in enumerate(("lh", "hl", "hh")):
                ctx = ops.empty_nhwc(N, h, w, 1, self.dev)
                ew(EW_COPY, as_nchw(context)[:, sidx:sidx + 1], out=as_nchw(ctx))
                prev = hat[lvl + 1][sb] if lvl < self.L - 1 else None
                s_hat = self.fusion_decompress(f"{coder}.context_fusion.{lvl}.{sb}", ctx, prev, dec, N, h, w)
                hat[lvl][sb] = s_hat
                context = self.ctx_forward_one_subband(coder, lstm_state, s_hat, sb, lvl)
        rec_ll = ew(EW_DIVS, hat[self.L - 1]["ll"], alpha=q_scale_ll)
        out = None
        for lvl in range(self.L - 1, -1, -1):
            sbs = {"ll": rec_ll}
            for sb in ("lh", "hl", "hh"):
                sbs[sb] = ew(EW_DIVS, hat[lvl][sb], alpha=q_scale)
            out = self.backward_lift_2d(coder, sbs)
            rec_ll = out
        return self.post_process(coder, out, 256.0, 256.0)

    def decompress_mv(self, string, height, width, dpb, stage_idx=0, q_index=0, me_downsample=1):
        """pMCTF.decompress_mv (pMCTF_L.py:497-523); height/width: size of the plane motion was estimated on"""
        s = min(self.num_me_stages - 1, stage_idx)
        _, q_dec = self.get_mv_y_q(q_index, s)
        dec = HostDecoder(self.tables, string)
        p = 64
        hz = int(((int(height) + p - 1) // p * p) / p + 0.5)
        wz = int(((int(width) + p - 1) // p * p) / p + 0.5)
        idx = np.repeat(np.arange(64, dtype=np.int16), hz * wz)
        zsym = torch.from_numpy(dec.decode(idx, f"z{s}")).to(self.dev)
        z_hat = ops.sym_to_nhwc(zsym, hz, wz, 64)
        common = self.mv_prior_param_decoder(z_hat, self.to_nhwc_input(dpb["ref_mv_y"]), s)
        hy, wy = common.shape[1], common.shape[2]
        so_far = torch.empty((1, hy, wy, 64), dtype=torch.float32, device=self.dev)
        sp = None
        for t in range(4):
            if t > 0:
                x = self.conv(f"mv_y_spatial_prior_adaptor_{t}.{s}")(self.cat_channels(so_far, common))
                for i in range(3):
                    x = self.depth_conv_block(f"mv_y_spatial_prior.{s}.{i}", x)
                sp = x
            idx_d = ops.mv_fourpart_indexes(common, sp, hy, wy, t, self.lmin, self.lstep)
            sym = self._decode(dec, idx_d, "gauss")
            ops.mv_fourpart_dequant(sym, common, sp, so_far, t)
        mv_y_hat = ops.mv_dequant(so_far, common)
        mv_hat, mv_feature = self.mv_dec(s, mv_y_hat, q_dec)
        if me_downsample > 1:       # :516-517
            mv_hat = ops.bilinear_up2(mv_hat, float(me_downsample), me_downsample)
        return {"mv_hat": mv_hat, "mv_feature": mv_feature, "mv_y_hat": mv_y_hat}


class HostDecoder:
    """One bitstream on the host range decoder (libpmctf_rans.so), sharing the engine's CDF tables."""

    def __init__(self, tables, stream):
        self.R = _lib.rans()
        self.tables = tables
        self.h = self.R.pmctf_rans_decoder_create(1)
        buf = np.frombuffer(bytes(stream), dtype=np.uint8).copy()
        _lib.check(self.R.pmctf_rans_decoder_set_stream(self.h, buf.ctypes.data, buf.size), "rans set_stream")
        self.words = np.frombuffer(buf[1:].tobytes(), dtype=np.uint32)       # payload words after the flag byte

    def __del__(self):
        if getattr(self, "h", None):
            self.R.pmctf_rans_decoder_destroy(self.h)
            self.h = None

    def decode(self, idx, table):
        idx = np.ascontiguousarray(idx, dtype=np.int16).reshape(-1)
        cdf, sizes, offsets = self.tables[table]
        out = np.empty(idx.size, np.int16)
        _lib.check(self.R.pmctf_rans_decoder_decode_stream(self.h, idx.ctypes.data, idx.size, cdf.ctypes.data,
                                                           cdf.shape[0], cdf.shape[1], sizes.ctypes.data,
                                                           offsets.ctypes.data, out.ctypes.data), "rans decode_stream")
        return out

    def get_state(self):
        x, pos = C.c_uint64(0), C.c_int64(0)
        _lib.check(self.R.pmctf_rans_decoder_get_state(self.h, C.byref(x), C.byref(pos)), "rans get_state")
        return x.value, pos.value

    def set_state(self, x, pos):
        _lib.check(self.R.pmctf_rans_decoder_set_state(self.h, C.c_uint64(x), pos), "rans set_state")


def as_nchw_ew(op, t_nhwc, alpha=0.0, beta=0.0):
    """elementwise op on an NHWC tensor, result NHWC (N,H,W,C)"""
    return ew(op, as_nchw(t_nhwc), alpha=alpha, beta=beta).permute(0, 2, 3, 1)


as_nhwc_ew = as_nchw_ew
