"""Captured launch plans for `encode_one_stage` (write-stream branch, pMCTF_L.py:553-637).

A harness calls `encode_one_stage` once per frame pair and looks at the bit counts before the next call
(test_pMCTF_flex.py:236-249), so every pair is a batch of its own: ~3 000 launches, a third of them on the small
planes of the wavelet pyramid, where the host cannot enqueue as fast as the GPU executes and where a lone coder
leaves most of the 256 CUs idle.  A `PairPlan` records the launches of one configuration (plane shapes, temporal
stage, rate point, with/without L, first/later pair of the motion-context chain) ONCE as HIP graphs — the same
engine code runs under stream capture — and replays them for every later pair of that configuration:

    main stream   inputs -> [motion estimation] -> [motion codec + D2H of its symbols]
    luma stream       [lifting + H analysis + D2H] (-> [L analysis + D2H])      .. -> [synthesis] -> results
    chroma stream     [lifting + H analysis + D2H] (-> [L analysis + D2H])      .. -> [synthesis] -> results

Luma and chroma are independent once the motion field exists, so their graphs run concurrently: the small-plane
launches of one fill the CUs the other leaves idle, and replay costs the host microseconds, not milliseconds.
The synthesis transforms (needed for the returned reconstructions only) start when both analyses are done, so
that every symbol stream reaches the host range coder as early as possible and the coder's tail hides under GPU
work.  A graph replays the very kernels the stream path launches, in the same order with the same arguments:
results are bit-identical (tests/test_gpu_engine.py::test_pair_plan_equals_stream_launches).

Static storage: a plan owns its input planes, the pinned host buffers of its symbol streams and (inside the
graphs' memory pools) what one segment hands to the next; results are copied out into fresh tensors per call, because
the caller keeps them across the GOP (test_pMCTF_flex.py:225-226).  All plans of an engine share TWO graph memory
pools — one for the main / luma chain, one for the chroma chain (the two run concurrently, everything inside a chain
and every two plans run one after the other) — so the scratch memory of the ~3 000 launches of a pair exists once per
chain, not once per plan: ~8 GB per engine at 1080p instead of ~60.
"""
import os
import threading
import weakref

import torch

from . import ops


class _Resources:
    """Who destroys a launch plan's HIP objects, and when.

    Destroying a HIP graph (and handing its memory pool back) while ANY stream of the process is being captured ends
    the process, and Python decides by itself when an object dies: a model that was dropped leaves its engine and plans
    behind as garbage, and the collector — or a worker thread letting go of the last reference — may find them at any
    moment, including in the middle of another model's recording.  So no plan owns its graphs alone: this registry holds
    a second reference to everything a plan (and an engine's plan context) keeps alive on the device.  When the owner
    dies, its finaliser only moves that bundle to `retired` (pure Python, legal anywhere); `drain()` — called where no
    capture can be open: before a recording starts (under the capture gate), when an engine is built or released —
    synchronises the device and lets the bundles go."""

    def __init__(self):
        self.lock = threading.Lock()
        self.live = {}              # token -> bundle (dict / list of graphs, static tensors, pinned buffers, streams)
        self.retired = []
        self.recording = 0          # recordings in progress (they hold the capture gate exclusively: 0 or 1)
        self.broken = []            # graph objects of failed captures, see _Capture.abort
        self.next = 0
        self.stats = {"adopted": 0, "retired": 0, "destroyed": 0, "retired_while_recording": 0}

    def adopt(self, owner, bundle):
        with self.lock:
            token = self.next
            self.next += 1
            self.live[token] = bundle
            self.stats["adopted"] += 1
        weakref.finalize(owner, self._retire, token)
        return token

    def _retire(self, token):
        with self.lock:
            bundle = self.live.pop(token, None)
            if bundle is not None:
                self.retired.append(bundle)
                self.stats["retired"] += 1
                if self.recording:
                    self.stats["retired_while_recording"] += 1

    def drain(self):
        """destroy what dead plans left behind; a no-op while a recording is in progress (the next safe point does it)"""
        with self.lock:
            if self.recording or not self.retired:
                return 0
            gone, self.retired = self.retired, []
        if torch.cuda.is_available():
            assert not torch.cuda.is_current_stream_capturing()
            torch.cuda.synchronize()
        n = len(gone)
        del gone[:]                 # the graphs, their pools, the pinned buffers go here
        with self.lock:
            self.stats["destroyed"] += n
        return n

    def begin_recording(self):
        self.drain()
        with self.lock:
            self.recording += 1

    def end_recording(self):
        with self.lock:
            self.recording -= 1


RESOURCES = _Resources()
IN_CAPTURE_HOOK = None      # tests: called once inside an open capture of every recording (e.g. to force a collection)


class _Capture:
    """Cuts the launches issued by ordinary Python code into a sequence of HIP graphs."""

    def __init__(self, pool):
        self.graphs = []
        self.g = None
        self.pool = pool
        self.stream = None

    def begin(self):
        self.stream = torch.cuda.current_stream()
        self.g = torch.cuda.CUDAGraph()
        self.g.capture_begin(pool=self.pool, capture_error_mode="relaxed")

    def cut(self):
        self.end()
        self.begin()

    def end(self):
        self.g.capture_end()
        self.graphs.append(self.g)
        self.g = None

    def abort(self):
        """a recording failed: close the open capture — on the stream it was opened on, whatever the caller's current
        stream is by now (an exception has usually unwound the `with torch.cuda.stream(...)` block already; ending the
        capture from another stream fails and leaves the stream capturing for good) — so that the engine can go on with
        stream launches"""
        if self.g is not None:
            try:
                with torch.cuda.stream(self.stream):
                    self.g.capture_end()
                self.graphs.append(self.g)      # a valid, partial graph: destroyed like any other (RESOURCES)
            except Exception:  # noqa: BLE001 - the original error is the one to report
                # a graph object whose capture was invalidated throws from its destructor (which ends the process): it
                # is never destroyed.  Bounded: an engine whose recording failed stops recording
                # (pMCTF.encode_one_stage), so at most one object per engine ends up here.
                RESOURCES.broken.append(self.g)
            self.g = None


class PairPlan:
    def __init__(self, eng, ry, rc, chained, code_lt, stage_idx, q_index, me_downsample=1):
        """ry / rc: a luma (1,1,H,W) and a chroma (2,1,H/2,W/2) plane of the size to plan for (shapes only).
        Must be called when every layer this configuration uses has been packed already (i.e. after one pair of the
        configuration went through the stream path) — packing synchronises, which a capture cannot."""
        dev = eng.dev
        pool_y, pool_c = eng.plan_context()["pools"]
        self.chained, self.code_lt = chained, code_lt
        _, _, H, W = ry.shape
        new = lambda t: torch.empty(tuple(t.shape), dtype=torch.float32, device=dev)
        self.in_ry, self.in_cy, self.in_rc, self.in_cc = new(ry), new(ry), new(rc), new(rc)
        self.in_dpb = {"mv_feature": None, "ref_mv_y": None}
        if chained:
            h, w = H // me_downsample, W // me_downsample
            self.in_dpb = {"mv_feature": torch.empty((1, h // 4, w // 4, 64), dtype=torch.float32, device=dev).permute(0, 3, 1, 2),
                           "ref_mv_y": torch.empty((1, h // 16, w // 16, 64), dtype=torch.float32, device=dev).permute(0, 3, 1, 2)}
        pin = lambda n: (torch.empty(n, dtype=torch.int16, pin_memory=True), torch.empty(n, dtype=torch.int16, pin_memory=True))
        n_mv = eng.mv_symbol_count(H, W, me_downsample)
        n_y, n_c = eng.pwave_symbol_count(1, H, W), eng.pwave_symbol_count(2, H // 2, W // 2)
        kinds = ("H", "L") if code_lt else ("H",)
        self.host = {"mv": pin(n_mv)}
        for k in kinds:
            self.host[k] = pin(n_y)
            self.host[k + "c"] = pin(n_c)
        self.segments = {}          # job name -> SymbolStream.segments of that bitstream
        # The coders of a plan run side by side: the other stream fills the tail of a launch, so the convolutions are
        # recorded WITHOUT the whole-rounds + remainder cut that pays on a single stream, and cut by cout tile only below
        # 40 000 pixels instead of 70 000 (measured on the harness loop, tools/eager_gop.py: 5.62 -> 5.67 -> 5.68
        # frames/s).  The options travel WITH every launch (ops.launch_opts -> pmctf_conv2d_nhwc_opts_f32); nothing
        # process-wide changes, so a model on another host thread keeps its own launch shapes.
        # The device objects of this plan are co-owned by RESOURCES (see there): whoever drops the plan, whenever, its
        # graphs are destroyed at the next point where no capture can be open.
        torch.cuda.synchronize(dev)
        RESOURCES.begin_recording()
        try:
            with ops.launch_opts(eng.plan_launch_opts):
                self._record(eng, pool_y, pool_c, dev, code_lt, stage_idx, q_index, me_downsample)
        finally:
            RESOURCES.end_recording()
            RESOURCES.adopt(self, dict(self.__dict__))

    def _record(self, eng, pool_y, pool_c, dev, code_lt, stage_idx, q_index, me_downsample):
        cap = _Capture(pool_y)
        self._captures = [cap]      # every graph of this recording, also of a failed one, is co-owned by RESOURCES
        cur = torch.cuda.current_stream(dev)
        side = eng.plan_context()["capture"]
        side.wait_stream(cur)
        try:
            with torch.cuda.stream(side):
                cap.begin()
                if IN_CAPTURE_HOOK is not None:
                    IN_CAPTURE_HOOK()
                est = eng.motion_estimate(self.in_ry, self.in_cy, me_downsample)
                cap.cut()
                mv = eng.motion_code(est, self.in_dpb, stage_idx, q_index, False, me_downsample)
                mv["stream"].copy_to(*self.host["mv"])
                self.segments["mv"] = list(mv["stream"].segments)
                cap.end()
                self.g_me, self.g_mv = cap.graphs
                self.mv = {k: mv[k] for k in ("mv_hat", "mv_feature", "mv_y_hat")}
                del est

                def analysis(ref, cur_, chroma):
                    c = _Capture(pool_c if chroma else pool_y)
                    self._captures.append(c)
                    suffix = "c" if chroma else ""

                    def on_stream(kind, stream):
                        stream.copy_to(*self.host[kind + suffix])
                        self.segments[kind + suffix] = list(stream.segments)
                        if kind == "H" and code_lt:
                            c.cut()
                    c.begin()
                    try:
                        out = eng.compress_one_stage(ref, cur_, code_lt, self.mv["mv_hat"], chroma, stage_idx, q_index, False,
                                                     on_stream=on_stream, defer=True)
                        c.end()
                    except BaseException:
                        c.abort()
                        raise
                    s = _Capture(pool_c if chroma else pool_y)
                    self._captures.append(s)
                    s.begin()
                    try:
                        out["finish"]()
                        s.end()
                    except BaseException:
                        s.abort()
                        raise
                    names = ("L_t_hat" if code_lt else "L_t", "H_t_hat")
                    return {n: out[n] for n in names}, c.graphs, s.graphs[0]
                del mv
                self.luma, self.g_luma, self.g_luma_syn = analysis(self.in_ry, self.in_cy, False)
                self.chroma, self.g_chroma, self.g_chroma_syn = analysis(self.in_rc, self.in_cc, True)
        except BaseException:
            cap.abort()
            raise
        cur.wait_stream(side)

    def run(self, eng, ry, cy, rc, cc, dpb, submit, on_dpb=None):
        """Replays the plan on the caller's frames.  dpb: the motion context (dict of logical-NCHW tensors, or a
        zero-argument callable delivering it after the motion estimation).  submit(job, hs, hi, event, segments) hands one
        bitstream's pinned symbol buffers to the range coder.  Returns the tensors of encode_one_stage's result."""
        A, B = eng.plan_context()["streams"]
        main = torch.cuda.current_stream(eng.dev)
        if eng.plan_timing is not None:
            e_begin = torch.cuda.Event(enable_timing=True)
            e_begin.record(main)
        for dst, src in ((self.in_ry, ry), (self.in_cy, cy), (self.in_rc, rc), (self.in_cc, cc)):
            dst.copy_(src)
        self.g_me.replay()
        if callable(dpb):
            dpb = dpb()
        if self.chained:
            for k in ("mv_feature", "ref_mv_y"):
                self.in_dpb[k].copy_(dpb[k])
        self.g_mv.replay()
        timing = eng.plan_timing is not None
        mark = lambda: torch.cuda.Event(enable_timing=timing)
        if timing:
            e_start = mark()
            e_start.record(main)
        e_mv = mark()
        e_mv.record(main)
        fresh = lambda t: torch.empty(tuple(t.shape), dtype=torch.float32, device=eng.dev)
        mv = self.mv
        res = {"mv_hat": fresh(mv["mv_hat"]), "mv_feature": fresh(mv["mv_feature"]), "mv_y_hat": fresh(mv["mv_y_hat"])}
        for k in ("mv_hat", "mv_feature", "mv_y_hat"):
            res[k].copy_(mv[k])
        if on_dpb is not None:
            on_dpb({"mv_feature": res["mv_feature"].permute(0, 3, 1, 2), "ref_mv_y": res["mv_y_hat"].permute(0, 3, 1, 2)})
        submit("mv", *self.host["mv"], e_mv, self.segments["mv"])
        kinds = ("H", "L") if self.code_lt else ("H",)
        done = []
        for st, graphs, suffix in ((A, self.g_luma, ""), (B, self.g_chroma, "c")):
            st.wait_event(e_mv)
            with torch.cuda.stream(st):
                for g, kind in zip(graphs, kinds):
                    g.replay()
                    ev = mark()
                    ev.record(st)
                    submit(kind + suffix, *self.host[kind + suffix], ev, self.segments[kind + suffix])
                done.append(ev)
        if eng.syn_after_analysis:      # every symbol stream is on its way before any synthesis transform starts
            A.wait_event(done[1])
            B.wait_event(done[0])
        for st, g, out, suffix in ((A, self.g_luma_syn, self.luma, ""), (B, self.g_chroma_syn, self.chroma, "c")):
            # the result tensors belong to the caller's stream (allocated there); only the copies run on the side stream
            names = ("L_t_hat" if self.code_lt else "L_t", "H_t_hat")
            outs = [fresh(out[n]) for n in names]
            with torch.cuda.stream(st):
                g.replay()
                for o, n in zip(outs, names):
                    o.copy_(out[n])
            res["L_t" + suffix], res["H_t" + suffix] = outs
        if timing:
            e_syn = []
            for st in (A, B):
                e = mark()
                e.record(st)
                e_syn.append(e)
        main.wait_stream(A)
        main.wait_stream(B)
        if timing:
            e_end = mark()
            e_end.record(main)
            eng.plan_timing.append((e_start, e_mv, done[0], done[1], e_syn[0], e_syn[1], e_begin, e_end))
        return res
