#!/usr/bin/env python3
"""Where does the decode of one 1080p pair go?  Wall-clock per phase (the decoder alternates GPU work and host range
decoding, so phases are timed on the host with a device synchronisation at their ends)."""
import collections, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learned-pmctf_amd"))
import torch
import pmctf_synth
from pMCTF.models.video.pMCTF_L import pMCTF
from pMCTF.hip import engine as E
net = pMCTF(num_me_stages=4).eval()
net.load_state_dict(pmctf_synth.synth_state_dict(net.state_dict(), seed=0), strict=True)
net = net.cuda(); net.update(force=True)
W, H = 1920, 1080
fr = [list(pmctf_synth.frames_to_tensors(f, device="cuda")) for f in pmctf_synth.synth_yuv420(W, H, 2)]
tmp = tempfile.mkdtemp()
eng = net.engine()
acc = collections.defaultdict(float); cnt = collections.Counter()
def wrap(obj, name, label, sync=True):
    f = getattr(obj, name)
    def g(*a, **k):
        if sync: torch.cuda.synchronize()
        t = time.time(); r = f(*a, **k)
        if sync: torch.cuda.synchronize()
        acc[label] += time.time() - t; cnt[label] += 1
        return r
    setattr(obj, name, g)
code_lt = len(sys.argv) > 1 and sys.argv[1] == "L"
with torch.no_grad():
    dpb = {"mv_feature": None, "ref_mv_y": None}
    for it in range(2):
        r = net.encode_one_stage(fr[0], fr[1], code_lt, dpb, output_path=os.path.join(tmp, "1.bin"), pic_width=W, pic_height=H,
                                 skip_decoding=False, stage_idx=0, q_index=3)
    print("decoding_time (unprofiled)", r["decoding_time"])
    if os.environ.get("PROFILE", "1") == "1":
        wrap(eng, "decompress_mv", "motion stream (decompress_mv)")
        wrap(eng, "ll_ar_finish", "wait for the sequential LL decode")
        wrap(eng, "post_process", "post-processing CNN")
        wrap(eng, "backward_lift_2d", "inverse DWT level")
        wrap(E.HostDecoder, "decode", "host range decoding (decode_stream)", sync=False)
        orig_decode = eng._decode
        def _decode(dec, idx_dev, table):
            torch.cuda.synchronize(); t = time.time()
            r = orig_decode(dec, idx_dev, table)
            torch.cuda.synchronize(); acc["index D2H + host decode + symbol H2D"] += time.time() - t; cnt["index D2H + host decode + symbol H2D"] += 1
            return r
        eng._decode = _decode
        torch.cuda.synchronize(); t0 = time.time()
        r = net.encode_one_stage(fr[0], fr[1], code_lt, dpb, output_path=os.path.join(tmp, "1.bin"), pic_width=W, pic_height=H,
                                 skip_decoding=False, stage_idx=0, q_index=3)
        print("decoding_time (with per-phase synchronisation)", r["decoding_time"])
        for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
            print(f"  {v * 1e3:8.1f} ms  {cnt[k]:4d} calls  {k}")
