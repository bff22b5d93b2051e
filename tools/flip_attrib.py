#!/usr/bin/env python3
"""Flip attribution: WHICH layers of the path produce the entropy-coding decisions that differ between the reference's
arithmetic (ATen CPU ops) and the product's (PM-F32)?

The oracle can evaluate every primitive with either back-end (oracle/pmctf_oracle/kernels.py).  Here every primitive call
is assigned to a LAYER GROUP by the parameter key it is made with (convolutions) or by the function it sits in
(transcendentals); a run takes a set of groups from PM-F32 and everything else from ATen ("only" mode: the flips that
group alone produces) or the other way round ("except" mode: the flips that remain when that group alone is given ATen's
arithmetic).  The all-ATen run reproduces the real reference's symbols (tests/test_oracle_vs_golden.py), so a flip against
it is a flip against the reference.  Every differing symbol / CDF row is tagged with pair, stream, subband level, subband,
coding step and plane position.

CPU only (test infrastructure); no GPU, no reference import.

  python tools/flip_attrib.py --size 448x256 --q 4 12 20 [--gop 4] [--mode only except] [--groups ...] [--out FILE]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("learned-pmctf_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))

import torch  # noqa: E402

import pmctf_gop  # noqa: E402
from helpers import frames, synth_sd_cpu  # noqa: E402
from pmctf_oracle.kernels import CdefK, TorchK  # noqa: E402
from pmctf_oracle.model import Oracle  # noqa: E402

KINDS = ("mv", "H", "Hc", "L", "Lc")
GROUPS = ("spynet", "mv_enc", "mv_dec", "pu_temporal", "dwt_skip", "dwt_pu_c1", "dwt_pu_c23", "dwt_pu_c4", "ll_net", "lstm", "fs_first", "fs_heads", "fs_3x3_lvl3",
          "fs_3x3_lvl2", "fs_3x3_lvl1", "fs_3x3_lvl0", "post", "tanh_pu", "act_lstm", "log_index")


def group_of(key):
    """layer group of a convolution, from its parameter key"""
    if key.startswith("optic_flow."):
        return "spynet"
    if key.startswith("temporal_filtering."):
        return "pu_temporal"
    if key.startswith("mv_"):
        # what only decides the motion SYMBOLS (analysis side) / what the decoder also runs and mv_hat comes out of
        return "mv_enc" if key.startswith(("mv_encoder.", "mv_hyper_prior_encoder.")) else "mv_dec"
    if ".wavelet_transform." in key:
        leaf = key.rsplit(".", 1)[-1]
        if leaf.startswith("conv_"):
            return "dwt_skip"                       # the learned 3x1 filter of a lifting step
        return {"conv1": "dwt_pu_c1", "conv2": "dwt_pu_c23", "conv3": "dwt_pu_c23", "conv4": "dwt_pu_c4"}[leaf]
    if ".context_prediction." in key:
        return "lstm"
    if ".dequantModule." in key:
        return "post"
    if ".context_fusion." in key:
        tail = key.split(".context_fusion.")[1]
        lvl, sb, rest = tail.split(".", 2)
        if sb == "ll":
            return "ll_net"
        if rest.startswith("y_hierarchical_prior_out") or rest.endswith("_out.2"):
            return "fs_heads"                       # 1x1 / depthwise layers that end in (scale, mean)
        if rest.startswith("lower_level_subband") or rest.startswith("conv1_context") or \
                (rest.startswith("y_spatial_prior_") and rest.endswith(".0") and "_out" not in rest):
            return "fs_first"                       # first layers, 1-2 input channels
        return f"fs_3x3_lvl{lvl}"                   # the 112 -> 112 3x3 layers
    raise KeyError(key)


class _MixK(TorchK):
    """primitive back-end that asks the oracle which arithmetic the current call takes"""
    name = "mixed"

    def __init__(self, owner):
        self.o = owner
        self.t, self.c = TorchK(), CdefK()

    def _pick(self, group):
        return self.c if self.o.pm(group) else self.t

    def tanh(self, x):
        return self._pick("tanh_pu" if self.o.where == "pu" else "act_lstm").tanh(x.contiguous())

    def sigmoid(self, x):
        return self._pick("act_lstm").sigmoid(x.contiguous())

    def build_indexes(self, tables, scales):
        return self._pick("log_index").build_indexes(tables, scales)


class MixedOracle(Oracle):
    """pm_groups: the layer groups evaluated in PM-F32; everything else is ATen"""

    def __init__(self, sd, num_me_stages, pm_groups):
        super().__init__(sd, num_me_stages, "torch")
        self.pm_groups = frozenset(pm_groups)
        self.where = None
        self.K = _MixK(self)

    def pm(self, group):
        return group in self.pm_groups

    def conv(self, p, x, stride=1, padding=0, groups=1):
        w, b = self.sd[p + ".weight"], self.sd.get(p + ".bias")
        if self.pm(group_of(p)):        # PM-F32 with the summation rule the product gives this layer (Oracle.sum_rule)
            return self.K.c.conv2d(x.contiguous(), w, b, stride=stride, padding=padding, groups=groups,
                                   rule=self.sum_rule(p, x, w, groups, stride))
        return self.K.t.conv2d(x.contiguous(), w, b, stride=stride, padding=padding, groups=groups)

    def predict_update(self, p, x):
        self.where = "pu"
        try:
            return super().predict_update(p, x)
        finally:
            self.where = None


def push_tags(kind, n_push):
    """what each push of a stream codes (coding order: SURVEY appendix C)"""
    if kind == "mv":
        return ["z"] + [f"y{j}" for j in range(n_push - 1)]
    tags = ["ll"]
    for lvl in (3, 2, 1, 0):
        for sb in ("lh", "hl", "hh"):
            tags += [f"l{lvl}.{sb}.s{j}" for j in range(4)]
    assert len(tags) == n_push, (len(tags), n_push)
    return tags


def run(sd, fr, width, height, q_index, pm_groups):
    orc = MixedOracle(sd, 1, pm_groups)
    with tempfile.TemporaryDirectory() as td, torch.no_grad():
        enc = pmctf_gop.encode_gop(orc, fr, height, width, q_index, td)
    return {"traces": [r["traces"] for r in enc["results"]], "files": [r["files"] for r in enc["results"]],
            "bits": enc["bits"]}


def compare(a, b, shapes=None):
    """flips of run b against run a, tagged"""
    flips = []
    n_sym = 0
    files_diff = 0
    for i, (ta, tb) in enumerate(zip(a["traces"], b["traces"])):
        for k in KINDS:
            if ta.get(k) is None:
                continue
            tags = push_tags(k, len(ta[k]))
            for j, ((sa, ia), (sb_, ib)) in enumerate(zip(ta[k], tb[k])):
                n_sym += sa.size
                d = (sa != sb_) | (ia != ib)
                for pos in np.flatnonzero(d):
                    flips.append({"pair": i, "stream": k, "push": tags[j], "pos": int(pos), "of": int(sa.size),
                                  "sym": [int(sa[pos]), int(sb_[pos])], "row": [int(ia[pos]), int(ib[pos])]})
            files_diff += a["files"][i][k] != b["files"][i][k]
    dbits = (np.array(b["bits"]) - np.array(a["bits"])).astype(int).tolist()
    return {"symbols": n_sym, "flips": flips, "files_differ": int(files_diff), "dbits": dbits}


def summarise(c):
    by = {}
    for f in c["flips"]:
        key = f"{f['stream']}:{f['push'].split('.s')[0]}"
        by[key] = by.get(key, 0) + 1
    return by


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="448x256")
    ap.add_argument("--q", nargs="+", type=int, default=[4, 12, 20])
    ap.add_argument("--gop", type=int, default=4)
    ap.add_argument("--mode", nargs="+", default=["only"], choices=["only", "except", "all"])
    ap.add_argument("--groups", nargs="+", default=list(GROUPS))
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    w, h = (int(v) for v in args.size.split("x"))
    sd = synth_sd_cpu(1)
    fr = frames(w, h, args.gop, seed=args.seed)
    rows = []
    print("| size | q | PM-F32 groups | symbols | flips | flips / M | where (stream:subband = count) | files differ | "
          "bit deltas |", flush=True)
    print("|---|---|---|---|---|---|---|---|---|", flush=True)
    for q in args.q:
        t0 = time.time()
        base = run(sd, fr, w, h, q, ())
        sets = []
        if "all" in args.mode:
            sets.append(("ALL", tuple(GROUPS)))
        if "only" in args.mode:
            sets += [(g, (g,)) for g in args.groups]
        if "except" in args.mode:
            sets += [("ALL except " + g, tuple(x for x in GROUPS if x != g)) for g in args.groups]
        for label, pm in sets:
            c = compare(base, run(sd, fr, w, h, q, pm))
            s = summarise(c)
            rows.append({"size": args.size, "q": q, "pm_groups": label, "symbols": c["symbols"],
                         "n_flips": len(c["flips"]), "where": s, "files_differ": c["files_differ"],
                         "dbits": c["dbits"], "flips": c["flips"][:200]})
            print(f"| {args.size} | {q} | {label} | {c['symbols']} | {len(c['flips'])} | "
                  f"{1e6 * len(c['flips']) / c['symbols']:.2f} | "
                  f"{', '.join(f'{k}={v}' for k, v in sorted(s.items())) or '-'} | {c['files_differ']} | "
                  f"{[d for d in c['dbits']]} |", flush=True)
            if args.out:
                with open(args.out, "w") as f:
                    json.dump(rows, f, indent=1)
        print(f"<!-- q {q}: {time.time() - t0:.0f} s -->", flush=True)


if __name__ == "__main__":
    main()
