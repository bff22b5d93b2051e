#!/usr/bin/env python3
"""Merge the per-configuration reports a GPU run of tests/test_gpu_engine.py wrote (PMCTF_HEADLINE_REPORT=<dir>) into the
pins files under tests/golden: headline_pins.json (panning sequence, 1080p) and second_sequence_pins.json ("layers").
usage: write_pins.py <report dir>"""
import glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
head, second = {}, {}
for f in sorted(glob.glob(os.path.join(sys.argv[1], "gop*_q*.json"))):
    name = os.path.basename(f)[:-5]
    m = re.fullmatch(r"(gop\d+_q\d+)(_layers)?", name)
    if not m:
        continue                      # other sizes / half-resolution motion: asserted directly by their tests
    (second if m.group(2) else head)[m.group(1)] = json.load(open(f))
for path, d in (("headline_pins.json", head), ("second_sequence_pins.json", second)):
    if d:
        keep = {k: {kk: v[kk] for kk in ("dbits", "psnr_err", "same", "diff", "lengths_equal")} for k, v in sorted(d.items())}
        json.dump(keep, open(os.path.join(ROOT, "tests", "golden", path), "w"), indent=1)
        print(path, {k: (v["same"], v["diff"], any(v["dbits"])) for k, v in keep.items()})
