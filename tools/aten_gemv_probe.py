#!/usr/bin/env python3
"""Probe inputs (powers of two, half-ulp terms, inexact products) that reveal in which order ATen sums the nine products of
a 1 -> 1 3x3 convolution on a single small plane (im2col + sgemm with one output channel = MKL gemv): which pairs of
terms are added before they meet the bias or a given term, and which products are fused (fma) onto which accumulator.
The order it prints is restated as PMCTF_SUM_GEMV_3X3 (include/pmctf_hip.h).  Build-container tool (torch only)."""
import numpy as np, torch, torch.nn.functional as F, itertools, sys
h, w = 96, 176
if len(sys.argv) > 1: h, w = int(sys.argv[1]), int(sys.argv[2])
cy, cx = 40, 70
def probe(terms, bias, wts=None):
    x = torch.zeros(1, 1, h, w)
    k = 0
    for ky in range(3):
        for kx in range(3):
            x[0, 0, cy + ky - 1, cx + kx - 1] = float(terms[k]); k += 1
    wt = torch.ones(1, 1, 3, 3) if wts is None else torch.tensor(wts, dtype=torch.float32).reshape(1, 1, 3, 3)
    return float(F.conv2d(x, wt, torch.tensor([float(bias)]), padding=1)[0, 0, cy, cx])
e = 2.0 ** -24
print("bias=1, pairs (i,j) with 2^-24 each: '+' = combined before meeting the bias")
for i in range(9):
    row = ""
    for j in range(9):
        if i == j: row += " ."; continue
        t = [0.0] * 9; t[i] = e; t[j] = e
        row += " +" if probe(t, 1.0) > 1.0 else " -"
    print(i, row)
for p in (0, 4, 8):
    print(f"bias=0, term {p} = 1, pairs (i,j): '+' = i,j combined before meeting term {p}")
    for i in range(9):
        row = ""
        for j in range(9):
            if i == j or i == p or j == p: row += " ."; continue
            t = [0.0] * 9; t[p] = 1.0; t[i] = e; t[j] = e
            row += " +" if probe(t, 0.0) > 1.0 else " -"
        print(i, row)
# fma or mul+add? term product inexact: x = 1+2^-12, w = 1+2^-12 -> product 1 + 2^-11 + 2^-24 ; with bias -1: fma gives 2^-11+2^-24 exactly, mul+add gives 2^-11
t = [0.0] * 9; t[4] = 1 + 2.0 ** -12
wts = [1.0] * 9; wts[4] = 1 + 2.0 ** -12
r = probe(t, -1.0, wts)
print("fma test (centre tap):", r, "fma" if r != 2.0 ** -11 else "mul+add")
for k in range(9):
    t = [0.0] * 9; t[k] = 1 + 2.0 ** -12
    wts = [1.0] * 9; wts[k] = 1 + 2.0 ** -12
    r = probe(t, -1.0, wts)
    print(k, "fma" if r != 2.0 ** -11 else "mul+add", end="; ")
print()
print("--- final combination of E (bias chain 4,6), O (5,7), A (0..3), then tap 8")
e = 2.0 ** -24
t = [0.0] * 9; t[0] = -1.0; t[5] = e
print("E=1, A=-1, O=e:", probe(t, 1.0), "(E+O)+A -> 0 ; (E+A)+O -> 2^-24 =", e)
t = [0.0] * 9; t[0] = e; t[5] = -1.0
print("E=1, O=-1, A=e:", probe(t, 1.0), "(E+O)+A -> e ; (E+A)+O -> 0")
# inside A: t_e = p0 (+) p2, t_o = p1 (+) p3, A = t_e + t_o ?
t = [0.0] * 9; t[0] = 1.0; t[2] = e; t[1] = e; 
print("A: p0=1,p2=e,p1=e ->", probe(t, 0.0), " [(p0+p2)+(p1+p3) -> 1 ; ((p0+p1)+p2) -> 1; p0+(p1+p2) -> 1+2e]")
t = [0.0] * 9; t[0] = 1.0; t[1] = -1.0; t[2] = e
print("A: p0=1,p1=-1,p2=e ->", probe(t, 0.0), " [(p0+p2)+(p1+p3) -> 0 ; (p0+p1)+p2.. -> e]")
t = [0.0] * 9; t[0] = 1.0; t[2] = -1.0; t[1] = e
print("A: p0=1,p2=-1,p1=e ->", probe(t, 0.0), " [(p0+p2)+(p1+p3) -> e]")
# fma inside chains: p2 fma onto p0? product inexact test within A: x0*w0 = -1 exactly, x2*w2 = (1+2^-12)^2
t = [0.0] * 9; t[0] = -1.0; t[2] = 1 + 2.0 ** -12
wts = [1.0] * 9; wts[2] = 1 + 2.0 ** -12
r = probe(t, 0.0, wts); print("tap2 onto tap0:", "fma" if r != 2.0 ** -11 else "mul+add")
t = [0.0] * 9; t[1] = -1.0; t[3] = 1 + 2.0 ** -12
wts = [1.0] * 9; wts[3] = 1 + 2.0 ** -12
r = probe(t, 0.0, wts); print("tap3 onto tap1:", "fma" if r != 2.0 ** -11 else "mul+add")
t = [0.0] * 9; t[5] = -1.0; t[7] = 1 + 2.0 ** -12
wts = [1.0] * 9; wts[7] = 1 + 2.0 ** -12
r = probe(t, 0.0, wts); print("tap7 onto tap5:", "fma" if r != 2.0 ** -11 else "mul+add")
t = [0.0] * 9; t[4] = -1.0; t[6] = 1 + 2.0 ** -12
wts = [1.0] * 9; wts[6] = 1 + 2.0 ** -12
r = probe(t, 0.0, wts); print("tap6 onto tap4:", "fma" if r != 2.0 ** -11 else "mul+add")
print("--- order inside the chains: 'a then b' is fma of b onto a")
for a, b in ((0, 2), (2, 0), (1, 3), (3, 1), (4, 6), (6, 4), (5, 7), (7, 5)):
    t = [0.0] * 9; t[a] = -1.0; t[b] = 1 + 2.0 ** -12
    wts = [1.0] * 9; wts[b] = 1 + 2.0 ** -12
    r = probe(t, 0.0, wts); print(f"tap{b} onto tap{a}:", "fma (a first)" if r != 2.0 ** -11 else "rounded product")
# bias vs E chain: bias + p6 + p4 ?  bias=-1, p6 inexact -> fma (known). bias = 0? 
t = [0.0] * 9; t[6] = -1.0; t[4] = 1 + 2.0 ** -12
wts = [1.0] * 9; wts[4] = 1 + 2.0 ** -12
print("bias=0: tap4 onto tap6:", probe(t, 0.0, wts) != 2.0 ** -11)
# is A added as (E+O)+A with A = (p0,p2 chain)+(p1,p3 chain)?  and is tap 8 last: ((E+O)+A) then fma p8
t = [0.0] * 9; t[8] = 1 + 2.0 ** -12
wts = [1.0] * 9; wts[8] = 1 + 2.0 ** -12
t[0] = -1.0
print("tap8 fma onto total (A=-1):", probe(t, 0.0, wts) != 2.0 ** -11)
