"""Which summation rule ATen's CPU convolution applies to a layer of the path — the part of the arithmetic
specification that depends on a layer's SHAPE.

The reference's CPU path is `F.conv2d` on ATen: oneDNN's direct convolution for KH*KW > 1 (rule "blocks",
include/pmctf_hip.h), oneDNN's jit_1x1 kernel for 1x1 layers.  jit_1x1 runs ONE fmaf chain from the bias over all input
channels unless its blocking heuristic cuts the reduction into blocks of B channels, in which case every block after the
first starts from zero and the block results are added in turn ("reduce-B").  B follows from the layer's shape alone;
`onednn_1x1_reduce_block` restates that heuristic (oneDNN 3.7.1 as built into torch 2.10, avx512_core, the machine the
fixtures under tests/golden were generated on):

  * base block: 80 channels when 10 < H < 28 and Cout <= 256, 256 when H <= 10, else 512 (capped at Cin);
  * cache-aliasing reduction: when H*W*block exceeds 7 ways of a 64 KB way (16 384 floats), the kernel walks the
    input in steps of one way (1 024 sixteen-float vectors) and stops at the 7th step that lands on the first `ur`
    vectors of a channel block's plane; the block is cut to 16 * (vectors walked / plane size).  For planes whose
    size is a multiple of 1 024 this gives 96 channels (288x480, 576x960, 64x96 ...).

It was fitted and checked against F.conv2d bit for bit on ~300 (Cin, Cout, H, W) combinations by
tools/aten_conv_rules.py --fit (profiles/round4_aten_conv_rules.md); planes on which ATen does not use oneDNN at all for
a 1x1 layer (one image of at most 20 480 input elements: sgemm, one chain from zero, bias last) are handled in
`conv1x1_sum_rule` for layers of up to 16 input channels and otherwise keep the chain.
"""


def _jit_1x1_ur(oh, os_):
    """jcp.ur of the avx512_core forward kernel: the register-blocking search over 9 .. 6 output vectors"""
    max_regs, min_regs, size_threshold = 9, 6, 14
    for u in range(max_regs, min_regs - 1, -1):
        if (oh >= size_threshold and oh % u == 0) or (oh < size_threshold and os_ % u == 0):
            return u
    ur = min(max_regs, os_)
    tail = os_ % max_regs
    for i in range(max_regs, min_regs - 1, -1):
        t = os_ % i
        if t > tail or t == 0:
            ur, tail = i, t
            if t == 0:
                break
    return ur


def onednn_1x1_reduce_block(cin, cout, h, w):
    """-> B: the 1x1 layer's reduction runs in blocks of B input channels (B >= cin: one chain from the bias)"""
    if 10 < h < 28 and cout <= 256:
        rb = min(cin, 80)
    elif h > 10:
        rb = min(cin, 512)
    else:
        rb = min(cin, 256)
    sp = h * w
    way, max_hits, simd = 16384, 7, 16
    if sp * rb > way * max_hits:
        ur = _jit_1x1_ur(h, sp)
        nrb, wl = rb // simd, way // simd
        for start in range(ur):
            off, hits = start, 0
            while off < sp * nrb:
                if off % sp < ur:
                    hits += 1
                    if hits >= max_hits:
                        rb = min(rb, simd * max(1, (off + wl) // sp))
                        break
                off += wl
    return rb


def conv1x1_sum_rule(cin, cout, n, h, w):
    """sum_rule argument (include/pmctf_hip.h) of a 1x1, stride-1 layer of the signal path whose reference tensor is
    (n, cin, h, w): PMCTF_SUM_CHAIN, or the block size B of "reduce-B"."""
    if n == 1 and cin * h * w <= 20480:
        # ATen does not take the oneDNN path here (Convolution.cpp use_mkldnn) but a plain sgemm: ONE chain from zero, the
        # bias added last.  For up to 16 input channels that is rule "blocks" (1); wider layers on such small planes
        # (frames of a few hundred pixels a side) have no rule here and keep the chain from the bias.
        return 1 if cin <= 16 else 0
    b = onednn_1x1_reduce_block(cin, cout, h, w)
    return b if b < cin and b % 16 == 0 else 0


def conv_kxk_sum_rule(cin, cout, kh, kw, n, h, w):
    """sum_rule argument of a KH*KW > 1 layer whose reference tensor is (n, cin, h, w): 1 = "blocks" (oneDNN's direct
    convolution), or 2 = "gemm" where ATen leaves oneDNN (Convolution.cpp use_mkldnn: one image of at most 20 480 input
    elements, a filter of at most 3 rows or columns) for im2col + sgemm — one chain from zero over (ci, ky, kx), bias last.
    Restated for layers of up to 4 input channels (the small-cin kernels; with one channel both orders coincide) and for the
    1 -> 1 3x3 layer (3 = "gemv 3x3"); wider layers meet the condition only on planes of a few hundred pixels and keep
    "blocks"."""
    if n == 1 and cin * h * w <= 20480 and (kh <= 3 or kw <= 3):
        if 1 < cin <= 4 and cout > 1:
            return 2
        if cin == 1 and cout == 1 and kh == 3 and kw == 3:
            return 3           # one output channel: sgemm degenerates to a matrix-vector product ("gemv 3x3", pmctf_hip.h)
    return 1
