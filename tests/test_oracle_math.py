"""PM-F32 transcendentals vs libm/torch in float64: <= 3 ulp; and oracle C primitives vs ATen CPU ops."""
import numpy as np
import torch
import torch.nn.functional as F


def _ulp(a, ref):
    return np.abs(a.astype(np.float64) - ref) / np.spacing(np.abs(ref).astype(np.float32))


def test_transcendentals_accuracy():
    from pmctf_oracle import clib
    r = np.random.default_rng(0)
    v = (r.standard_normal(200000) * 4).astype(np.float32)
    assert _ulp(clib.tanh(v), np.tanh(v.astype(np.float64))).max() <= 3.0
    assert _ulp(clib.sigmoid(v), 1 / (1 + np.exp(-v.astype(np.float64)))).max() <= 3.0
    assert _ulp(clib.exp(v), np.exp(v.astype(np.float64))).max() <= 2.0
    s = np.exp(r.uniform(np.log(1e-5), np.log(100), 200000)).astype(np.float32)
    assert np.abs(clib.log(s) - np.log(s.astype(np.float64))).max() <= 1e-6


def test_conv_matches_aten_within_fp_noise():
    from pmctf_oracle import clib
    r = np.random.default_rng(1)
    for (cin, cout, k, s, p) in [(112, 112, 3, 1, 1), (8, 32, 7, 1, 3), (64, 64, 3, 2, 1), (64, 64, 1, 2, 0), (2, 64, 3, 2, 1),
                                 (16, 1, 3, 1, 1)]:
        x = r.standard_normal((2, cin, 21, 37), dtype=np.float32)
        w = (r.standard_normal((cout, cin, k, k), dtype=np.float32) * .05).astype(np.float32)
        b = r.standard_normal(cout, dtype=np.float32)
        y = clib.conv2d(x, w, b, s, (p, p))
        yt = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), stride=s, padding=p).numpy()
        assert y.shape == yt.shape and np.abs(y - yt).max() < 5e-5
    x = r.standard_normal((2, 64, 17, 23), dtype=np.float32)
    w = r.standard_normal((64, 1, 3, 3), dtype=np.float32)
    b = r.standard_normal(64, dtype=np.float32)
    yt = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), padding=1, groups=64).numpy()
    assert np.abs(clib.dwconv2d(x, w, b) - yt).max() < 1e-5


def test_resampling_matches_aten():
    from pmctf_oracle import clib
    x = torch.randn(1, 2, 128, 256) * 10
    assert np.array_equal(F.avg_pool2d(x, 2, 2).numpy(), clib.avgpool2(x.numpy()))
    up = F.interpolate(x, (256, 512), mode="bilinear", align_corners=False).numpy()
    assert np.abs(up - clib.bilinear_up2(x.numpy())).max() < 1e-5
    dn = F.interpolate(x, (64, 128), mode="bilinear", align_corners=False).numpy()
    assert np.abs(dn - clib.bilinear_down2(x.numpy())).max() < 1e-5
    H, W = 64, 96
    im = torch.rand(2, 1, H, W) * 255
    flow = torch.randn(1, 2, H, W) * 5
    lx, ly = torch.linspace(-1, 1, W), torch.linspace(-1, 1, H)
    grid = torch.cat([lx.view(1, 1, 1, W).expand(1, -1, H, -1), ly.view(1, 1, H, 1).expand(1, -1, -1, W)], 1)
    grid = grid + torch.cat([flow[:, 0:1] / ((W - 1.0) / 2.0), flow[:, 1:2] / ((H - 1.0) / 2.0)], 1)
    ref = F.grid_sample(im, grid.tile((2, 1, 1, 1)).permute(0, 2, 3, 1), mode="bilinear", padding_mode="border",
                        align_corners=True).numpy()
    mine = clib.flow_warp(im.numpy(), flow.numpy(), lx.numpy(), ly.numpy())
    assert np.abs(ref - mine).max() < 1e-4
