"""ORACLE — test infrastructure only; never imported by the product path.

Two interchangeable primitive back-ends for the functional restatement in model.py,
both operating on torch CPU float32 NCHW tensors:

  TorchK  the ATen CPU ops the reference itself calls (F.conv2d, F.grid_sample, ...).  On the
          machine that generated tests/golden this reproduces the reference bit for bit; it is
          how the restatement's control flow is pinned against the real reference.
  CdefK   the PM-F32 C restatement (oracle/c): fixed summation order, specified transcendentals.
          The HIP kernels are bit-exact against this back-end.

Elementwise +,-,*,/ , round, clamp, max and comparisons are plain torch ops in both (single IEEE
roundings, no fusion), and the HIP kernels perform the same single-rounded operations.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import clib


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


class TorchK:
    name = "torch"

    def conv2d(self, x, w, b, stride=1, padding=0, groups=1):
        return F.conv2d(x, w, b, stride=stride, padding=padding, groups=groups)

    def tanh(self, x):
        return torch.tanh(x)

    def sigmoid(self, x):
        return torch.sigmoid(x)

    sigmoid_plain = sigmoid

    def lin_tables(self, H, W):
        return torch.linspace(-1.0, 1.0, W), torch.linspace(-1.0, 1.0, H)

    def flow_warp(self, im, flow):
        # pMCTF/layers/video/video_net.py:32-50
        N, _, H, W = flow.size()
        lx, ly = self.lin_tables(H, W)
        hor = lx.view(1, 1, 1, W).expand(N, -1, H, -1)
        ver = ly.view(1, 1, H, 1).expand(N, -1, -1, W)
        grid = torch.cat([hor, ver], 1)
        fl = torch.cat([flow[:, 0:1] / ((im.size(3) - 1.0) / 2.0), flow[:, 1:2] / ((im.size(2) - 1.0) / 2.0)], 1)
        grid = grid + fl
        return F.grid_sample(im, grid.permute(0, 2, 3, 1), mode="bilinear", padding_mode="border",
                             align_corners=True)

    def avg_pool2(self, x):
        return F.avg_pool2d(x, kernel_size=2, stride=2)

    def bilinear_up2(self, x, f=2):
        return F.interpolate(x, (x.size(2) * f, x.size(3) * f), mode="bilinear", align_corners=False)

    def bilinear_down2(self, x, f=2):
        return F.interpolate(x, (x.size(2) // f, x.size(3) // f), mode="bilinear", align_corners=False)

    def build_indexes(self, tables, scales):
        return tables.build_indexes_torch(scales)

    # ---- estimate mode (gaussian_model.py:36-53,65-67) ------------------------------------------------------
    def laplace_bits(self, y, sigma):
        """CompressionModel.get_y_laplace_bits: the reference's own expressions"""
        import math
        mu = torch.zeros_like(sigma)
        sigma = sigma.clamp(1e-5, 1e10)
        lap = torch.distributions.laplace.Laplace(mu, sigma)
        probs = lap.cdf(y + 0.5) - lap.cdf(y - 0.5)
        bits = -1.0 * torch.log(probs + 1e-5) / math.log(2.0)
        return torch.clamp_min(bits, 0)

    def bitparm_cdf(self, x, params):
        """BitEstimator.get_cdf (entropy_models.py:72-77,114-122); params = [(softplus(h), b, tanh(a) | None)] x 4"""
        for sp_h, b, th_a in params:
            x = x * sp_h + b
            if th_a is not None:
                x = x + torch.tanh(x) * th_a
        return torch.sigmoid(x)

    def z_bits(self, z, params):
        import math
        probs = self.bitparm_cdf(z + 0.5, params) - self.bitparm_cdf(z - 0.5, params)
        bits = -1.0 * torch.log(probs + 1e-5) / math.log(2.0)
        return torch.clamp_min(bits, 0)

    def total(self, t):
        """sum of a tensor of per-element bits / squared errors, as the reference takes it (f32 torch.sum)"""
        return float(torch.sum(t))


class CdefK(TorchK):
    name = "cdef"

    def conv2d(self, x, w, b, stride=1, padding=0, groups=1, rule=0):
        pad = padding if isinstance(padding, (tuple, list)) else (padding, padding)
        stride = stride[0] if isinstance(stride, (tuple, list)) else stride
        xn = x.detach().numpy()
        wn = w.detach().numpy()
        bn = None if b is None else b.detach().numpy()
        if groups == 1:
            return _t(clib.conv2d(xn, wn, bn, stride, pad, rule))
        assert groups == x.size(1) == w.size(0) and w.size(1) == 1 and stride == 1 and pad[0] == w.size(2) // 2
        return _t(clib.dwconv2d(xn, wn, bn))

    def tanh(self, x):
        return _t(clib.tanh(x.numpy()))

    aten_threads = 0       # > 0: torch.sigmoid with that many intra-op threads (scalar tails of the threads' slices)

    def sigmoid(self, x):
        return _t(clib.sigmoid(x.contiguous().numpy(), self.aten_threads))

    def sigmoid_plain(self, x):
        return _t(clib.sigmoid(x.contiguous().numpy()))

    def flow_warp(self, im, flow):
        N, _, H, W = im.shape
        lx, ly = self.lin_tables(H, W)
        if flow.size(0) != 1 and flow.size(0) != N:
            raise ValueError("flow batch")
        return _t(clib.flow_warp(im.numpy(), flow.numpy(), lx.numpy(), ly.numpy()))

    def avg_pool2(self, x):
        return _t(clib.avgpool2(x.numpy()))

    def bilinear_up2(self, x, f=2):
        return _t(clib.bilinear_up(x.numpy(), f))

    def bilinear_down2(self, x, f=2):
        return _t(clib.bilinear_down(x.numpy(), f))

    def build_indexes(self, tables, scales):
        return tables.build_indexes_cdef(scales)

    # ---- estimate mode, PM-F32: every line one rounding; expm1(x) := pm_exp(x) - 1; totals are f64 sums ------------
    def _neglog2(self, probs):
        import math
        bits = -1.0 * _t(clib.log((probs + 1e-5).numpy())) / math.log(2.0)
        return torch.clamp_min(bits, 0)

    def laplace_bits(self, y, sigma):
        sigma = sigma.clamp(1e-5, 1e10)

        def cdf(v):
            e = _t(clib.exp((-v.abs() / sigma).numpy())) - 1.0
            return 0.5 - 0.5 * v.sign() * e
        return self._neglog2(cdf(y + 0.5) - cdf(y - 0.5))

    def bitparm_cdf(self, x, params):
        for sp_h, b, th_a in params:
            x = x * sp_h + b
            if th_a is not None:
                x = x + self.tanh(x.contiguous()) * th_a
        return self.sigmoid_plain(x.contiguous())     # rate estimate only: no claim on the threads' scalar tails

    def z_bits(self, z, params):
        return self._neglog2(self.bitparm_cdf(z + 0.5, params) - self.bitparm_cdf(z - 0.5, params))

    def total(self, t):
        return float(t.double().sum())
