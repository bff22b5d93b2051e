"""ORACLE — test infrastructure only; never imported by the product path.

Entropy-model tables and the symbol -> (int16 symbol, int16 CDF row) hand-off,
restated from pMCTF/entropy_models/entropy_models.py.  Table building is one-off
host work done with torch CPU tensor ops exactly as the reference does (the
Laplace CDF is torch.distributions' own closed form), quantised by the C
restatement of ops.cpp (clib.pmf_to_quantized_cdf).
"""
import math

import numpy as np
import torch

from . import clib


def laplace_cdf(x, scale):
    # torch.distributions.laplace.Laplace(0, scale).cdf(x)
    return 0.5 - 0.5 * x.sign() * torch.expm1(-x.abs() / scale)


def pmf_to_cdf(pmf, tail_mass, pmf_length, max_length, quantizer=None):
    """entropy_models.py:24-32"""
    quantizer = quantizer or clib.pmf_to_quantized_cdf
    cdf = np.zeros((len(pmf_length), int(max_length) + 2), dtype=np.int32)
    for i in range(len(pmf_length)):
        prob = torch.cat((pmf[i][: int(pmf_length[i])], tail_mass[i]), dim=0)
        q = quantizer(prob.numpy().astype(np.float32), 16)
        cdf[i, : len(q)] = np.asarray(q, dtype=np.int64).astype(np.int32)
    return cdf


class GaussianTables:
    """GaussianEncoder(distribution='laplace'), entropy_models.py:203-273."""

    def __init__(self, quantizer=None):
        self.scale_min, self.scale_max, self.scale_level = 0.01, 64.0, 256
        self.scale_table = torch.exp(torch.linspace(math.log(self.scale_min), math.log(self.scale_max),
                                                    self.scale_level))
        self.log_scale_min = math.log(self.scale_min)
        self.log_scale_max = math.log(self.scale_max)
        self.log_scale_step = (self.log_scale_max - self.log_scale_min) / (self.scale_level - 1)
        # update(): :228-267
        pmf_center = torch.zeros_like(self.scale_table) + 50
        scales = torch.zeros_like(pmf_center) + self.scale_table
        for i in range(50, 1, -1):
            samples = torch.zeros_like(pmf_center) + i
            probs = laplace_cdf(samples, scales)
            pmf_center = torch.where(probs > torch.zeros_like(pmf_center) + 0.9999,
                                     torch.zeros_like(pmf_center) + i, pmf_center)
        pmf_center = pmf_center.int()
        pmf_length = 2 * pmf_center + 1
        max_length = torch.max(pmf_length).item()
        samples = (torch.arange(max_length) - pmf_center[:, None]).float()
        scales = torch.zeros_like(samples) + self.scale_table[:, None]
        upper = laplace_cdf(samples + 0.5, scales)
        lower = laplace_cdf(samples - 0.5, scales)
        pmf = upper - lower
        tail_mass = 2 * lower[:, :1]
        self.cdf = pmf_to_cdf(pmf, tail_mass, pmf_length, max_length, quantizer)
        self.cdf_length = (pmf_length + 2).reshape(-1).int().numpy()
        self.offset = (-pmf_center).reshape(-1).int().numpy()

    def cdf_info(self):
        return self.cdf, self.cdf_length, self.offset

    def build_indexes_torch(self, scales):
        """entropy_models.py:269-273, torch CPU arithmetic (backend 'torch')."""
        scales = torch.maximum(scales, torch.zeros_like(scales) + 1e-5)
        indexes = (torch.log(scales) - self.log_scale_min) / self.log_scale_step
        indexes = indexes.clamp_(0, self.scale_level - 1)
        return indexes.int()

    def build_indexes_cdef(self, scales):
        """Same formula in PM-F32: idx = trunc(clamp((pm_log(max(s,1e-5)) - lmin) / step, 0, 255)),
        f32 subtraction and IEEE f32 division by the f32-rounded constants."""
        s = np.maximum(scales.numpy().astype(np.float32), np.float32(1e-5))
        lg = clib.log(s)
        idx = (lg - np.float32(self.log_scale_min)) / np.float32(self.log_scale_step)
        idx = np.clip(idx, np.float32(0), np.float32(self.scale_level - 1))
        return torch.from_numpy(idx.astype(np.int32))


def bitparm(x, h, b, a):
    """Bitparm.forward, entropy_models.py:73-78"""
    x = x * torch.nn.functional.softplus(h) + b
    if a is None:
        return x
    return x + torch.tanh(x) * torch.tanh(a)


class BitEstimatorTables:
    """BitEstimator.update + build_indexes, entropy_models.py:102-187 (factorized prior of mv_z)."""

    def __init__(self, sd, prefix, channel=64, quantizer=None):
        self.channel = channel
        p = lambda n: sd[prefix + n].detach().float()
        self.params = [(p("f1.h"), p("f1.b"), p("f1.a")), (p("f2.h"), p("f2.b"), p("f2.a")),
                       (p("f3.h"), p("f3.b"), p("f3.a")), (p("f4.h"), p("f4.b"), None)]
        medians = torch.zeros(channel)
        minima = medians + 50
        for i in range(50, 1, -1):
            samples = (torch.zeros_like(medians) - i)[None, :, None, None]
            probs = torch.squeeze(self.get_cdf(samples))
            minima = torch.where(probs < torch.zeros_like(medians) + 0.0001, torch.zeros_like(medians) + i, minima)
        maxima = medians + 50
        for i in range(50, 1, -1):
            samples = (torch.zeros_like(medians) + i)[None, :, None, None]
            probs = torch.squeeze(self.get_cdf(samples))
            maxima = torch.where(probs > torch.zeros_like(medians) + 0.9999, torch.zeros_like(medians) + i, maxima)
        minima, maxima = minima.int(), maxima.int()
        offset = -minima
        pmf_start = medians - minima
        pmf_length = maxima + minima + 1
        max_length = pmf_length.max()
        samples = torch.arange(max_length)
        samples = samples[None, :] + pmf_start[:, None, None]
        lower = self.get_cdf(samples - 0.5).squeeze(0)
        upper = self.get_cdf(samples + 0.5).squeeze(0)
        pmf = (upper - lower)[:, 0, :]
        tail_mass = lower[:, 0, :1] + (1.0 - upper[:, 0, -1:])
        self.cdf = pmf_to_cdf(pmf, tail_mass, pmf_length, max_length, quantizer)
        self.cdf_length = (pmf_length + 2).reshape(-1).int().numpy()
        self.offset = offset.reshape(-1).int().numpy()

    def get_cdf(self, x):
        for h, b, a in self.params:
            x = bitparm(x, h, b, a)
        return torch.sigmoid(x)

    def cdf_info(self):
        return self.cdf, self.cdf_length, self.offset

    @staticmethod
    def build_indexes(size):
        N, Cc, H, W = size
        return torch.arange(Cc, dtype=torch.int).view(1, -1, 1, 1).repeat(N, 1, H, W)


class EntropyCoder:
    """EntropyCoder facade (entropy_models.py:9-55) over the C restatement of the range coder.
    `trace` (optional list) records every (symbols int16, indexes int16) push for parity tests."""

    def __init__(self, trace=None):
        self.encoder = clib.RansEncoder()
        self.decoder = clib.RansDecoder()
        self.trace = trace

    def reset(self):
        self.encoder.reset()

    def encode_with_indexes(self, symbols, indexes, cdf, cdf_length, offset):
        s = symbols.clamp(-30000, 30000).to(torch.int16).numpy().reshape(-1)
        i = indexes.to(torch.int16).numpy().reshape(-1)
        if self.trace is not None:
            self.trace.append((s.copy(), i.copy()))
        self.encoder.encode_with_indexes(s, i, cdf, cdf_length, offset)

    def flush(self):
        self.encoder.flush()

    def get_encoded_stream(self):
        return self.encoder.get_encoded_stream().tobytes()

    def set_stream(self, stream):
        self.decoder.set_stream(stream)

    def decode_stream(self, indexes, cdf, cdf_length, offset):
        rv = self.decoder.decode_stream(indexes.to(torch.int16).numpy().reshape(-1), cdf, cdf_length, offset)
        return torch.from_numpy(rv.astype(np.float32))
