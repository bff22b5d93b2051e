"""Entropy-model tables and the host range-coder facade of the MI355X pMCTF path.

Same public names as the reference module (pMCTF/entropy_models/entropy_models.py): EntropyCoder,
Bitparm, BitEstimator, GaussianEncoder.  Table building is one-off host work (torch CPU tensor
ops); symbols reach the coder as int16 arrays produced by the HIP quantisation kernels.  The coder
itself is libpmctf_rans.so (C ABI include/pmctf_rans.h) instead of the pybind11 modules
MLCodec_rans / MLCodec_CXX.
"""
import ctypes as C
import math

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from pMCTF.hip import lib as _lib


def pmf_to_quantized_cdf(pmf, precision=16):
    """MLCodec_CXX.pmf_to_quantized_cdf (ops.cpp:24-82) through the C ABI."""
    p = np.ascontiguousarray(np.asarray(pmf, dtype=np.float32))
    cdf = np.zeros(p.size + 1, np.uint32)
    _lib.check(_lib.rans().pmctf_pmf_to_quantized_cdf(p.ctypes.data, p.size, precision, cdf.ctypes.data),
               "pmf_to_quantized_cdf")
    return cdf


class EntropyCoder:
    """entropy_models.py:9-55 — numpy/int16 facade over the range coder."""

    def __init__(self, ec_thread=False, stream_part=1):
        R = _lib.rans()
        self._R = R
        self.encoder = R.pmctf_rans_encoder_create(int(bool(ec_thread)), int(stream_part))
        self.decoder = R.pmctf_rans_decoder_create(int(stream_part))
        if not self.encoder or not self.decoder:
            raise RuntimeError("could not create range coder")

    def __del__(self):
        R = getattr(self, "_R", None)
        if R is not None:
            if getattr(self, "encoder", None):
                R.pmctf_rans_encoder_destroy(self.encoder)
            if getattr(self, "decoder", None):
                R.pmctf_rans_decoder_destroy(self.decoder)
            self.encoder = self.decoder = None

    @staticmethod
    def pmf_to_quantized_cdf(pmf, precision=16):
        return torch.from_numpy(pmf_to_quantized_cdf(pmf.tolist(), precision).astype(np.int64)).int()

    @staticmethod
    def pmf_to_cdf(pmf, tail_mass, pmf_length, max_length):
        cdf = torch.zeros((len(pmf_length), int(max_length) + 2), dtype=torch.int32)
        for i, p in enumerate(pmf):
            prob = torch.cat((p[: int(pmf_length[i])], tail_mass[i]), dim=0)
            _cdf = EntropyCoder.pmf_to_quantized_cdf(prob, 16)
            cdf[i, : _cdf.size(0)] = _cdf
        return cdf

    def reset(self):
        _lib.check(self._R.pmctf_rans_encoder_reset(self.encoder), "rans reset")

    @staticmethod
    def _i16(t):
        if isinstance(t, torch.Tensor):
            t = t.detach().to("cpu").numpy()
        return np.ascontiguousarray(t, dtype=np.int16).reshape(-1)

    def encode_with_indexes(self, symbols, indexes, cdf, cdf_length, offset):
        if isinstance(symbols, torch.Tensor):
            symbols = symbols.clamp(-30000, 30000)
        s, i = self._i16(symbols), self._i16(indexes)
        cdf = np.ascontiguousarray(cdf, dtype=np.int32)
        sizes = np.ascontiguousarray(cdf_length, dtype=np.int32)
        offs = np.ascontiguousarray(offset, dtype=np.int32)
        _lib.check(self._R.pmctf_rans_encoder_encode_with_indexes(
            self.encoder, s.ctypes.data, i.ctypes.data, s.size, cdf.ctypes.data, cdf.shape[0], cdf.shape[1],
            sizes.ctypes.data, offs.ctypes.data), "rans encode_with_indexes")

    def flush(self):
        _lib.check(self._R.pmctf_rans_encoder_flush(self.encoder), "rans flush")

    def get_encoded_stream(self):
        n = self._R.pmctf_rans_encoder_stream_size(self.encoder)
        buf = np.empty(n, np.uint8)
        _lib.check(self._R.pmctf_rans_encoder_get_encoded_stream(self.encoder, buf.ctypes.data, n), "rans get stream")
        return buf.tobytes()

    def set_stream(self, stream):
        s = np.frombuffer(bytes(stream), dtype=np.uint8).copy()
        _lib.check(self._R.pmctf_rans_decoder_set_stream(self.decoder, s.ctypes.data, s.size), "rans set_stream")

    def decode_stream(self, indexes, cdf, cdf_length, offset):
        i = self._i16(indexes)
        cdf = np.ascontiguousarray(cdf, dtype=np.int32)
        sizes = np.ascontiguousarray(cdf_length, dtype=np.int32)
        offs = np.ascontiguousarray(offset, dtype=np.int32)
        out = np.empty(i.size, np.int16)
        _lib.check(self._R.pmctf_rans_decoder_decode_stream(
            self.decoder, i.ctypes.data, i.size, cdf.ctypes.data, cdf.shape[0], cdf.shape[1], sizes.ctypes.data,
            offs.ctypes.data, out.ctypes.data), "rans decode_stream")
        return torch.from_numpy(out.astype(np.float32))


class Bitparm(nn.Module):
    """entropy_models.py:58-78"""

    def __init__(self, channel, final=False):
        super().__init__()
        self.final = final
        mk = lambda: nn.Parameter(torch.nn.init.normal_(torch.empty(channel).view(1, -1, 1, 1), 0, 0.01))
        self.h = mk()
        self.b = mk()
        self.a = None if final else mk()

    def forward(self, x):
        x = x * F.softplus(self.h) + self.b
        if self.final:
            return x
        return x + torch.tanh(x) * torch.tanh(self.a)


class AEHelper:
    def __init__(self):
        super().__init__()
        self.entropy_coder = None
        self._offset = None
        self._quantized_cdf = None
        self._cdf_length = None

    def set_entropy_coder(self, coder):
        self.entropy_coder = coder

    def set_cdf_info(self, quantized_cdf, cdf_length, offset):
        self._quantized_cdf = quantized_cdf.cpu().numpy()
        self._cdf_length = cdf_length.reshape(-1).int().cpu().numpy()
        self._offset = offset.reshape(-1).int().cpu().numpy()

    def get_cdf_info(self):
        return self._quantized_cdf, self._cdf_length, self._offset


class BitEstimator(AEHelper, nn.Module):
    """Factorized prior of the MV hyper-latent (entropy_models.py:102-200).  update() runs on the host."""

    def __init__(self, channel):
        super().__init__()
        self.f1 = Bitparm(channel)
        self.f2 = Bitparm(channel)
        self.f3 = Bitparm(channel)
        self.f4 = Bitparm(channel, True)
        self.channel = channel

    def forward(self, x):
        return self.get_cdf(x)

    def get_logits_cdf(self, x):
        return self.f4(self.f3(self.f2(self.f1(x))))

    def get_cdf(self, x):
        return torch.sigmoid(self.get_logits_cdf(x))

    def _host_cdf(self):
        ps = [(f.h.detach().cpu().float(), f.b.detach().cpu().float(), None if f.a is None else f.a.detach().cpu().float())
              for f in (self.f1, self.f2, self.f3, self.f4)]

        def cdf(x):
            for h, b, a in ps:
                x = x * F.softplus(h) + b
                if a is not None:
                    x = x + torch.tanh(x) * torch.tanh(a)
            return torch.sigmoid(x)
        return cdf

    def update(self, force=False, entropy_coder=None):
        if entropy_coder is not None:
            self.entropy_coder = entropy_coder
        if not force and self._offset is not None:
            return
        with torch.no_grad():
            cdf = self._host_cdf()
            medians = torch.zeros(self.channel)
            minima = medians + 50
            for i in range(50, 1, -1):
                probs = torch.squeeze(cdf((torch.zeros_like(medians) - i)[None, :, None, None]))
                minima = torch.where(probs < torch.zeros_like(medians) + 0.0001, torch.zeros_like(medians) + i, minima)
            maxima = medians + 50
            for i in range(50, 1, -1):
                probs = torch.squeeze(cdf((torch.zeros_like(medians) + i)[None, :, None, None]))
                maxima = torch.where(probs > torch.zeros_like(medians) + 0.9999, torch.zeros_like(medians) + i, maxima)
            minima, maxima = minima.int(), maxima.int()
            offset = -minima
            pmf_start = medians - minima
            pmf_length = maxima + minima + 1
            max_length = pmf_length.max()
            samples = torch.arange(max_length)[None, :] + pmf_start[:, None, None]
            lower = cdf(samples - 0.5).squeeze(0)
            upper = cdf(samples + 0.5).squeeze(0)
            pmf = (upper - lower)[:, 0, :]
            tail_mass = lower[:, 0, :1] + (1.0 - upper[:, 0, -1:])
            quantized_cdf = EntropyCoder.pmf_to_cdf(pmf, tail_mass, pmf_length, max_length)
            self.set_cdf_info(quantized_cdf, pmf_length + 2, offset)

    @staticmethod
    def build_indexes(size):
        N, Cc, H, W = size
        return torch.arange(Cc, dtype=torch.int).view(1, -1, 1, 1).repeat(N, 1, H, W)

    def encode(self, x):
        indexes = self.build_indexes(x.size())
        return self.entropy_coder.encode_with_indexes(x.reshape(-1), indexes.reshape(-1), *self.get_cdf_info())

    def decode_stream(self, size, dtype, device):
        output_size = (1, self.channel, size[0], size[1])
        indexes = self.build_indexes(output_size)
        val = self.entropy_coder.decode_stream(indexes.reshape(-1), *self.get_cdf_info())
        return val.reshape(indexes.shape).to(dtype).to(device)


class GaussianEncoder(AEHelper):
    """Laplace tables: 256 log-spaced scales 0.01..64 (entropy_models.py:203-285)."""

    def __init__(self, distribution="laplace"):
        super().__init__()
        assert distribution in ("laplace", "gaussian")
        self.distribution = distribution
        if distribution == "laplace":
            self.scale_min, self.scale_max, self.scale_level = 0.01, 64.0, 256
        else:
            self.scale_min, self.scale_max, self.scale_level = 0.11, 64.0, 256
        self.scale_table = torch.exp(torch.linspace(math.log(self.scale_min), math.log(self.scale_max),
                                                    self.scale_level))
        self.log_scale_min = math.log(self.scale_min)
        self.log_scale_max = math.log(self.scale_max)
        self.log_scale_step = (self.log_scale_max - self.log_scale_min) / (self.scale_level - 1)

    def _cdf(self, x, scales):
        if self.distribution == "laplace":
            return torch.distributions.laplace.Laplace(torch.zeros_like(scales), scales).cdf(x)
        return torch.distributions.normal.Normal(torch.zeros_like(scales), scales).cdf(x)

    def update(self, force=False, entropy_coder=None):
        if entropy_coder is not None:
            self.entropy_coder = entropy_coder
        if not force and self._offset is not None:
            return
        pmf_center = torch.zeros_like(self.scale_table) + 50
        scales = torch.zeros_like(pmf_center) + self.scale_table
        for i in range(50, 1, -1):
            probs = torch.squeeze(self._cdf(torch.zeros_like(pmf_center) + i, scales))
            pmf_center = torch.where(probs > torch.zeros_like(pmf_center) + 0.9999, torch.zeros_like(pmf_center) + i,
                                     pmf_center)
        pmf_center = pmf_center.int()
        pmf_length = 2 * pmf_center + 1
        max_length = torch.max(pmf_length).item()
        samples = (torch.arange(max_length) - pmf_center[:, None]).float()
        scales = torch.zeros_like(samples) + self.scale_table[:, None]
        upper = self._cdf(samples + 0.5, scales)
        lower = self._cdf(samples - 0.5, scales)
        pmf = upper - lower
        tail_mass = 2 * lower[:, :1]
        quantized_cdf = EntropyCoder.pmf_to_cdf(pmf, tail_mass, pmf_length, max_length)
        self.set_cdf_info(quantized_cdf, pmf_length + 2, -pmf_center)

    def build_indexes(self, scales):
        """Host (torch) form, kept for API parity; the encode path computes indexes on the GPU
        (pmctf_fourstep_quant_f32 & co) with the same formula in PM-F32 arithmetic."""
        scales = torch.maximum(scales, torch.zeros_like(scales) + 1e-5)
        indexes = (torch.log(scales) - self.log_scale_min) / self.log_scale_step
        return indexes.clamp_(0, self.scale_level - 1).int()

    def encode(self, x, scales):
        indexes = self.build_indexes(scales)
        return self.entropy_coder.encode_with_indexes(x.reshape(-1), indexes.reshape(-1), *self.get_cdf_info())

    def decode_stream(self, scales, dtype, device):
        indexes = self.build_indexes(scales)
        val = self.entropy_coder.decode_stream(indexes.reshape(-1), *self.get_cdf_info())
        return val.reshape(scales.shape).to(device).to(dtype)
