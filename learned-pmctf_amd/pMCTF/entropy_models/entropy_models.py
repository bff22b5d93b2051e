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
    """What every table owner carries: the shared range coder and its (cdf rows, row lengths, symbol offsets) as host
    int32 arrays, None until update() has run."""

    def __init__(self):
        super().__init__()
        self.entropy_coder = None
        self._quantized_cdf = self._cdf_length = self._offset = None

    def set_entropy_coder(self, coder):
        self.entropy_coder = coder

    def set_cdf_info(self, quantized_cdf, cdf_length, offset):
        host = lambda t: t.detach().cpu().numpy()
        self._quantized_cdf = host(quantized_cdf)
        self._cdf_length = host(cdf_length.reshape(-1).int())
        self._offset = host(offset.reshape(-1).int())

    def get_cdf_info(self):
        return self._quantized_cdf, self._cdf_length, self._offset

    def _code(self, values, rows):
        return self.entropy_coder.encode_with_indexes(values.reshape(-1), rows.reshape(-1), *self.get_cdf_info())

    def _decode(self, rows):
        return self.entropy_coder.decode_stream(rows.reshape(-1), *self.get_cdf_info()).reshape(rows.shape)


class BitEstimator(AEHelper, nn.Module):
    """Factorized prior of the MV hyper-latent (entropy_models.py:102-200).  update() runs on the host."""

    def __init__(self, channel):
        super().__init__()
        self.channel = channel
        for k in (1, 2, 3, 4):                      # parameter names f1..f4 are part of the checkpoint layout
            setattr(self, f"f{k}", Bitparm(channel, final=(k == 4)))

    def get_logits_cdf(self, x):
        for k in (1, 2, 3, 4):
            x = getattr(self, f"f{k}")(x)
        return x

    def get_cdf(self, x):
        return torch.sigmoid(self.get_logits_cdf(x))

    forward = get_cdf

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
        """Tables of the factorized prior, built once on the host.  Support of channel c = [-lo_c, hi_c] where lo_c / hi_c
        is the first integer of 2..50 at which the CDF has fallen below 1e-4 / risen above 0.9999 (50 if none); the pmf of
        an integer is cdf(v + 0.5) - cdf(v - 0.5) and what lies outside the support is the escape symbol's mass."""
        if entropy_coder is not None:
            self.entropy_coder = entropy_coder
        if self._offset is not None and not force:
            return
        with torch.no_grad():
            cdf = self._host_cdf()
            C_ = self.channel
            grid = torch.arange(2, 51, dtype=torch.float32)                      # candidate half-widths
            at = lambda v: cdf(v.t().reshape(-1, C_, 1, 1)).reshape(-1, C_)       # rows: candidates, cols: channels
            left = at(-grid[None, :].expand(C_, -1))
            right = at(grid[None, :].expand(C_, -1))
            first = lambda hit: torch.where(hit.any(0), grid[hit.float().argmax(0)], torch.full((C_,), 50.0))
            lo = first(left < 0.0001).int()
            hi = first(right > 0.9999).int()
            width = hi + lo + 1
            longest = int(width.max())
            pos = torch.arange(longest, dtype=torch.float32)[None, :] - lo[:, None]         # (C, longest) integer values
            vals = pos.t().reshape(longest, C_, 1, 1)
            below = cdf(vals - 0.5).reshape(longest, C_).t()
            above = cdf(vals + 0.5).reshape(longest, C_).t()
            pmf = above - below
            outside = below[:, :1] + (1.0 - above[:, -1:])
            table = EntropyCoder.pmf_to_cdf(pmf, outside, width, longest)
            self.set_cdf_info(table, width + 2, -lo)

    @staticmethod
    def build_indexes(size):
        """table row of an element = its channel"""
        n, c, h, w = size
        return torch.arange(c, dtype=torch.int)[None, :, None, None].expand(n, c, h, w).contiguous()

    def encode(self, x):
        return self._code(x, self.build_indexes(x.size()))

    def decode_stream(self, size, dtype, device):
        rows = self.build_indexes((1, self.channel, size[0], size[1]))
        return self._decode(rows).to(dtype).to(device)


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
        """One table row per scale: symmetric support [-r, r] with r the first integer of 2..50 whose CDF exceeds 0.9999
        (50 if none); pmf(v) = cdf(v + 0.5) - cdf(v - 0.5); escape mass = both tails = 2 * cdf(-r - 0.5)."""
        if entropy_coder is not None:
            self.entropy_coder = entropy_coder
        if self._offset is not None and not force:
            return
        S = self.scale_table
        grid = torch.arange(2, 51, dtype=torch.float32)
        reach = self._cdf(grid[:, None].expand(-1, S.numel()), S[None, :].expand(grid.numel(), -1)) > 0.9999
        radius = torch.where(reach.any(0), grid[reach.float().argmax(0)], torch.full_like(S, 50.0)).int()
        width = 2 * radius + 1
        longest = int(width.max())
        vals = (torch.arange(longest) - radius[:, None]).float()                  # (rows, longest)
        sc = S[:, None].expand(-1, longest)
        above = self._cdf(vals + 0.5, sc + torch.zeros_like(vals))
        below = self._cdf(vals - 0.5, sc + torch.zeros_like(vals))
        table = EntropyCoder.pmf_to_cdf(above - below, 2 * below[:, :1], width, longest)
        self.set_cdf_info(table, width + 2, -radius)

    def build_indexes(self, scales):
        """Host (torch) form, kept for API parity; the encode path computes indexes on the GPU
        (pmctf_fourstep_quant_f32 & co) with the same formula in PM-F32 arithmetic."""
        scales = torch.maximum(scales, torch.zeros_like(scales) + 1e-5)
        indexes = (torch.log(scales) - self.log_scale_min) / self.log_scale_step
        return indexes.clamp_(0, self.scale_level - 1).int()

    def encode(self, x, scales):
        return self._code(x, self.build_indexes(scales))

    def decode_stream(self, scales, dtype, device):
        return self._decode(self.build_indexes(scales)).to(device).to(dtype)
