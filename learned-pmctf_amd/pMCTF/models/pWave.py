"""pWave — learned spatial wavelet image coder (parameter tree of pMCTF/models/pWave.py:27-86).

In this implementation pWave is the owner of one coder's parameters; the numeric path
(`compress`) runs in pMCTF.hip.engine.HipEngine.pwave_compress on the GPU and is
driven from pMCTF.encode_one_stage.  A standalone pWave can still compress a plane through its own
engine (`compress`), mirroring pWave.compress(x, sideinfo, file_name, q_index, skip_decoding, qp_scale).
"""
import os

import torch
from torch import nn

from pMCTF.entropy_models.gaussian_model import CompressionModel
from pMCTF.layers.modules import (ContextFusionFourStep, ContextFusionSubband, LiftingScheme2D, PostProcess,
                                  SubbandContext)


class pWave(nn.Module):
    def __init__(self, bitdepth=8, decomp_levels=4, lossy=True):
        super().__init__()
        if not lossy or bitdepth != 8:
            raise NotImplementedError("the MI355X path implements the lossy 8-bit coder the video model uses")
        self.bitdepth = 8
        self.dynamic_range = float(2 ** bitdepth)
        self.lossy = lossy
        self.in_channels = 1
        self.decomp_levels = decomp_levels
        self.wavelet_transform = LiftingScheme2D()
        self.clip_value = 8192.
        self.context_prediction = SubbandContext(decomp_levels=decomp_levels)
        self.dequantModule = PostProcess()
        self.num_params = 2
        self.em = CompressionModel(y_distribution="laplace")
        self.context_fusion = nn.ModuleDict({
            str(lvl): nn.ModuleDict({sb: ContextFusionFourStep(ctx_channels=2 if lvl < decomp_levels - 1 else 1)
                                     for sb in ("lh", "hl", "hh")})
            for lvl in range(decomp_levels)})
        self.context_fusion[str(decomp_levels - 1)]["ll"] = ContextFusionSubband()
        self.QP = nn.Parameter(torch.ones((2, 1, 1, 1), dtype=torch.float) * 1 / 16)
        self.QP_ll = nn.Parameter(torch.ones((2, 1, 1, 1), dtype=torch.float) * 1 / 16)
        self.apply(self._init_weights)
        self._engine = None

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Conv2d) and m.weight.size(-1) == m.weight.size(-2):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)

    @staticmethod
    def get_qp_num():
        return 21

    def update(self, force=False):
        self.em.update(force)
        self._engine = None

    def get_curr_q(self, q_scale, q_index):
        """pWave.py:217-226"""
        from pMCTF.hip.engine import get_curr_q
        return get_curr_q(q_scale.detach().cpu(), q_index)

    def engine(self):
        """This coder's parameters on the GPU engine (built on first use, after update())."""
        if getattr(self, "_engine", None) is None:
            from pMCTF.hip.engine import HipEngine
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise RuntimeError("pWave (MI355X build) codes on the GPU only: move the model to 'cuda'; "
                                   "there is no CPU path")
            ge = self.em.gaussian_encoder
            if ge.get_cdf_info()[0] is None:
                raise RuntimeError("call update(force=True) before coding")
            g = {"cdf_info": ge.get_cdf_info(), "log_scale_min": ge.log_scale_min, "log_scale_step": ge.log_scale_step}
            sd = {"coder." + k: v for k, v in self.state_dict().items()}
            self._engine = HipEngine(sd, 0, dev, g, [], decomp_levels=self.decomp_levels,
                                     precision=getattr(self, "precision", None) or os.environ.get("PMCTF_PRECISION", "f32"))
        return self._engine

    @torch.no_grad()
    def forward(self, x, q_index=None, qp_scale=None):
        """pWave.py:231-312 at inference: x_hat and the Laplace bit estimates of all subbands"""
        if self.training:
            raise NotImplementedError("training forward is not part of this build")
        eng = self.engine()
        qs = None if qp_scale is None else float(qp_scale)
        x = x.contiguous().float()
        r = eng.pwave_forward("coder", x, q_index, qs)
        N, _, H, W = r["x_hat"].shape
        vals = torch.cat([r["bits"], r["sq_err"]]).cpu()
        bits, sq = vals[:N], float(vals[N])
        t = lambda v: torch.tensor(v, dtype=torch.float32)
        total = float(bits.sum())
        return {"x_hat": r["x_hat"], "bits": {"bits_total": bits.float()}, "likelihoods": {"bits_total": bits.float()},
                "subbands": r["subbands"], "bpp_total": t(total / (H * W * N)), "bits_total": t(total / N),
                "mse": t(sq / x.numel())}

    @torch.no_grad()
    def compress(self, x, sideinfo=None, file_name=None, q_index=None, skip_decoding=False, qp_scale=None):
        """pWave.py:380-464: code one image (Y, UV or RGB planes; sizes already padded) into `file_name`, return x_hat.
        With skip_decoding=False the LL subband is written in the sequential decoder's order."""
        from pMCTF.utils.stream_helper import image_header
        _, num_channels, height, width = sideinfo
        x_in = torch.cat([x[:, c:c + 1] for c in range(3)], dim=0) if num_channels == 3 else x
        eng = self.engine()
        qs = None if qp_scale is None else float(qp_scale)
        x_hat, stream = eng.pwave_compress("coder", x_in.contiguous().float(), q_index, qs, ar_order=not skip_decoding)
        hdr = lambda n: image_header(height, width, num_channels, n)
        eng.coder.submit(stream, eng.tables, hdr, file_name).result()
        if num_channels == 3:
            x_hat = torch.cat([x_hat[c:c + 1] for c in range(3)], dim=1)
        return x_hat

    @torch.no_grad()
    def decompress(self, file_name, padding=64, q_index=None, qp_scale=None):
        """pWave.py:466-529"""
        eng = self.engine()
        with open(file_name, "rb") as f:
            data = f.read()
        qs = None if qp_scale is None else float(qp_scale)
        x_hat = eng.pwave_decompress("coder", data, padding, q_index, qs)
        if x_hat.shape[0] == 3:
            x_hat = torch.cat([x_hat[c:c + 1] for c in range(3)], dim=1)
        return {"x_hat": x_hat}
