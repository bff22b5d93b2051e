"""pWave — learned spatial wavelet image coder (parameter tree of pMCTF/models/pWave.py:27-86).

In this implementation pWave is the owner of one coder's parameters; the numeric path
(`compress`) runs in pMCTF.hip.engine.HipEngine.pwave_compress on the GPU and is
driven from pMCTF.encode_one_stage.  A standalone pWave can still compress a plane through its own
engine (`compress`), mirroring pWave.compress(x, sideinfo, file_name, q_index, skip_decoding, qp_scale).
"""
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
