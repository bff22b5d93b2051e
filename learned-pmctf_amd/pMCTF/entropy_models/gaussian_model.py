"""CompressionModel: the entropy-coding context a coder owns — the Laplace scale tables and, after update(), one host
range coder (interface of pMCTF/entropy_models/gaussian_model.py:13-72; the bit ESTIMATES of that class live on the GPU
in this build, csrc/estimate_ops.hip)."""
from torch import nn

from .entropy_models import EntropyCoder, GaussianEncoder


class CompressionModel(nn.Module):
    def __init__(self, y_distribution, ec_thread=False, stream_part=1):
        super().__init__()
        if y_distribution not in ("laplace", "gaussian"):
            raise ValueError(f"unknown distribution {y_distribution!r}")
        self._coder_args = (bool(ec_thread), int(stream_part))
        self.y_distribution = y_distribution
        self.gaussian_encoder = GaussianEncoder(distribution=y_distribution)
        self.entropy_coder = None           # created by update()

    @property
    def ec_thread(self):
        return self._coder_args[0]

    @property
    def stream_part(self):
        return self._coder_args[1]

    def update(self, force=False):
        """(Re)create the range coder and (re)build the tables; every table user of the model shares this coder."""
        coder = EntropyCoder(*self._coder_args)
        self.gaussian_encoder.update(force=force, entropy_coder=coder)
        self.entropy_coder = coder
