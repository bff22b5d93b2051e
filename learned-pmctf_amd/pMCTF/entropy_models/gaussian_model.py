"""CompressionModel (pMCTF/entropy_models/gaussian_model.py:13-72): owns the Laplace tables and the coder."""
from torch import nn

from .entropy_models import EntropyCoder, GaussianEncoder


class CompressionModel(nn.Module):
    def __init__(self, y_distribution, ec_thread=False, stream_part=1):
        super().__init__()
        self.y_distribution = y_distribution
        self.entropy_coder = None
        self.gaussian_encoder = GaussianEncoder(distribution=y_distribution)
        self.ec_thread = ec_thread
        self.stream_part = stream_part

    def update(self, force=False):
        self.entropy_coder = EntropyCoder(self.ec_thread, self.stream_part)
        self.gaussian_encoder.update(force=force, entropy_coder=self.entropy_coder)
