"""pMCTF — MI355X-native implementation of the pMCTF temporal-decomposition encode path.

Same import paths as the reference package (pMCTF.models.video.pMCTF_L.pMCTF, pMCTF.utils.*),
compute done by hand-written gfx950 HIP kernels in ../csrc through the C ABI in ../../include.
"""
