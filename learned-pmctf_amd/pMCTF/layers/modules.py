"""Parameter containers of the pMCTF model tree.

These nn.Modules exist to own parameters/buffers under exactly the names and shapes of the
reference's modules (so that a reference checkpoint loads with strict=True, test_pMCTF_flex.py:365)
— see tests/golden/state_dict_keys_me*.json.  They carry no forward(): the numeric path is
pMCTF.hip.engine.HipEngine, which reads the packed weights by key prefix.
Reference definitions: pMCTF/layers/{lifting_1d,wavelet_transform,long_context,postprocessing,
context_fusion,context_fusion_4step,layers}.py and pMCTF/layers/video/{layers,video_net,
wavelet_transform_temporal_mctf}.py.
"""
import torch
from torch import nn


def conv(cin, cout, k, stride=1, padding=None):
    if padding is None:
        padding = k // 2 if isinstance(k, int) else 0
    return nn.Conv2d(cin, cout, k, stride=stride, padding=padding)


class MaskedConv2d(nn.Conv2d):
    """PixelCNN mask buffer 'A' (centre excluded) / 'B' (centre kept): layers/layers.py:22-51"""

    def __init__(self, cin, cout, mask_type="A"):
        super().__init__(cin, cout, 3, padding=1)
        mask = torch.ones_like(self.weight.data)
        mask[:, :, 1, 1 + (mask_type == "B"):] = 0
        mask[:, :, 2:] = 0
        self.register_buffer("mask", mask)


class PredictUpdate(nn.Module):          # lifting_1d.py:25-34
    def __init__(self):
        super().__init__()
        self.conv1 = conv(1, 16, 3)
        self.conv2 = conv(16, 16, 3)
        self.conv3 = conv(16, 16, 3)
        self.conv4 = conv(16, 1, 3)


class iWave1D(nn.Module):                # lifting_1d.py:52-101
    COEFFS = (-1.586134342059924, -0.052980118572961, 0.882911075530934, 0.443506852043971)

    def __init__(self):
        super().__init__()
        c = self.COEFFS
        for name, taps in (("conv_P1", (0.0, c[0], c[0])), ("conv_U1", (c[1], c[1], 0.0)),
                           ("conv_P2", (0.0, c[2], c[2])), ("conv_U2", (c[3], c[3], 0.0))):
            m = nn.Conv2d(1, 1, (3, 1))
            m.weight.data = torch.tensor(taps, dtype=torch.float32).view(1, 1, 3, 1)
            setattr(self, name, m)
        self.P_1, self.P_2, self.U_1, self.U_2 = PredictUpdate(), PredictUpdate(), PredictUpdate(), PredictUpdate()


class LiftingScheme2D(nn.Module):        # wavelet_transform.py:8-21 (lift_v aliases lift_h)
    def __init__(self):
        super().__init__()
        self.lift_h = iWave1D()
        self.lift_v = self.lift_h


class LSTM2D(nn.Module):                 # long_context.py:8-14
    def __init__(self, cin, hidden):
        super().__init__()
        self.conv_in = conv(cin, hidden, 3)
        self.conv_hidden = conv(hidden, hidden, 3)


class UpsampleModule(nn.Module):         # long_context.py:41-58 (mode="nearest")
    def __init__(self, ch):
        super().__init__()
        self.conv = conv(ch, ch, 3)


class SubbandContext(nn.Module):         # long_context.py:64-101
    def __init__(self, decomp_levels=4, hidden=32):
        super().__init__()
        self.LSTM1, self.LSTM2, self.LSTM3 = LSTM2D(1, hidden), LSTM2D(hidden, hidden), LSTM2D(hidden, 3)
        for name, ch in (("deconv_h1", hidden), ("deconv_c1", hidden), ("deconv_h2", hidden), ("deconv_c2", hidden),
                         ("deconv_h3", 3), ("deconv_c3", 3)):
            setattr(self, name, nn.ModuleList(UpsampleModule(ch) for _ in range(decomp_levels - 1)))


class ResBlock(nn.Module):               # postprocessing.py:6-18 / context_fusion_4step.py:9-20
    def __init__(self, ch):
        super().__init__()
        self.conv1 = conv(ch, ch, 3)
        self.conv2 = conv(ch, ch, 3)


class PostProcess(nn.Module):            # postprocessing.py:21-33
    def __init__(self, ch=64):
        super().__init__()
        self.resBlocks = nn.ModuleList(ResBlock(ch) for _ in range(6))
        self.conv1 = conv(1, ch, 3)
        self.conv2 = conv(ch, ch, 3)
        self.conv3 = conv(ch, 1, 3)


class DepthConv(nn.Module):              # video/layers.py:108-126
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(cin, cin, 1), nn.LeakyReLU(0.01))
        self.depth_conv = nn.Conv2d(cin, cin, 3, padding=1, groups=cin)
        self.conv2 = nn.Conv2d(cin, cout, 1)
        self.adaptor = nn.Conv2d(cin, cout, 1) if cin != cout else None


class ConvFFN(nn.Module):                # video/layers.py:139-149
    def __init__(self, ch):
        super().__init__()
        internal = max(min(ch * 4, 1024), ch * 2)
        self.conv = nn.Sequential(nn.Conv2d(ch, internal, 1), nn.LeakyReLU(0.1), nn.Conv2d(internal, ch, 1),
                                  nn.LeakyReLU(0.1))


class ConvFFN3(nn.Module):               # video/layers.py:155-161
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch * 4, 1)
        self.conv_out = nn.Conv2d(ch * 2, ch, 1)


class DepthConvBlock(nn.Module):         # video/layers.py:171-178
    def __init__(self, cin, cout):
        super().__init__()
        self.block = nn.Sequential(DepthConv(cin, cout), ConvFFN(cout))


class DepthConvBlock4(nn.Module):        # video/layers.py:184-190
    def __init__(self, cin, cout):
        super().__init__()
        self.block = nn.Sequential(DepthConv(cin, cout), ConvFFN3(cout))


class ResidualBlockWithStride(nn.Module):  # video/layers.py:46-62
    def __init__(self, cin, cout, stride=2):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride=stride, padding=1)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.downsample = nn.Conv2d(cin, cout, 1, stride=stride) if stride != 1 else None


def subpel_conv1x1(cin, cout, r):        # video/layers.py:34-38
    return nn.Sequential(nn.Conv2d(cin, cout * r * r, 1), nn.PixelShuffle(r))


class ResidualBlockUpsample(nn.Module):  # video/layers.py:80-94
    def __init__(self, cin, cout, r=2):
        super().__init__()
        self.subpel_conv = subpel_conv1x1(cin, cout, r)
        self.conv = nn.Conv2d(cout, cout, 3, padding=1)
        self.upsample = subpel_conv1x1(cin, cout, r)


class ContextFusionFourStep(nn.Module):  # context_fusion_4step.py:23-90
    def __init__(self, ctx_channels, nf=112, nparams=2):
        super().__init__()
        self.y_hierarchical_prior_enc = nn.Sequential(ResBlock(nf), ResBlock(nf))
        self.conv1_context = conv(ctx_channels, nf, 3)
        if ctx_channels > 1:
            self.lower_level_subband = nn.Sequential(nn.Upsample(scale_factor=2, mode="nearest"), conv(1, 1, 3))
        self.y_hierarchical_prior_out = DepthConvBlock(nf, nparams)
        for k in (1, 2, 3):
            setattr(self, f"y_spatial_prior_{k}", nn.Sequential(conv(1, nf, 3), ResBlock(nf)))
            setattr(self, f"y_spatial_prior_{k}_out", nn.Sequential(ResBlock(nf), ResBlock(nf), conv(nf, nparams, 1)))


class MaskResidual(nn.Module):           # context_fusion.py:8-16
    def __init__(self, nf):
        super().__init__()
        self.conv1 = MaskedConv2d(nf, nf, "B")
        self.conv2 = MaskedConv2d(nf, nf, "B")


class ContextFusionSubband(nn.Module):   # context_fusion.py:56-98 (context=False)
    def __init__(self, nf=128, nparams=2):
        super().__init__()
        self.maskedConv1 = MaskedConv2d(1, nf, "A")
        self.residualBlocks = nn.ModuleList(MaskResidual(nf) for _ in range(2))
        self.maskedConv2 = MaskedConv2d(nf, nf, "B")
        self.convs = nn.ModuleList([conv(nf, nf, 1), conv(nf, nf, 1), conv(nf, nparams, 1)])


class MEBasic(nn.Module):                # video_net.py:74-82
    def __init__(self):
        super().__init__()
        self.conv1 = conv(8, 32, 7)
        self.conv2 = conv(32, 64, 7)
        self.conv3 = conv(64, 32, 7)
        self.conv4 = conv(32, 16, 7)
        self.conv5 = conv(16, 2, 7)


class ME_Spynet(nn.Module):              # video_net.py:93-97
    def __init__(self, L=6):
        super().__init__()
        self.L = L
        self.moduleBasic = nn.ModuleList(MEBasic() for _ in range(L))


class MvEnc(nn.Module):                  # video_net.py:124-139
    def __init__(self, cin=2, ch=64):
        super().__init__()
        self.enc_1 = nn.Sequential(ResidualBlockWithStride(cin, ch), DepthConvBlock(ch, ch))
        self.enc_2 = ResidualBlockWithStride(ch, ch)
        self.adaptor_0 = DepthConvBlock(ch, ch)
        self.adaptor_1 = DepthConvBlock(ch * 2, ch)
        self.enc_3 = nn.Sequential(ResidualBlockWithStride(ch, ch), DepthConvBlock(ch, ch),
                                   nn.Conv2d(ch, ch, 3, stride=2, padding=1))


class MvDec(nn.Module):                  # video_net.py:152-166
    def __init__(self, cout=2, ch=64):
        super().__init__()
        self.dec_1 = nn.Sequential(DepthConvBlock(ch, ch), ResidualBlockUpsample(ch, ch), DepthConvBlock(ch, ch),
                                   ResidualBlockUpsample(ch, ch), DepthConvBlock(ch, ch))
        self.dec_2 = ResidualBlockUpsample(ch, ch)
        self.dec_3 = nn.Sequential(DepthConvBlock(ch, ch), subpel_conv1x1(ch, cout, 2))


def get_hyper_enc_model(n, mv):          # video_net.py:176-183
    return nn.Sequential(DepthConvBlock4(mv, n), nn.Conv2d(n, n, 3, stride=2, padding=1), nn.LeakyReLU(),
                         nn.Conv2d(n, n, 3, stride=2, padding=1))


def get_hyper_dec_model(n, mv):          # video_net.py:185-191
    return nn.Sequential(ResidualBlockUpsample(n, n), ResidualBlockUpsample(n, n), DepthConvBlock4(n, mv))


class TemporalLifting(nn.Module):        # wavelet_transform_temporal_mctf.py:11-25
    def __init__(self):
        super().__init__()
        self.P_t = PredictUpdate()
        self.U_t = PredictUpdate()
