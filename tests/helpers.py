"""Shared test helpers: synthetic model/inputs for oracle and product."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name="reference_128x128.npz"):
    return np.load(os.path.join(GOLDEN, name))


def product_model(num_me_stages=1, device="cuda", lazy=False, weights_seed=0):
    """pMCTF (HIP product) with the deterministic synthetic weights, ready to encode.  lazy=False: encode_one_stage
    returns finished results call by call (what most tests assume: they look at files and tensors right away);
    lazy=True: the product's default, pairs deferred and coded per temporal stage (pMCTF.hip.deferred)."""
    import pmctf_synth
    from pMCTF.models.video.pMCTF_L import pMCTF
    net = pMCTF(num_me_stages=num_me_stages).eval()
    sd = pmctf_synth.synth_state_dict(net.state_dict(), seed=weights_seed)
    net.load_state_dict(sd, strict=True)
    net = net.to(device)
    net.update(force=True)
    net.lazy_stages = lazy
    return net, sd


def synth_sd_cpu(num_me_stages=1):
    """The same synthetic state_dict built without touching the GPU (masks from the container modules)."""
    import pmctf_synth
    from pMCTF.models.video.pMCTF_L import pMCTF
    net = pMCTF(num_me_stages=num_me_stages)
    return pmctf_synth.synth_state_dict(net.state_dict(), seed=0)


def frames(width, height, n, device="cpu", seed=1234):
    import pmctf_synth
    f8 = pmctf_synth.synth_yuv420(width, height, n, seed=seed)
    return [list(pmctf_synth.frames_to_tensors(f, device=device)) for f in f8]


def assert_same(a, b, what):
    a, b = (x.force() if hasattr(x, "force") else x for x in (a, b))        # deferred results of the product
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    neq = a != b
    if neq.any():
        i = tuple(np.argwhere(neq)[0])
        raise AssertionError(f"{what}: {int(neq.sum())}/{a.size} elements differ; first at {i}: {a[i]!r} vs {b[i]!r}; "
                             f"max abs diff {np.abs(a.astype(np.float64) - b.astype(np.float64)).max():.3e}")


def golden_448():
    """files / bit counts / PSNR of the real reference on the 448x256 GOP-4 (tools/make_golden.py --width 448 --height 256,
    reduced to the arrays that are not full-size tensors)"""
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_448x256_files.npz"))
