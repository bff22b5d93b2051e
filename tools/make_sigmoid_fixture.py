#!/usr/bin/env python3
"""Writes tests/golden/reference_torch_sigmoid.npz: inputs and torch.sigmoid's outputs (float32, CPU) on the machine the
other fixtures under tests/golden were generated on — the known answers that pin pm_aten_sigmoidf (pm_sleef_f32.h) for
the conv-LSTM gates (pMCTF/layers/long_context.py:24-31).  Build-container tool; needs only torch."""
import os

import numpy as np
import torch

r = np.random.default_rng(23)
x = np.concatenate([r.standard_normal(4096) * s for s in (0.05, 0.5, 2.0, 8.0, 30.0)]).astype(np.float32)
x = np.concatenate([x, np.linspace(-110, 110, 4096, dtype=np.float32),
                    np.array([0.0, -0.0, 1e-30, -1e-30, 88.0, -88.0, 104.0, -104.0] * 4, np.float32)])
assert x.size % 32 == 0 and x.size < 32768   # whole vectors, one thread's slice: ATen's scalar tail goes through libm
y = torch.sigmoid(torch.from_numpy(x)).numpy()
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_torch_sigmoid.npz")
np.savez_compressed(out, sigmoid_x=x, sigmoid_y=y)
print(out, x.size)
