"""Oracle pinned against the known answers recorded from the real reference — CPU only."""
import hashlib

import numpy as np
import torch


def test_pmf_to_quantized_cdf_kat():
    from pmctf_oracle import clib
    assert clib.pmf_to_quantized_cdf(np.array([.1, .2, .3, .35, .05], np.float32)).tolist() == \
        [0, 6553, 19659, 39319, 62256, 65536]


def test_laplace_tables_sha1():
    from pmctf_oracle import entropy
    cdf, ln, off = entropy.GaussianTables().cdf_info()
    assert cdf.shape == (256, 103) and cdf.dtype == np.int32
    assert hashlib.sha1(cdf.tobytes()).hexdigest() == "2e0d0570db9b9fd62dfab006b1ac57759e55eb72"
    assert hashlib.sha1(ln.tobytes()).hexdigest() == "e4fe70191acf4a1d41056176c2599c5d24ac842e"
    assert hashlib.sha1(off.tobytes()).hexdigest() == "93de49f7943bafffbc0cc66c53a21ff2b31c7b6b"
    assert cdf[0, :7].tolist() == [0, 1, 2, 65533, 65534, 65535, 65536]
    assert cdf[128, :17].tolist() == [0, 7, 33, 124, 437, 1510, 5179, 17717, 47798, 60336, 64005, 65078, 65391,
                                      65482, 65508, 65515, 65536]


SCALES = [0, .5, .5, 1, 2, .01, .02, 4, 4, 8, 8, 64, 100, 1e-9, .3, .3]
INDEXES = [0, 113, 113, 133, 154, 0, 20, 174, 174, 194, 194, 255, 255, 0, 98, 98]
SYMBOLS = [0, 1, -1, 2, -3, 0, 0, 5, -7, 40, -60, 0, 1, 0, 0, -1]
STREAM = "01f6e0e56434010000eb9e2b66326159b4"


def test_build_indexes_kat_both_backends():
    from pmctf_oracle import entropy
    g = entropy.GaussianTables()
    sc = torch.tensor(SCALES)
    assert g.build_indexes_torch(sc).tolist() == INDEXES
    assert g.build_indexes_cdef(sc).tolist() == INDEXES


def test_rans_known_answer_stream_and_roundtrip():
    from pmctf_oracle import entropy
    g = entropy.GaussianTables()
    ec = entropy.EntropyCoder()
    ec.reset()
    ec.encode_with_indexes(torch.tensor(SYMBOLS), torch.tensor(INDEXES), *g.cdf_info())
    ec.flush()
    s = ec.get_encoded_stream()
    assert s.hex() == STREAM          # 17 bytes; exercises the bypass path via 40 / -60
    ec.set_stream(s)
    assert ec.decode_stream(torch.tensor(INDEXES), *g.cdf_info()).int().tolist() == SYMBOLS


def test_reference_ops_cpp_agrees_when_built():
    """oracle/_ref holds the reference's own ops.cpp (built in the build container only)."""
    import glob, importlib.util, os
    from pmctf_oracle import clib
    so = glob.glob(os.path.join(os.path.dirname(clib.__file__), "..", "_ref", "MLCodec_CXX*.so"))
    if not so:
        import pytest
        pytest.skip("oracle/_ref not built here")
    spec = importlib.util.spec_from_file_location("MLCodec_CXX", so[0])
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    rng = np.random.default_rng(0)
    for n in (3, 7, 40, 101):
        p = rng.random(n).astype(np.float32) ** 4
        p /= p.sum()
        assert list(m.pmf_to_quantized_cdf(p.tolist(), 16)) == clib.pmf_to_quantized_cdf(p).tolist()


def test_synthetic_sequences_are_pinned():
    """The fixtures under tests/golden/ were written by the real reference on these sequences: the generators must
    not drift (SHA-1 over the 8-bit planes of five 192x108 frames)."""
    import hashlib
    import pmctf_synth
    for gen, want in ((pmctf_synth.synth_yuv420, "fff11caa60613d9a249bdc4b3a9b71a0da2c3a56"),
                      (pmctf_synth.synth_yuv420_layers, "058d0de7d26eac2685d9239e858501ed7b2a91db")):
        frames = gen(192, 108, 5)
        assert hashlib.sha1(b"".join(p.tobytes() for fr in frames for p in fr)).hexdigest() == want, gen.__name__
