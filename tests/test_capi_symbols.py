"""The C-ABI library builds, loads and exports every symbol include/gscan_hip.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from multimodal_seq2seq_gscan_amd import build, _lib
    build.build()
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gscan_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gscan_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    from multimodal_seq2seq_gscan_amd import _lib
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in gscan_hip.h but not exported"
        assert n in _lib.PROTOTYPES, f"{n} has no ctypes prototype"
    assert sorted(_lib.PROTOTYPES) == names


def test_abi_version_and_struct_sizes(lib):
    from multimodal_seq2seq_gscan_amd import _lib
    assert lib.gscan_abi_version() == _lib.ABI_VERSION
    assert ctypes.sizeof(_lib.Dims) == 18 * 4
    assert ctypes.sizeof(_lib.Params) == (32 + 8 * (_lib.MAX_ENC_LAYERS - 1)) * 8
    assert ctypes.sizeof(_lib.Batch) == 6 * 8 and ctypes.sizeof(_lib.Masks) == 8 * 8


def test_workspace_query_and_errors(lib):
    """Host-only entry points: workspace sizing and the error channel."""
    from multimodal_seq2seq_gscan_amd import _lib
    d = _lib.Dims(B=256, L=10, T=20, G=6, C=16, Co=50, K3=7, E=25, He=100, H=100, Vi=21, V=9, conditional=1,
                  auxiliary=0, bidirectional=1, pad_in=0, pad_tgt=0)
    n = lib.gscan_workspace_bytes(ctypes.byref(d))
    assert 50e6 < n < 400e6
    off, cnt = ctypes.c_size_t(), ctypes.c_size_t()
    assert lib.gscan_workspace_find(ctypes.byref(d), b"S", ctypes.byref(off), ctypes.byref(cnt)) == 0
    assert cnt.value == 256 * 20 * 400 and off.value % 256 == 0
    assert lib.gscan_workspace_find(ctypes.byref(d), b"nope", ctypes.byref(off), ctypes.byref(cnt)) != 0
    assert b"nope" in lib.gscan_last_error()
    # shapes the register/LDS-resident kernels have no variant for are sized too (they run on the streaming kernels) ...
    for odd in (dict(H=77), dict(H=256, He=256), dict(G=12), dict(L=128)):
        kw = dict(B=4, L=10, T=20, G=6, C=16, Co=50, K3=7, E=25, He=100, H=100, Vi=21, V=9, conditional=1,
                  auxiliary=0, bidirectional=1, pad_in=0, pad_tgt=0)
        kw.update(odd)
        dd = _lib.Dims(**kw)
        assert lib.gscan_workspace_bytes(ctypes.byref(dd)) > 0, odd
    # ... and what no kernel takes fails with the reason
    bad = _lib.Dims(B=4, L=10, T=20, G=6, C=16, Co=50, K3=7, E=25, He=100, H=1100, Vi=21, V=9, conditional=1,
                    auxiliary=0, bidirectional=1, pad_in=0, pad_tgt=0)
    assert lib.gscan_workspace_bytes(ctypes.byref(bad)) == 0
    assert b"decoder_hidden_size 1100" in lib.gscan_last_error()
    even = _lib.Dims(B=4, L=10, T=20, G=6, C=16, Co=50, K3=4, E=25, He=100, H=100, Vi=21, V=9, conditional=1,
                     auxiliary=0, bidirectional=1, pad_in=0, pad_tgt=0)
    assert lib.gscan_workspace_bytes(ctypes.byref(even)) == 0
    assert b"cnn_kernel_size" in lib.gscan_last_error()


def test_decoder_kernel_family_reports_the_resident_path_conditions(lib):
    """gscan_decoder_kernel_family (ABI 14): 1 = the register/LDS-resident decoder kernels, 0 = the streaming ones; the
    conditions are the ones the header lists above gscan_dims (a target vocabulary above 16 used to fall to the
    streaming kernels without a word)."""
    from multimodal_seq2seq_gscan_amd import _lib
    base = dict(B=256, L=10, T=20, G=6, C=16, Co=50, K3=7, E=25, He=100, H=100, Vi=21, V=9, conditional=1,
                auxiliary=0, bidirectional=1, pad_in=0, pad_tgt=0)

    def family(**kw):
        return lib.gscan_decoder_kernel_family(ctypes.byref(_lib.Dims(**dict(base, **kw))))
    assert family() == 1                                   # the benchmark shape
    assert family(V=16) == 1 and family(V=17) == 0         # one matrix-core tile of logits
    assert family(H=96) == 1 and family(H=98) == 0 and family(H=128) == 0
    assert family(G=8) == 1 and family(G=9) == 0           # at most 64 memories per attention (81 cells)
    assert family(L=64, G=6) in (0, 1) and family(L=65) == 0
    assert family(L=44, G=6) == 1 and family(L=64, G=8) == 0    # the row's memories must fit 160 KB of LDS
    assert family(H=0) < 0 and b"" != lib.gscan_last_error()
    assert lib.gscan_decoder_kernel_family(None) < 0
    # the early-gradient wait needs a backward pass to have been issued in this process
    assert lib.gscan_comm_set_early_allreduce(None, None, 0) == 0
