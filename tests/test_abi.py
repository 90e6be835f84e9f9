"""CPU: the C-ABI library loads and exports every symbol include/merfish_hip.h
declares (no compute calls: there is no GPU here); host-only helpers agree
with the oracle."""
import ctypes
import os
import re

import numpy as np

import merfish_oracle as oracle
from conftest import ROOT
from merlin_amd import _lib


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    header = open(os.path.join(ROOT, 'include', 'merfish_hip.h')).read()
    declared = set(re.findall(r'\b(mf_[a-z_0-9]+)\s*\(', header))
    assert declared == set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.mf_version() >= 1


def test_gaussian_taps_match_oracle():
    for ks, s in ((1, 0.0), (3, 0.5), (5, 1.0), (7, 1.5), (9, 2.0), (5, 0.8)):
        fx, fl = _lib.gaussian_taps(ks, s)
        if ks == 1:
            assert fx.tolist() == [65536]
            continue
        assert np.array_equal(fx.astype(np.int64),
                              oracle.gaussian_kernel_fixed16(ks, s))
        assert np.array_equal(
            fl, oracle.gaussian_kernel_f64(ks, s).astype(np.float32))


def test_argument_errors_are_reported():
    L = _lib.lib()
    fx = (ctypes.c_uint32 * 4)()
    assert L.mf_gaussian_taps(4, 1.0, fx, None) == -1
    assert b'ksize' in L.mf_last_error_string()
    # null pointers are rejected before any GPU work
    rc = L.mf_decode_pixels(None, 0, 16, 1, 8, 8, 64, 64, None, 1, 0, None, None,
                            1.0, 0.5, 1.0, None, None, None, 1, None, 0, 0,
                            1, 0, None)
    assert rc == -1
    rc = L.mf_label_components(None, 1, 8, 8, 0, None, None, None)
    assert rc == -1
