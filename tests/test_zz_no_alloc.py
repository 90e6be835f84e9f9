"""GPU: mf_region_frame_static (the entry point that allocates nothing) against
mf_region_frame, both called through the C-ABI on the same region table.

Collected last (test_zz_*): the round-end run uses ``pytest -x``.  Two calls in
a row, the second with a smaller table, so stale contents of the static array
would show; bar = bit-exact (same kernels, integer / copy work)."""
import ctypes

import numpy as np
import pytest

from conftest import ArrayCodebook, load_golden, synth_stack
from merlin_amd import _lib

pytestmark = pytest.mark.gpu


def _region_table(dec, image, torch):
    B = image.shape[0]
    vol = dec.decode_pixels_device(
        torch.from_numpy(image).to(dec._dev()), np.full(B, 500.0), np.zeros(B),
        0.5167, 1.0, 1.0, returnTraces='decoded')
    t = dec.region_table(vol, False)
    torch.cuda.synchronize()
    return t


def _frames(table, B, crop, shape, min_area, static, torch):
    """(count, out64[:, :count], out32[:, :count]) through one entry point."""
    L = _lib.lib()
    n = len(table)
    dev = table.rec_d.device
    o64 = torch.full((_lib.MF_FRAME_COLS, max(n, 1)), -7.0, dtype=torch.float64,
                     device=dev)
    o32 = torch.full((B, max(n, 1)), -7.0, dtype=torch.float32, device=dev)
    cnt = torch.full((1,), -1, dtype=torch.int64, device=dev)
    sp = torch.cuda.current_stream(dev).cuda_stream
    args = [table.rec_d.data_ptr(), table.inten_d.data_ptr(), n, B, 0,
            float(crop), float(shape[0]), float(shape[1]), float(min_area),
            o64.data_ptr(), o32.data_ptr(), cnt.data_ptr()]
    if static:
        rc = L.mf_region_frame_static(*args, sp)
    else:
        nb = _lib.check(L.mf_region_frame_workspace_bytes(n))
        ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
        rc = L.mf_region_frame(*args, ws.data_ptr(), nb, sp)
    _lib.check(rc)
    torch.cuda.synchronize(dev)
    m = int(cnt.item())
    return m, o64.cpu().numpy()[:, :m], o32.cpu().numpy()[:, :m]


def test_region_frame_static_matches_region_frame(gpu_device):
    import torch
    from merlin_amd.decoding import PixelBasedDecoder
    g = load_golden('g1_testsuite_fov0')
    cb = g['codebook']
    B = cb.shape[1]
    dec = PixelBasedDecoder(ArrayCodebook(cb), device=gpu_device)
    big = synth_stack(cb, (384, 320), 900, 5)
    small = synth_stack(cb, (96, 128), 60, 6)
    sizes = []
    # two calls in a row on the same GPU, the second with the smaller table
    for image, crop, min_area in ((big, 10, 2), (small, 0, 0)):
        table = _region_table(dec, image, torch)
        shape = image.shape[1:]
        m0, a64, a32 = _frames(table, B, crop, shape, min_area, False, torch)
        m1, b64, b32 = _frames(table, B, crop, shape, min_area, True, torch)
        assert m0 == m1 and m0 > 0
        assert np.array_equal(a64, b64)
        assert np.array_equal(a32, b32)
        sizes.append((len(table), m0))
    assert sizes[1][0] < sizes[0][0]
    assert sizes[0][1] < sizes[0][0]          # the filter dropped rows


def test_region_frame_static_limits(gpu_device):
    import torch
    L = _lib.lib()
    cnt = torch.full((1,), -1, dtype=torch.int64, device=gpu_device)
    sp = torch.cuda.current_stream(gpu_device).cuda_stream
    # an empty table only clears the count
    _lib.check(L.mf_region_frame_static(None, None, 0, 16, 0, 0.0, 8.0, 8.0, 0.0,
                                        None, None, cnt.data_ptr(), sp))
    torch.cuda.synchronize(gpu_device)
    assert int(cnt.item()) == 0
    # beyond the static array: a clear error, nothing launched
    rc = L.mf_region_frame_static(cnt.data_ptr(), cnt.data_ptr(),
                                  _lib.MF_FRAME_STATIC_MAX_ROWS + 1, 16, 0, 0.0,
                                  8.0, 8.0, 0.0, cnt.data_ptr(), cnt.data_ptr(),
                                  cnt.data_ptr(), sp)
    assert rc == -3
    assert b'mf_region_frame_static holds at most' in L.mf_last_error_string()
