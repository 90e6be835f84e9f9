"""GPU: mf_label_components (tile-local labelling in LDS + links across tile
edges) against scipy's labelling of every barcode, on images built to stress
it: few codes and a high fill (components spanning many 64 x 32 tiles),
diagonal-only contacts, thin long structures, sizes that are no multiple of
the tile, both connectivities."""
import numpy as np
import pytest
from scipy import ndimage

pytestmark = pytest.mark.gpu


def _label_gpu(dec, conn3d, dev):
    import torch
    from merlin_amd import _lib
    L = _lib.lib()
    Z, H, W = dec.shape
    d = torch.from_numpy(np.ascontiguousarray(dec)).to(dev)
    labels = torch.empty((Z, H, W), dtype=torch.int32, device=dev)
    counts = torch.zeros(_lib.MF_LABEL_COUNTS_LEN, dtype=torch.int64, device=dev)
    _lib.check(L.mf_label_components(
        d.data_ptr(), Z, H, W, int(conn3d), labels.data_ptr(),
        counts.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
    torch.cuda.synchronize(dev)
    return labels.cpu().numpy(), counts.cpu().numpy()[:2]


def _label_ref(dec, conn3d):
    """label = linear index of the component's first pixel (raster order),
    components per barcode with full connectivity (skimage's default,
    decoding.py:165-168)."""
    Z, H, W = dec.shape
    out = np.full(dec.shape, -1, dtype=np.int64)
    lin = np.arange(dec.size).reshape(dec.shape)
    n_comp = 0
    for c in np.unique(dec[dec >= 0]):
        m = dec == c
        if conn3d:
            lab, n = ndimage.label(m, structure=np.ones((3, 3, 3)))
            first = ndimage.minimum(lin, lab, np.arange(1, n + 1))
            out[m] = np.asarray(first, dtype=np.int64)[lab[m] - 1]
            n_comp += n
        else:
            for z in range(Z):
                lab, n = ndimage.label(m[z], structure=np.ones((3, 3)))
                if n == 0:
                    continue
                first = ndimage.minimum(lin[z], lab, np.arange(1, n + 1))
                out[z][m[z]] = np.asarray(first, dtype=np.int64)[lab[m[z]] - 1]
                n_comp += n
    return out, n_comp


def _images():
    rng = np.random.default_rng(5)
    # 1. two codes at the percolation threshold: components span many tiles
    a = rng.integers(0, 2, size=(2, 150, 333)).astype(np.int16)
    a[rng.random(a.shape) < 0.35] = -1
    yield 'percolation', a
    # 2. smoothed noise -> blobs of three codes, odd size
    f = ndimage.gaussian_filter(rng.standard_normal((3, 97, 201)), (0, 4, 4))
    b = np.digitize(f, [-0.08, 0.0, 0.08]).astype(np.int16) - 1
    yield 'blobs', b
    # 3. diagonals only (8-connectivity) across tile corners, and a long
    # one-pixel snake through many tiles
    c = np.full((2, 130, 260), -1, dtype=np.int16)
    i = np.arange(130)
    c[0, i, i] = 3
    c[0, i, 259 - i] = 3
    c[1, 64, :] = 7
    c[1, 10:120, 63] = 7
    c[1, 10:120, 64] = 9
    c[1, 31, 5:200] = 9
    c[1, 32, 5:200] = 7
    yield 'lines', c
    # 4. everything one component; nothing assigned; a single pixel
    yield 'full', np.zeros((1, 70, 129), dtype=np.int16)
    yield 'empty', np.full((2, 33, 65), -1, dtype=np.int16)
    e = np.full((1, 40, 40), -1, dtype=np.int16)
    e[0, 39, 39] = 5
    yield 'single', e
    # 5. checkerboard of two codes: every same-code contact is diagonal
    yy, xx = np.mgrid[0:96, 0:192]
    yield 'checker', ((yy + xx) % 2).astype(np.int16)[None].repeat(3, 0)


@pytest.mark.parametrize('conn3d', [False, True])
def test_label_components_against_scipy(gpu_device, conn3d):
    for name, dec in _images():
        got, counts = _label_gpu(dec, conn3d, gpu_device)
        exp, n_comp = _label_ref(dec, conn3d)
        assert np.array_equal(got, exp), (name, conn3d)
        assert counts[0] == (dec >= 0).sum(), name
        assert counts[1] == n_comp, name


def test_misaligned_labels_are_rejected(gpu_device):
    """labels are read and written 16 bytes at a time: a pointer that is not
    16-byte aligned is an argument error, not a fault."""
    import torch
    from merlin_amd import _lib
    L = _lib.lib()
    d = torch.zeros((1, 32, 64), dtype=torch.int16, device=gpu_device)
    buf = torch.empty(32 * 64 + 4, dtype=torch.int32, device=gpu_device)
    counts = torch.zeros(_lib.MF_LABEL_COUNTS_LEN, dtype=torch.int64,
                         device=gpu_device)
    rc = L.mf_label_components(d.data_ptr(), 1, 32, 64, 0,
                               buf.data_ptr() + 4, counts.data_ptr(),
                               torch.cuda.current_stream(gpu_device).cuda_stream)
    assert rc == -1                                   # MF_EINVAL
    assert b'aligned' in L.mf_last_error_string()
