"""CPU: the oracle (oracle/merfish_oracle.py) against the golden vectors that
were produced by the reference's own decoding.py (oracle/make_goldens.py).

Tolerances: decoded indexes and every record/refactor value computed from the
golden images are bit-exact.  Magnitudes/traces/distances agree to float32
rounding only: the reference takes np.linalg.norm per pixel, whose float32
summation order belongs to the BLAS numpy was built against (OpenBLAS sdot
here); the oracle fixes a sequential order."""
import warnings

import numpy as np
import pytest

import merfish_oracle as oracle
from conftest import golden_names, load_golden, frame_to_array


def _decode(g):
    p = g['p']
    dm = oracle.normalized_codebook(g['codebook'])
    img = g['image']
    if not p['decode3d']:
        return oracle.decode_pixels(img, dm, g['sf'], g['bg'], p['dthr'],
                                    p['mthr'], p['sigma'])
    outs = [oracle.decode_pixels(img[:, z], dm, g['sf'], g['bg'], p['dthr'],
                                 p['mthr'], p['sigma'])
            for z in range(img.shape[1])]
    return (np.array([o[0] for o in outs]), np.array([o[1] for o in outs]),
            np.array([o[2] for o in outs]),
            np.array([o[3] for o in outs]).astype(np.float32))


@pytest.mark.parametrize('name', golden_names())
def test_decode_pixels_matches_reference(name):
    g = load_golden(name)
    di, pm, npt, d = _decode(g)
    assert di.dtype == np.int16 and pm.dtype == np.float32
    assert np.array_equal(di, g['decoded'])
    np.testing.assert_allclose(pm, g['magnitudes'], rtol=1e-6, atol=0)
    np.testing.assert_allclose(npt, g['traces'], rtol=0, atol=5e-7)
    np.testing.assert_allclose(d, g['distances'], rtol=1e-6, atol=2e-7)


@pytest.mark.parametrize('name', golden_names())
def test_records_match_reference(name):
    g = load_golden(name)
    p = g['p']
    N, B = g['codebook'].shape
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        rec = oracle.extract_all_barcodes(
            N, g['decoded'], g['magnitudes'], g['traces'], g['distances'],
            p['fov'], p['crop'], p['z_index'], g.get('tform'), p['min_area'])
    r = frame_to_array(rec, B)
    assert r.shape == g['records'].shape
    if p['decode3d']:
        # regionprops' float32 means / moment-based centroids: rounding only
        np.testing.assert_allclose(r, g['records'], rtol=1e-6, atol=1e-6)
    else:
        assert np.array_equal(r, g['records'])


@pytest.mark.parametrize('name', [n for n in golden_names()
                                  if 'decode3d' not in n])
def test_refactors_match_reference(name):
    g = load_golden(name)
    p = g['p']
    dm = oracle.normalized_codebook(g['codebook'])
    rf, bgr, seen = oracle.extract_refactors(
        g['decoded'], g['magnitudes'], g['traces'], dm, p['refactor_area'],
        False)
    assert np.array_equal(rf, g['refactors'], equal_nan=True)
    assert np.array_equal(seen, g['barcodes_seen'])
    if 'refactors_bg' in g:
        rf, bgr, seen = oracle.extract_refactors(
            g['decoded'], g['magnitudes'], g['traces'], dm,
            p['refactor_area'], True)
        assert np.array_equal(rf, g['refactors_bg'], equal_nan=True)
        assert np.array_equal(bgr, g['background_refactors'], equal_nan=True)


def test_blur_known_answers():
    """Hand-derived known answers for the fixed-point cv2 blur restatement
    (cv2 is absent here: this stage is otherwise parity-unpinned)."""
    k = oracle.gaussian_kernel_fixed16(5, 1.0)
    assert k.tolist() == [3571, 16004, 26386, 16004, 3571]
    assert k.sum() == 65536
    for ks, s in ((3, 0.5), (7, 1.5), (9, 2.0)):
        kk = oracle.gaussian_kernel_fixed16(ks, s)
        assert kk.sum() == 65536 and np.array_equal(kk, kk[::-1])
    # constant images are preserved exactly (taps sum to 1), any value
    for v in (0, 1, 100, 65535):
        img = np.full((9, 11), v, dtype=np.uint16)
        assert np.array_equal(oracle.gaussian_blur_u16(img, 5, 1.0), img)
    # impulse: separable outer product of the taps, rounded half up
    img = np.zeros((9, 9), dtype=np.uint16)
    img[4, 4] = 1000
    exp = (np.outer(k, k) * 1000 + (1 << 31)) >> 32
    out = oracle.gaussian_blur_u16(img, 5, 1.0)
    assert np.array_equal(out[2:7, 2:7], exp.astype(np.uint16))
    # BORDER_REFLECT_101 at a corner: pixel (0,0) sees rows/cols 2,1,0,1,2
    img = np.arange(36, dtype=np.uint16).reshape(6, 6) * 100
    idx = [2, 1, 0, 1, 2]
    acc = sum(int(k[i]) * int(k[j]) * int(img[idx[i], idx[j]])
              for i in range(5) for j in range(5))
    assert oracle.gaussian_blur_u16(img, 5, 1.0)[0, 0] == (acc + (1 << 31)) >> 32


def test_optimize_reductions():
    rng = np.random.default_rng(0)
    rf = rng.uniform(0.5, 1.5, (6, 16))
    rf[2, 3] = 0
    rf[4, 5] = np.nan
    prev = np.tile(rng.uniform(300, 600, 16), (6, 1))
    sf = oracle.reduce_scale_factors(rf, prev)
    r2 = rf.copy()
    r2[2, 3] = 1
    assert np.array_equal(sf, np.nanmedian(r2 * prev, axis=0))
    hist = np.zeros((2, 65534))
    hist[0, 100:200] = 1
    hist[1, 10] = 5
    hist[1, 4000] = 1
    assert oracle.initial_scale_factors(hist).tolist() == [191.0, 12.0]
