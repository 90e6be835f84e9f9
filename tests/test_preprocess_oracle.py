"""Preprocess oracle (oracle/preprocess_oracle.py) against the golden vectors the
reference's own deconvolve.py / imagefilters.py produced
(oracle/make_preprocess_goldens.py).  CPU only."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
import preprocess_oracle as po

PP_NAMES = sorted(os.path.basename(p)[:-4]
                  for p in glob.glob(os.path.join(GOLDEN_DIR, 'pp_*.npz')))


def load_pp(name):
    g = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    d = {k: g[k] for k in g.files}
    p = d['params']
    d['p'] = dict(hp_sigma=float(p[0]), decon_sigma=float(p[1]),
                  decon_size=int(p[2]), iterations=int(p[3]))
    return d


def test_goldens_present():
    assert len(PP_NAMES) >= 6


@pytest.mark.parametrize('name', PP_NAMES)
def test_high_pass_bit_exact(name):
    g = load_pp(name)
    hp = po.high_pass_filter(g['image'],
                             po.filter_size_for_sigma(g['p']['hp_sigma']),
                             g['p']['hp_sigma'])
    assert hp.dtype == np.uint16
    assert np.array_equal(hp, g['highpass'])


@pytest.mark.parametrize('name', PP_NAMES)
def test_lucy_richardson_bit_exact(name):
    # same primitives + the reference's own loop: identical float64 bits
    g = load_pp(name)
    p = g['p']
    dec = po.deconvolve_lucyrichardson(g['highpass'].astype(np.float64),
                                       p['decon_size'], p['decon_sigma'],
                                       p['iterations'])
    assert np.array_equal(dec, g['deconvolved'])
    assert np.array_equal(po.to_uint16(dec), g['output'])


@pytest.mark.parametrize('name', PP_NAMES)
def test_preprocess_image_and_histogram(name):
    g = load_pp(name)
    p = g['p']
    out = po.preprocess_image(g['image'], p['hp_sigma'], p['decon_sigma'],
                              p['decon_size'], p['iterations'])
    assert out.dtype == np.uint16 and np.array_equal(out, g['output'])
    assert np.array_equal(po.pixel_histogram(out), g['histogram'])


def test_to_uint16_wraps_like_numpy_on_x86():
    v = np.array([0.0, 0.999, 1.0, 65535.9, 65536.0, 70000.5, 194617.5, -0.5,
                  -1.0, 2147483647.9, 2147483648.0, 1e300, np.nan, np.inf])
    want = np.array([0, 0, 1, 65535, 0, 70000 - 65536, 194617 - 2 * 65536, 0,
                     65535, 65535, 0, 0, 0, 0], dtype=np.uint16)
    assert np.array_equal(po.to_uint16(v), want)


def test_blur_known_answers():
    # constant image: taps sum to exactly 65536 / 1.0, borders replicate
    c = np.full((20, 24), 1234, dtype=np.uint16)
    assert np.array_equal(po.gaussian_blur_u16_replicate(c, 13, 3.0), c)
    f = po.gaussian_blur_f64_replicate(np.full((12, 12), 2.5), 9, 2.0)
    assert np.allclose(f, 2.5, rtol=1e-15)
    # impulse: the blur of a delta is the outer product of the taps
    from merfish_oracle import gaussian_kernel_f64
    d = np.zeros((21, 21))
    d[10, 10] = 1.0
    k = gaussian_kernel_f64(9, 2.0)
    assert np.allclose(po.gaussian_blur_f64_replicate(d, 9, 2.0)[6:15, 6:15],
                       np.outer(k, k), rtol=1e-14, atol=0)
    # a step at the border stays a step (replicate, not reflect)
    s = np.zeros((16, 16), dtype=np.uint16)
    s[:, 0] = 60000
    b = po.gaussian_blur_u16_replicate(s, 13, 3.0)
    assert b[8, 0] > b[8, 1] > b[8, 6] > 0 and b[8, 7] == 0


GUO_NAMES = sorted(os.path.basename(p)[:-4]
                   for p in glob.glob(os.path.join(GOLDEN_DIR, 'ppguo_*.npz')))


@pytest.mark.parametrize('name', GUO_NAMES)
def test_guo_variant_bit_exact(name):
    g = load_pp(name)
    p = g['p']
    pf, pb = po.calculate_projectors(p['decon_size'], p['decon_sigma'])
    np.testing.assert_allclose(pf, g['pf'], rtol=1e-14, atol=0)
    np.testing.assert_allclose(pb, g['pb'], rtol=1e-10, atol=1e-16)      # FFT builds differ
    dec = po.deconvolve_lucyrichardson_guo(g['highpass'].astype(np.float64), p['decon_size'],
                                           p['decon_sigma'], p['iterations'])
    np.testing.assert_allclose(dec, g['deconvolved'], rtol=1e-9, atol=1e-12)
    assert np.array_equal(po.to_uint16(g['deconvolved']), g['output'])


def test_product_projectors_match_golden():
    from merlin_amd.preprocess import calculate_projectors, matlab_gauss2D
    assert len(GUO_NAMES) == 3
    for name in GUO_NAMES:
        g = load_pp(name)
        pf, pb = calculate_projectors(g['p']['decon_size'], g['p']['decon_sigma'])
        np.testing.assert_allclose(pf, g['pf'], rtol=1e-14, atol=0)
        np.testing.assert_allclose(pb, g['pb'], rtol=1e-10, atol=1e-16)
    assert abs(matlab_gauss2D((9, 9), 2.0).sum() - 1.0) < 1e-15


EVEN_NAMES = sorted(os.path.basename(p)[:-4]
                    for p in glob.glob(os.path.join(GOLDEN_DIR, 'ppeven_*.npz')))


def test_ndimage_convolve_restatement_matches_scipy():
    """The even-window branch's blur (deconvolve.py:119, 127) against scipy
    itself, bit for bit: symmetric and asymmetric kernels, even / odd / mixed
    sizes, images smaller than the kernel, kernels with dropped taps."""
    from scipy import ndimage
    rng = np.random.default_rng(5)
    cases = [((37, 53), po.matlab_gauss2D((8, 8), 2.0)),
             ((20, 31), po.matlab_gauss2D((4, 4), 1.0)),
             ((64, 64), po.matlab_gauss2D((10, 10), 2.5)),
             ((30, 30), po.matlab_gauss2D((5, 5), 1.2)),
             ((16, 40), po.matlab_gauss2D((2, 2), 0.5)),
             ((24, 24), po.matlab_gauss2D((12, 12), 0.8)),      # 12 taps below DBL_EPSILON
             ((5, 7), po.matlab_gauss2D((8, 8), 2.0)),          # image smaller than the kernel
             ((12, 15), rng.random((4, 6))),                    # not symmetric: flip and origin
             ((12, 15), rng.random((3, 6)))]
    for shape, kernel in cases:
        img = rng.normal(100, 50, size=shape)
        assert np.array_equal(po.ndimage_convolve_constant(img, kernel),
                              ndimage.convolve(img, kernel, mode='constant'))


def test_even_goldens_present():
    assert len(EVEN_NAMES) == 3


@pytest.mark.parametrize('name', EVEN_NAMES)
def test_even_window_lucy_richardson(name):
    """Even windowSize = the scipy.ndimage.convolve branch (deconvolve.py:102-104,
    118-119, 126-128), goldens made by the reference with real scipy.  With the
    kernel the reference's interpreter computed the oracle is bit-exact; with
    this interpreter's np.exp the kernel may differ in the last bit."""
    g = load_pp(name)
    p = g['p']
    hp = g['highpass'].astype(np.float64)
    dec = po.deconvolve_lucyrichardson(hp, p['decon_size'], p['decon_sigma'], p['iterations'],
                                       gaussianFilter=g['kernel'])
    assert np.array_equal(dec, g['deconvolved'])
    np.testing.assert_allclose(po.matlab_gauss2D((p['decon_size'],) * 2, p['decon_sigma']),
                               g['kernel'], rtol=1e-14, atol=1e-30)
    own = po.deconvolve_lucyrichardson(hp, p['decon_size'], p['decon_sigma'], p['iterations'])
    np.testing.assert_allclose(own, g['deconvolved'], rtol=1e-9, atol=1e-9)
    assert np.array_equal(po.to_uint16(g['deconvolved']), g['output'])
