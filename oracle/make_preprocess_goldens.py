"""Golden vectors for the preprocess stage from the REFERENCE's own
``merlin/util/deconvolve.py`` and ``merlin/util/imagefilters.py``.

Run in the build container only (the reference does not travel):

    /opt/conda/bin/python3.9 oracle/make_preprocess_goldens.py

Both modules are imported unmodified from /root/reference.  cv2 is absent from
every interpreter here, so a stub module supplies the calls they make
(``GaussianBlur`` with ``dst`` / ``borderType``, ``subtract``, ``addWeighted``,
``multiply``, ``divide``) as the restatements of ``oracle/preprocess_oracle.py``.
Consequence: the goldens pin the Lucy-Richardson loop, the acceleration
parameter (numpy's own ``np.sum``), the clipping and the high-pass wrap/zero
logic against real reference code; the cv2 primitives themselves stay
"parity unpinned".

Outputs: ``tests/golden/pp_*.npz`` (inputs + expected outputs; data only).
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, HERE)
import preprocess_oracle as po  # noqa: E402


def install_shims():
    pkg = types.ModuleType('merlin')
    pkg.__path__ = [os.path.join(REF, 'merlin')]
    sys.modules['merlin'] = pkg
    util = types.ModuleType('merlin.util')
    util.__path__ = [os.path.join(REF, 'merlin', 'util')]
    sys.modules['merlin.util'] = util

    cv2 = types.ModuleType('cv2')
    cv2.BORDER_REPLICATE = 1

    def GaussianBlur(src, ksize, sigmaX, dst=None, borderType=None):
        assert ksize[0] == ksize[1] and borderType == cv2.BORDER_REPLICATE
        if src.dtype == np.uint16:
            out = po.gaussian_blur_u16_replicate(np.ascontiguousarray(src),
                                                 ksize[0], sigmaX)
        else:
            assert src.dtype == np.float64
            out = po.gaussian_blur_f64_replicate(np.ascontiguousarray(src),
                                                 ksize[0], sigmaX)
        if dst is not None:
            dst[...] = out
            return dst
        return out

    def subtract(a, b, dst):
        dst[...] = a - b
        return dst

    def addWeighted(a, alpha, b, beta, gamma, dst):
        dst[...] = a * float(alpha) + b * float(beta) + float(gamma)
        return dst

    def multiply(a, b, dst):
        dst[...] = a * b
        return dst

    def divide(a, b, dst):
        with np.errstate(divide='ignore', invalid='ignore'):
            q = a / b
        q[b == 0] = 0          # cv2.divide: x / 0 = 0
        dst[...] = q
        return dst

    def filter2D(src, ddepth, kernel, borderType=None):
        assert ddepth == -1 and borderType == cv2.BORDER_REPLICATE
        return po.filter2d_f64_replicate(np.ascontiguousarray(src, dtype=np.float64),
                                         np.asarray(kernel, dtype=np.float64))

    cv2.filter2D = filter2D
    cv2.GaussianBlur = GaussianBlur
    cv2.subtract = subtract
    cv2.addWeighted = addWeighted
    cv2.multiply = multiply
    cv2.divide = divide
    sys.modules['cv2'] = cv2


def camera_image(shape, n_spots, seed, offset=2000, bg=100, amp=(500, 6000),
                 psf=1.4, saturate=False):
    """A raw camera frame: offset + Poisson background + Gaussian spots."""
    rng = np.random.default_rng(seed)
    h, w = shape
    img = offset + rng.poisson(bg, size=shape).astype(np.float64)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(n_spots):
        cy, cx = rng.uniform(0, h), rng.uniform(0, w)
        a = rng.uniform(*amp)
        img += a * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * psf ** 2))
    if saturate:
        img[5:9, 5:9] = 70000
        img[h - 3:, w - 3:] = 70000
    return np.clip(np.rint(img), 0, 65535).astype(np.uint16)


def run_case(name, deconvolve, imagefilters, image, hp_sigma=3, decon_sigma=2,
             decon_size=None, iterations=20):
    hp_size = int(2 * np.ceil(2 * hp_sigma) + 1)          # preprocess.py:104
    if decon_size is None:
        decon_size = int(2 * np.ceil(2 * decon_sigma) + 1)   # preprocess.py:50-51
    hp = imagefilters.high_pass_filter(image, hp_size, hp_sigma)
    assert hp.dtype == np.uint16
    filtered = hp.astype(float)                           # preprocess.py:108
    dec = deconvolve.deconvolve_lucyrichardson(filtered, decon_size,
                                               decon_sigma, iterations)
    out16 = dec.astype(np.uint16)                         # preprocess.py:139
    hist = np.histogram(out16, bins=np.arange(0, np.iinfo(np.uint16).max, 1))[0]
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(
        path, image=image, highpass=hp, deconvolved=dec, output=out16,
        histogram=hist.astype(np.int64),
        params=np.array([hp_sigma, decon_sigma, decon_size, iterations],
                        dtype=np.float64))
    print('{:24s} max={:9.1f} nonzero={:6d} size={:.0f} KB'.format(
        name, dec.max(), int((out16 > 0).sum()), os.path.getsize(path) / 1024))


def even_cases(deconvolve, imagefilters):
    """Even window sizes: the reference's scipy.ndimage.convolve branch
    (deconvolve.py:102-104, 118-119, 126-128); real scipy, no stub involved in
    the blur.  ``python oracle/make_preprocess_goldens.py even`` writes only
    these."""
    for name, img, size, sigma, its in (
            ('ppeven_w8', camera_image((72, 100), 30, seed=11), 8, 2, 8),
            ('ppeven_w4', camera_image((50, 70), 20, seed=12), 4, 1.0, 5),
            ('ppeven_w12_sat', camera_image((64, 64), 25, seed=13, saturate=True), 12, 0.8, 4)):
        hp = imagefilters.high_pass_filter(img, 13, 3)
        dec = deconvolve.deconvolve_lucyrichardson(hp.astype(float), size, sigma, its)
        # the kernel as this interpreter's numpy makes it (np.exp differs in the
        # last bit between numpy builds): stored so that the comparison does not
        # depend on the numpy of whoever runs the tests
        from merlin.util import matlab
        kernel = matlab.matlab_gauss2D(shape=(size, size), sigma=sigma)
        np.savez_compressed(os.path.join(OUT, name + '.npz'), image=img, highpass=hp,
                            kernel=kernel, deconvolved=dec, output=dec.astype(np.uint16),
                            params=np.array([3, sigma, size, its], dtype=np.float64))
        print('{:24s} max={:9.1f}'.format(name, dec.max()))


def main():
    install_shims()
    from merlin.util import deconvolve, imagefilters
    os.makedirs(OUT, exist_ok=True)
    if sys.argv[1:] == ['even']:
        even_cases(deconvolve, imagefilters)
        return

    # one frame of the reference's own test data (raw camera frame)
    import tifffile
    frame = tifffile.imread(os.path.join(
        REF, 'test', 'auxiliary_files', 'test_0_0.tif'))[1]
    run_case('pp_testsuite_frame', deconvolve, imagefilters,
             np.ascontiguousarray(frame.astype(np.uint16)))
    # synthetic frames: defaults, non-default sigmas / iteration counts, edges
    run_case('pp_synth96', deconvolve, imagefilters,
             camera_image((96, 80), 40, seed=1))
    run_case('pp_synth_sat', deconvolve, imagefilters,
             camera_image((72, 100), 30, seed=2, saturate=True))
    run_case('pp_synth_params', deconvolve, imagefilters,
             camera_image((64, 64), 25, seed=3), hp_sigma=2, decon_sigma=1.5,
             iterations=7)
    run_case('pp_synth_iter2', deconvolve, imagefilters,
             camera_image((50, 70), 20, seed=4), hp_sigma=3, decon_sigma=1,
             decon_size=9, iterations=2)
    z = np.zeros((40, 40), dtype=np.uint16)
    z[20, 20] = 1000
    run_case('pp_point', deconvolve, imagefilters, z, iterations=5)

    # the Guo et al. variant (DeconvolutionPreprocessGuo: 2 iterations by default)
    for name, img, size, sigma, its in (
            ('ppguo_synth', camera_image((72, 100), 30, seed=5), 9, 2, 2),
            ('ppguo_sat', camera_image((64, 64), 25, seed=6, saturate=True), 9, 2, 3),
            ('ppguo_params', camera_image((50, 70), 20, seed=7), 7, 1.5, 4)):
        hp = imagefilters.high_pass_filter(img, 13, 3)
        pf, pb = deconvolve.calculate_projectors(size, sigma)
        dec = deconvolve.deconvolve_lucyrichardson_guo(hp.astype(float), size, sigma, its)
        np.savez_compressed(os.path.join(OUT, name + '.npz'), image=img, highpass=hp, pf=pf,
                            pb=pb, deconvolved=dec, output=dec.astype(np.uint16),
                            params=np.array([3, sigma, size, its], dtype=np.float64))
        print('{:24s} max={:9.1f}'.format(name, dec.max()))
    even_cases(deconvolve, imagefilters)


if __name__ == '__main__':
    main()
