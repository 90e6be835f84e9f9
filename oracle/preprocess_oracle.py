"""CPU oracle for the preprocess stage.  TEST INFRASTRUCTURE ONLY.

numpy restatement of what produces the decoder's uint16 stack in the reference
(emanuega/MERlin v0.1.6).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of the dev/bench scripts may import it; nothing under
``merlin_amd/`` does.

Reference files restated (file:line of /root/reference):

* ``merlin/util/imagefilters.py:9-28``     high_pass_filter
* ``merlin/util/deconvolve.py:70-133``     deconvolve_lucyrichardson (odd window)
* ``merlin/analysis/preprocess.py:103-108, 133-140``  _high_pass_filter,
  _preprocess_image (``.astype(np.uint16)`` of the float64 result)
* ``merlin/analysis/preprocess.py:114-129``  per-bit pixel histogram

Third-party arithmetic not under /root/reference, restated from OpenCV 4.x's
published sources (opencv-python>=3 is unpinned in requirements.txt:6; cv2 is
absent from this container, so these stages are **parity unpinned**):

* ``cv2.GaussianBlur`` on CV_16U, BORDER_REPLICATE: the bit-exact fixed-point
  path (16.16 taps with error diffusion, exact u32 / u64 sums,
  ``(acc + 2**31) >> 32``), as in ``merfish_oracle.gaussian_blur_u16`` but with
  the replicated border.
* ``cv2.GaussianBlur`` on CV_64F, BORDER_REPLICATE: ``sepFilter2D`` with the
  float64 kernel of getGaussianKernel.  Row pass = generic ``RowFilter``:
  ``s = k[0]*S[0]; s += k[j]*S[j]`` left to right.  Column pass =
  ``SymmColumnFilter``: ``s = k[c]*S[c]; s += k[c+j]*(S[c+j] + S[c-j])``.
  Every product and sum is rounded on its own (no fused multiply-add; an
  AVX2/FMA build of OpenCV may differ in the last ulp).
* ``cv2.subtract / addWeighted / multiply / divide`` on CV_64F: the plain
  float64 operation per element (``addWeighted`` = ``a*alpha + b*beta + gamma``).

Everything AROUND those calls (the Lucy-Richardson loop, its acceleration
parameter, the clipping and the uint16 hand-over) is pinned by golden vectors
made by the reference's own ``deconvolve.py`` / ``imagefilters.py``
(``oracle/make_preprocess_goldens.py``, ``tests/golden/pp_*.npz``).
"""
import numpy as np

from merfish_oracle import gaussian_kernel_f64, gaussian_kernel_fixed16


def filter_size_for_sigma(sigma: float) -> int:
    """preprocess.py:50-51, 104"""
    return int(2 * np.ceil(2 * sigma) + 1)


def gaussian_blur_u16_replicate(img: np.ndarray, ksize: int,
                                sigma: float) -> np.ndarray:
    """Exact integer separable blur of one uint16 plane, BORDER_REPLICATE."""
    assert img.dtype == np.uint16 and img.ndim == 2
    if ksize == 1:
        return img.copy()
    k = gaussian_kernel_fixed16(ksize, sigma).astype(np.uint64)
    r = ksize // 2
    h, w = img.shape
    p = np.pad(img.astype(np.uint64), r, mode='edge')
    hs = np.zeros((h + 2 * r, w), dtype=np.uint64)
    for j in range(ksize):
        hs += k[j] * p[:, j:j + w]
    vs = np.zeros((h, w), dtype=np.uint64)
    for j in range(ksize):
        vs += k[j] * hs[j:j + h, :]
    out = (vs + np.uint64(1 << 31)) >> np.uint64(32)
    return np.minimum(out, 65535).astype(np.uint16)


def gaussian_blur_f64_replicate(img: np.ndarray, ksize: int,
                                sigma: float) -> np.ndarray:
    """cv2.GaussianBlur on float64, BORDER_REPLICATE (see the module header for
    the summation order)."""
    assert img.dtype == np.float64 and img.ndim == 2
    if ksize == 1:
        return img.copy()
    k = gaussian_kernel_f64(ksize, sigma)
    r = ksize // 2
    h, w = img.shape
    p = np.pad(img, r, mode='edge')
    hs = k[0] * p[:, 0:w]
    for j in range(1, ksize):
        hs = hs + k[j] * p[:, j:j + w]
    vs = k[r] * hs[r:r + h, :]
    for j in range(1, r + 1):
        vs = vs + k[r + j] * (hs[r + j:r + j + h, :] + hs[r - j:r - j + h, :])
    return vs


def high_pass_filter(image: np.ndarray, windowSize: int,
                     sigma: float) -> np.ndarray:
    """imagefilters.py:9-28 for uint16 input: the wrapped uint16 difference,
    zeroed where the low-pass exceeds the image."""
    assert image.dtype == np.uint16
    lowpass = gaussian_blur_u16_replicate(image, windowSize, sigma)
    hp = image - lowpass                      # uint16 arithmetic wraps
    hp[lowpass > image] = 0
    return hp


def deconvolve_lucyrichardson(image: np.ndarray, windowSize: int,
                              sigmaG: float, iterationCount: int,
                              trace: list = None,
                              gaussianFilter: np.ndarray = None) -> np.ndarray:
    """deconvolve.py:70-133: odd windowSize = the cv2.GaussianBlur branch, even
    windowSize = the scipy.ndimage.convolve branch (:102-104, 118-119, 126-128).
    ``trace`` (optional list) receives the acceleration parameter of every
    iteration.  ``gaussianFilter`` replaces the even branch's
    matlab_gauss2D kernel (np.exp differs in the last bit between numpy builds;
    the goldens carry the kernel of the interpreter that ran the reference)."""
    even = windowSize % 2 != 1
    if even and gaussianFilter is None:
        gaussianFilter = matlab_gauss2D((windowSize, windowSize), sigmaG)
    image = np.asarray(image, dtype=np.float64)
    eps = np.finfo(float).eps
    Y = np.copy(image)
    J1 = np.copy(image)
    J2 = np.copy(image)
    wI = np.copy(image)
    T1 = np.zeros(image.shape, dtype=float)
    T2 = np.zeros(image.shape, dtype=float)
    lam = 0
    for i in range(iterationCount):
        if i > 1:
            lam = np.sum(T1 * T2) / (np.sum(T2 * T2) + eps)
            lam = max(min(lam, 1), 0)
        if trace is not None:
            trace.append(float(lam))
        Y = J1 - J2                           # cv2.subtract
        Y = J1 * 1.0 + Y * float(lam) + 0.0   # cv2.addWeighted
        np.clip(Y, 0, None, Y)
        if even:                              # deconvolve.py:118-119
            reblurred = ndimage_convolve_constant(Y, gaussianFilter)
        else:
            reblurred = gaussian_blur_f64_replicate(Y, windowSize, sigmaG)
        np.clip(reblurred, eps, None, reblurred)
        imR = wI / reblurred                  # cv2.divide (divisor never 0)
        imR += eps
        if even:                              # deconvolve.py:126-128
            imR = ndimage_convolve_constant(imR, gaussianFilter)
            imR[imR > 2 ** 16] = 0
        else:
            imR = gaussian_blur_f64_replicate(imR, windowSize, sigmaG)
        J2 = J1
        J1 = Y * imR
        T2 = T1
        T1 = J1 - Y
    return J1


def ndimage_convolve_constant(img: np.ndarray, weights: np.ndarray) -> np.ndarray:
    """scipy.ndimage.convolve(img, weights, mode='constant') for a float64 image
    and a 2-D kernel, restated (scipy/ndimage/_filters.py _correlate_or_convolve
    + src/ni_filters.c NI_Correlate): the weights are flipped, the origin of an
    even axis moves by one (so the kernel centre is index size // 2 of the
    UNFLIPPED kernel), taps with |w| <= DBL_EPSILON are dropped, and the
    remaining taps are accumulated in row-major order of the flipped kernel into
    a float64 that starts at 0; outside the image the input is 0.
    Pinned bit-exact against scipy itself in tests/test_preprocess_oracle.py."""
    img = np.asarray(img, dtype=np.float64)
    w = np.asarray(weights, dtype=np.float64)
    kh, kw = w.shape
    wf = w[::-1, ::-1]
    cy = kh // 2 - (0 if kh & 1 else 1)        # centre inside the flipped kernel
    cx = kw // 2 - (0 if kw & 1 else 1)
    H, W = img.shape
    pad = np.zeros((H + kh - 1, W + kw - 1))
    pad[cy:cy + H, cx:cx + W] = img
    out = np.zeros((H, W))
    for ky in range(kh):
        for kx in range(kw):
            if abs(wf[ky, kx]) > np.finfo(float).eps:
                out = out + pad[ky:ky + H, kx:kx + W] * wf[ky, kx]
    return out


def to_uint16(values: np.ndarray) -> np.ndarray:
    """``float64.astype(np.uint16)`` as x86 numpy builds do it: truncate toward
    zero to int32 (cvttsd2si; out of range or NaN gives 0x80000000) and keep
    the low 16 bits."""
    v = np.asarray(values, dtype=np.float64)
    ok = np.isfinite(v) & (np.abs(v) < 2147483648.0)
    t = np.where(ok, np.trunc(np.where(ok, v, 0.0)), 0.0).astype(np.int64)
    return (t & 0xffff).astype(np.uint16)


def preprocess_image(inputImage: np.ndarray, highpass_sigma: float = 3,
                     decon_sigma: float = 2, decon_filter_size: int = None,
                     decon_iterations: int = 20) -> np.ndarray:
    """DeconvolutionPreprocess._preprocess_image (preprocess.py:133-140) with
    the defaults of ``__init__`` (:45-53)."""
    if decon_filter_size is None:
        decon_filter_size = filter_size_for_sigma(decon_sigma)
    hp = high_pass_filter(inputImage, filter_size_for_sigma(highpass_sigma),
                          highpass_sigma).astype(np.float64)
    return to_uint16(deconvolve_lucyrichardson(
        hp, decon_filter_size, decon_sigma, decon_iterations))


def pixel_histogram(images: np.ndarray) -> np.ndarray:
    """preprocess.py:114-129 for the images of one bit: counts over the bins
    ``np.arange(0, 65535)`` (65534 bins; the last one holds 65533 and 65534,
    65535 falls outside)."""
    v = np.asarray(images, dtype=np.uint16).ravel()
    c = np.bincount(v, minlength=65536).astype(np.float64)
    out = c[:65534].copy()
    out[65533] += c[65534]
    return out


# --------------------------------------------------------------------------
# Guo et al. variant (DeconvolutionPreprocessGuo, preprocess.py:143-164)
# --------------------------------------------------------------------------

def matlab_gauss2D(shape=(3, 3), sigma=0.5) -> np.ndarray:
    """merlin/util/matlab.py:11-24 (fspecial('gaussian'))."""
    m, n = [(ss - 1.) / 2. for ss in shape]
    y, x = np.ogrid[-m:m + 1, -n:n + 1]
    h = np.exp(-(x * x + y * y) / (2. * sigma * sigma))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    sumh = h.sum()
    if sumh != 0:
        h /= sumh
    return h


def calculate_projectors(windowSize: int, sigmaG: float):
    """deconvolve.py:13-67: Gaussian forward projector and the
    Wiener-Butterworth back projector of Guo et al."""
    pf = matlab_gauss2D(shape=(windowSize, windowSize), sigma=sigmaG)
    pfFFT = np.fft.fft2(pf)
    alpha, beta, n = 0.001, 0.001, 8
    kc = 1.0 / (0.5 * 2.355 * sigmaG)
    kv = np.fft.fftfreq(pfFFT.shape[0])
    kx = np.tile(kv, (kv.size, 1))
    ky = np.transpose(kx)
    kk = np.sqrt(kx * kx + ky * ky)
    bWiener = pfFFT / (np.abs(pfFFT) * np.abs(pfFFT) + alpha)
    eps = np.sqrt(1.0 / (beta * beta) - 1)
    kkSqr = kk * kk / (kc * kc)
    bBWorth = 1.0 / np.sqrt(1.0 + eps * eps * np.power(kkSqr, n))
    pb = np.real(np.fft.ifft2(bWiener * bBWorth))
    return [pf, pb]


def filter2d_f64_replicate(img: np.ndarray, kernel: np.ndarray) -> np.ndarray:
    """cv2.filter2D(img, -1, kernel, borderType=BORDER_REPLICATE) on float64:
    correlation anchored at the kernel centre; OpenCV's Filter2D adds the
    non-zero taps in row-major kernel order, every product and sum rounded on
    its own.  Parity unpinned (cv2 absent)."""
    assert img.dtype == np.float64 and img.ndim == 2
    kh, kw = kernel.shape
    ry, rx = kh // 2, kw // 2
    h, w = img.shape
    p = np.pad(img, ((ry, ry), (rx, rx)), mode='edge')
    out = np.zeros((h, w))
    for i in range(kh):
        for j in range(kw):
            if kernel[i, j] != 0:
                out = out + kernel[i, j] * p[i:i + h, j:j + w]
    return out


def deconvolve_lucyrichardson_guo(image: np.ndarray, windowSize: int,
                                  sigmaG: float, iterationCount: int
                                  ) -> np.ndarray:
    """deconvolve.py:136-174"""
    pf, pb = calculate_projectors(windowSize, sigmaG)
    eps = 1.0e-6
    i_max = 2 ** 16 - 1
    image = np.asarray(image, dtype=np.float64)
    ek = np.copy(image)
    np.clip(ek, eps, None, ek)
    for _ in range(iterationCount):
        ekf = filter2d_f64_replicate(ek, pf)
        np.clip(ekf, eps, i_max, ekf)
        ek = ek * filter2d_f64_replicate(image / ekf, pb)
        np.clip(ek, eps, i_max, ek)
    return ek
