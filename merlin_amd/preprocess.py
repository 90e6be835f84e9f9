"""MI355X preprocess stage: what produces the decoder's uint16 stack.

Mirrors, for the image arithmetic only (no task framework, no data set I/O):

* ``merlin/util/imagefilters.py:9-28``    ``high_pass_filter``
* ``merlin/util/deconvolve.py:70-133``    ``deconvolve_lucyrichardson``
* ``merlin/analysis/preprocess.py:40-140`` ``DeconvolutionPreprocess``:
  parameter names and defaults, ``_high_pass_filter``, ``_preprocess_image``
  and the per-bit pixel histogram of ``_run_analysis``.

All arithmetic runs in libmerfish_hip.so (csrc/preprocess.hip).  There is no
CPU fallback: without the library or a GPU the calls raise.
"""
from typing import Optional, Tuple

import numpy as np
import torch

from merlin_amd import _lib

HIST_BINS = _lib.MF_HIST_BINS
# working set per pixel and image inside mf_deconvolve_lucyrichardson
_BYTES_PER_PIXEL = 26


def algorithmic_bytes_per_pixel(iterations: int = 20) -> int:
    """HBM bytes per pixel the preprocess stage has to move:
    high-pass 4, Lucy-Richardson iteration 0 reads the observed image and
    writes J (10), iteration 1 also reads J1 (18), iteration 2 J1 and J2
    (26), every later one J1, J2, J3 and the observed image (34); the last one
    writes the uint16 hand-over instead of J (28) from four iterations on,
    below that a separate hand-over pass reads J and writes uint16 (10)."""
    per_iteration = [min(10 + 8 * i, 34) for i in range(iterations)]
    if iterations >= 4:
        per_iteration[-1] -= 6
        return 4 + sum(per_iteration)
    return 4 + sum(per_iteration) + 10


_WORKSPACE_TARGET = 16 << 30


def default_batch(H: int, W: int) -> int:
    """Images deconvolved per launch: up to 16 GiB of workspace (26 B per pixel
    and image), between 1 and 128 -- the 112 frames of a 7-z 16-bit FOV of
    2048 x 2048 go through in one batch (12 GB).  Measured per 2048 x 2048
    frame (r02, 112 frames): 0.609 ms at batch 112, 0.614 at 56, 0.625 at 28,
    0.647 at 16, 1.5 at 1 (too few waves to fill the chip; every launch ends
    with a partly filled round of strips, and short batches cut the frames
    into more segments with more halo rows)."""
    return max(1, min(128, _WORKSPACE_TARGET // (_BYTES_PER_PIXEL * H * W)))


def even_batch(n: int, H: int, W: int, batch: Optional[int] = None) -> int:
    """Batch size for n frames: the default (or given) size evened out so that
    the last launch is as full as the others (300 frames -> 3 x 100, not
    2 x 128 + 44)."""
    if batch is None:
        batch = default_batch(H, W)
    batch = max(1, min(int(batch), max(n, 1)))
    launches = -(-max(n, 1) // batch)
    return -(-max(n, 1) // launches)


def _device(device=None):
    if device is not None:
        return torch.device(device)
    if not torch.cuda.is_available():
        raise _lib.MerfishHipError(
            'no GPU visible: merlin_amd has no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def _stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream


# One workspace per (device, stream), reused by every call on that stream (up
# to ~12 GiB each at the default batch).  release_workspaces() returns them to
# torch's allocator; a long-lived process that preprocesses on many side
# streams should call it when a stream retires.
_workspaces = {}


def _workspace(device, batch, H, W):
    nbytes = _lib.check(_lib.lib().mf_preprocess_workspace_bytes(batch, H, W))
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def release_workspaces(device=None) -> int:
    """Drop the cached deconvolution workspaces (of one device, or all);
    returns the number of bytes released.  Work already queued keeps its
    buffer alive through torch's stream-ordered allocator."""
    index = None if device is None else torch.device(device).index
    freed = 0
    for key in [k for k in _workspaces if index is None or k[0] == index]:
        freed += _workspaces.pop(key).numel()
    return freed


def _as_images(t: torch.Tensor):
    if t.dim() < 2:
        raise ValueError('image must have at least two dimensions')
    H, W = t.shape[-2:]
    return t.reshape(-1, H, W).contiguous(), t.shape


def high_pass_filter_device(image: torch.Tensor, windowSize: int,
                            sigma: float) -> torch.Tensor:
    """imagefilters.high_pass_filter for uint16 images [..., H, W] on the GPU."""
    if image.dtype != torch.uint16:
        raise NotImplementedError('high_pass_filter is built for uint16 images')
    flat, shape = _as_images(image)
    out = torch.empty_like(flat)
    n, H, W = flat.shape
    with torch.cuda.device(flat.device):
        _lib.check(_lib.lib().mf_high_pass_filter(
            flat.data_ptr(), n, H, W, int(windowSize), float(sigma),
            out.data_ptr(), _stream_ptr(flat.device)))
    return out.reshape(shape)


def high_pass_filter(image: np.ndarray, windowSize: int,
                     sigma: float) -> np.ndarray:
    """Drop-in for merlin.util.imagefilters.high_pass_filter (uint16 in,
    uint16 out, numpy)."""
    image = np.asarray(image)
    if image.dtype != np.uint16:
        raise NotImplementedError('high_pass_filter is built for uint16 images')
    dev = _device()
    t = torch.from_numpy(np.ascontiguousarray(image)).to(dev)
    return high_pass_filter_device(t, windowSize, sigma).cpu().numpy()


def deconvolve_lucyrichardson_device(
        image: torch.Tensor, windowSize: int, sigmaG: float,
        iterationCount: int, out_dtype=torch.float64,
        histogram: Optional[torch.Tensor] = None, hist_group: int = 1,
        batch: Optional[int] = None, return_lambdas: bool = False,
        gaussianFilter: Optional[np.ndarray] = None):
    """deconvolve_lucyrichardson for images [..., H, W] (uint16 or float64) on
    the GPU.  ``out_dtype`` float64 is the reference's return value, uint16 its
    ``.astype(np.uint16)``.  ``histogram`` (int64 [rows, 65534], zeroed by the
    caller) receives the counts of image i in row i // hist_group.

    An odd ``windowSize`` is the reference's cv2.GaussianBlur branch, an even
    one its scipy.ndimage.convolve branch with the matlab_gauss2D kernel
    (deconvolve.py:102-104, 118-119, 126-128).  ``gaussianFilter`` replaces
    that kernel (even windows only): np.exp differs in the last bit between
    numpy builds, so a caller comparing with results made elsewhere passes the
    kernel that was used there."""
    if image.dtype == torch.uint16:
        dt = _lib.MF_DTYPE_U16
    elif image.dtype == torch.float64:
        dt = _lib.MF_DTYPE_F64
    else:
        raise NotImplementedError('image must be uint16 or float64')
    if out_dtype not in (torch.float64, torch.uint16):
        raise ValueError('out_dtype must be float64 or uint16')
    flat, shape = _as_images(image)
    n, H, W = flat.shape
    dev = flat.device
    batch = even_batch(n, H, W, batch)
    out = torch.empty((n, H, W), dtype=out_dtype, device=dev)
    lam = torch.zeros((n, max(int(iterationCount), 1)), dtype=torch.float64,
                      device=dev) if return_lambdas else None
    if histogram is not None:
        rows = (n + hist_group - 1) // hist_group
        if histogram.dtype != torch.int64 or not histogram.is_contiguous() or \
                tuple(histogram.shape) != (rows, HIST_BINS):
            raise ValueError('histogram must be contiguous int64 [{}, {}]'
                             .format(rows, HIST_BINS))
    tail = (int(iterationCount), out.data_ptr(),
            _lib.MF_DTYPE_F64 if out_dtype == torch.float64 else _lib.MF_DTYPE_U16,
            None if histogram is None else histogram.data_ptr(), int(hist_group),
            None if lam is None else lam.data_ptr(), batch)
    with torch.cuda.device(dev):
        ws = _workspace(dev, batch, H, W)
        if int(windowSize) % 2 == 1:
            if gaussianFilter is not None:
                raise ValueError('gaussianFilter only applies to even windows')
            _lib.check(_lib.lib().mf_deconvolve_lucyrichardson(
                flat.data_ptr(), dt, n, H, W, int(windowSize), float(sigmaG),
                *tail, ws.data_ptr(), ws.numel(), _stream_ptr(dev)))
        else:
            import ctypes
            w = int(windowSize)
            kernel = matlab_gauss2D((w, w), sigmaG) if gaussianFilter is None \
                else gaussianFilter
            kernel = np.ascontiguousarray(kernel, dtype=np.float64)
            if kernel.shape != (w, w):
                raise ValueError('gaussianFilter must be {} x {}'.format(w, w))
            _lib.check(_lib.lib().mf_deconvolve_lucyrichardson_kernel2d(
                flat.data_ptr(), dt, n, H, W, w,
                kernel.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                *tail, ws.data_ptr(), ws.numel(), _stream_ptr(dev)))
    out = out.reshape(shape)
    return (out, lam) if return_lambdas else out


def deconvolve_lucyrichardson(image: np.ndarray, windowSize: int,
                              sigmaG: float, iterationCount: int,
                              gaussianFilter: Optional[np.ndarray] = None
                              ) -> np.ndarray:
    """Drop-in for merlin.util.deconvolve.deconvolve_lucyrichardson (float64
    numpy in and out; odd and even windowSize)."""
    dev = _device()
    t = torch.from_numpy(np.ascontiguousarray(image, dtype=np.float64)).to(dev)
    return deconvolve_lucyrichardson_device(
        t, windowSize, sigmaG, iterationCount,
        gaussianFilter=gaussianFilter).cpu().numpy()


def matlab_gauss2D(shape=(3, 3), sigma=0.5) -> np.ndarray:
    """MATLAB's fspecial('gaussian', shape, sigma) (merlin/util/matlab.py:11-24):
    a Gaussian sampled on the window's centred integer grid, entries below
    eps * max dropped, normalised to unit sum."""
    rows, cols = shape
    dy = np.arange(rows, dtype=np.float64)[:, None] - (rows - 1.0) / 2.0
    dx = np.arange(cols, dtype=np.float64)[None, :] - (cols - 1.0) / 2.0
    window = np.exp(-(dx * dx + dy * dy) / (2.0 * sigma * sigma))
    window = np.where(window < np.finfo(np.float64).eps * window.max(),
                      0.0, window)
    total = window.sum()
    return window / total if total != 0 else window


def calculate_projectors(windowSize: int, sigmaG: float) -> list:
    """deconvolve.py:13-67: the Gaussian forward projector and the
    Wiener-Butterworth back projector of Guo et al. (alpha = beta = 0.001,
    order 8, cut-off 1 / (0.5 * 2.355 * sigma)).  Host work: two windowSize^2
    FFTs."""
    pf = matlab_gauss2D(shape=(windowSize, windowSize), sigma=sigmaG)
    spectrum = np.fft.fft2(pf)
    alpha, beta, order = 0.001, 0.001, 8
    cutoff = 1.0 / (0.5 * 2.355 * sigmaG)
    freq = np.fft.fftfreq(spectrum.shape[0])
    kx = np.tile(freq, (freq.size, 1))
    radius = np.sqrt(kx * kx + kx.T * kx.T)
    wiener = spectrum / (np.abs(spectrum) * np.abs(spectrum) + alpha)
    ripple = np.sqrt(1.0 / (beta * beta) - 1)
    butterworth = 1.0 / np.sqrt(
        1.0 + ripple * ripple * np.power(radius * radius / (cutoff * cutoff), order))
    pb = np.real(np.fft.ifft2(wiener * butterworth))
    return [pf, pb]


def deconvolve_lucyrichardson_guo_device(
        image: torch.Tensor, windowSize: int, sigmaG: float,
        iterationCount: int, out_dtype=torch.float64,
        histogram: Optional[torch.Tensor] = None, hist_group: int = 1,
        batch: Optional[int] = None) -> torch.Tensor:
    """deconvolve_lucyrichardson_guo (deconvolve.py:136-174) for images
    [..., H, W] (uint16 or float64) on the GPU."""
    import ctypes
    if image.dtype == torch.uint16:
        dt = _lib.MF_DTYPE_U16
    elif image.dtype == torch.float64:
        dt = _lib.MF_DTYPE_F64
    else:
        raise NotImplementedError('image must be uint16 or float64')
    if out_dtype not in (torch.float64, torch.uint16):
        raise ValueError('out_dtype must be float64 or uint16')
    if windowSize % 2 != 1:
        raise NotImplementedError('projector windows must be odd')
    pf, pb = [np.ascontiguousarray(k, dtype=np.float64)
              for k in calculate_projectors(windowSize, sigmaG)]
    flat, shape = _as_images(image)
    n, H, W = flat.shape
    dev = flat.device
    batch = even_batch(n, H, W, batch)
    out = torch.empty((n, H, W), dtype=out_dtype, device=dev)
    dp = ctypes.POINTER(ctypes.c_double)
    with torch.cuda.device(dev):
        ws = _workspace(dev, batch, H, W)
        _lib.check(_lib.lib().mf_deconvolve_lucyrichardson_guo(
            flat.data_ptr(), dt, n, H, W, int(windowSize),
            pf.ctypes.data_as(dp), pb.ctypes.data_as(dp), int(iterationCount),
            out.data_ptr(),
            _lib.MF_DTYPE_F64 if out_dtype == torch.float64 else _lib.MF_DTYPE_U16,
            None if histogram is None else histogram.data_ptr(), int(hist_group),
            batch, ws.data_ptr(), ws.numel(), _stream_ptr(dev)))
    return out.reshape(shape)


def deconvolve_lucyrichardson_guo(image: np.ndarray, windowSize: int,
                                  sigmaG: float,
                                  iterationCount: int) -> np.ndarray:
    """Drop-in for merlin.util.deconvolve.deconvolve_lucyrichardson_guo."""
    dev = _device()
    t = torch.from_numpy(np.ascontiguousarray(image, dtype=np.float64)).to(dev)
    return deconvolve_lucyrichardson_guo_device(
        t, windowSize, sigmaG, iterationCount).cpu().numpy()


class DeconvolutionPreprocess(object):
    """Image arithmetic of merlin.analysis.preprocess.DeconvolutionPreprocess.

    ``parameters`` takes the reference's keys with the reference's defaults
    (preprocess.py:45-53): highpass_sigma 3, decon_sigma 2, decon_filter_size
    2*ceil(2*decon_sigma)+1, decon_iterations 20."""

    def __init__(self, parameters=None, device=None, batch=None):
        self.parameters = dict(parameters or {})
        if 'highpass_sigma' not in self.parameters:
            self.parameters['highpass_sigma'] = 3
        if 'decon_sigma' not in self.parameters:
            self.parameters['decon_sigma'] = 2
        if 'decon_filter_size' not in self.parameters:
            self.parameters['decon_filter_size'] = \
                int(2 * np.ceil(2 * self.parameters['decon_sigma']) + 1)
        if 'decon_iterations' not in self.parameters:
            self.parameters['decon_iterations'] = 20
        self._highPassSigma = self.parameters['highpass_sigma']
        self._deconSigma = self.parameters['decon_sigma']
        self._deconIterations = self.parameters['decon_iterations']
        self._device = device
        self._batch = batch

    # ------------------------------------------------ reference-shaped API --

    def _high_pass_filter(self, inputImage: np.ndarray) -> np.ndarray:
        """preprocess.py:103-108 (returns float64 like the reference)."""
        highPassFilterSize = int(2 * np.ceil(2 * self._highPassSigma) + 1)
        hpImage = high_pass_filter(inputImage, highPassFilterSize,
                                   self._highPassSigma)
        return hpImage.astype(float)

    def _preprocess_image(self, inputImage: np.ndarray) -> np.ndarray:
        """preprocess.py:133-140: one camera frame -> uint16."""
        inputImage = np.asarray(inputImage)
        if inputImage.dtype != np.uint16:
            raise NotImplementedError('camera frames must be uint16')
        dev = _device(self._device)
        t = torch.from_numpy(np.ascontiguousarray(inputImage)).to(dev)
        out, _ = self.preprocess_device(t, histogram=False)
        return out.cpu().numpy()

    # ------------------------------------------------- device-resident API --

    def preprocess_device(self, images: torch.Tensor, histogram: bool = True,
                          hist_group: Optional[int] = None
                          ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """``_preprocess_image`` for every uint16 image of ``images``
        [..., H, W] on the GPU, e.g. the warped stack [B,Z,H,W] of one FOV:
        the result is what ``get_processed_image_set`` returns
        (preprocess.py:79-93) and what the decoder reads.

        With ``histogram`` the second value is the int64 pixel histogram
        [rows, 65534] of ``_run_analysis`` (preprocess.py:114-129);
        ``hist_group`` images share a row (default: the length of the second
        dimension of a 4-D stack, i.e. one row per bit)."""
        if images.dtype != torch.uint16:
            raise NotImplementedError('camera frames must be uint16')
        flat, shape = _as_images(images)
        n, H, W = flat.shape
        dev = flat.device
        batch = even_batch(n, H, W, self._batch)
        if hist_group is None:
            hist_group = shape[1] if len(shape) == 4 else 1
        out = torch.empty_like(flat)
        hist = None
        if histogram:
            hist = torch.zeros(((n + hist_group - 1) // hist_group, HIST_BINS),
                               dtype=torch.int64, device=dev)
        if int(self.parameters['decon_filter_size']) % 2 == 0:
            # the reference's ndimage.convolve branch: high-pass, then the
            # K x K Lucy-Richardson entry point
            highPassFilterSize = int(2 * np.ceil(2 * self._highPassSigma) + 1)
            hp = high_pass_filter_device(flat, highPassFilterSize,
                                         self._highPassSigma)
            out = deconvolve_lucyrichardson_device(
                hp, int(self.parameters['decon_filter_size']),
                self._deconSigma, self._deconIterations,
                out_dtype=torch.uint16, histogram=hist,
                hist_group=hist_group, batch=self._batch)
            return out.reshape(shape), hist
        with torch.cuda.device(dev):
            ws = _workspace(dev, batch, H, W)
            _lib.check(_lib.lib().mf_preprocess_images(
                flat.data_ptr(), n, H, W, float(self._highPassSigma),
                int(self.parameters['decon_filter_size']),
                float(self._deconSigma), int(self._deconIterations),
                out.data_ptr(), None if hist is None else hist.data_ptr(),
                int(hist_group), batch, ws.data_ptr(), ws.numel(),
                _stream_ptr(dev)))
        return out.reshape(shape), hist


class DeconvolutionPreprocessGuo(DeconvolutionPreprocess):
    """merlin.analysis.preprocess.DeconvolutionPreprocessGuo (preprocess.py:
    143-164): the same stage with the Guo et al. iteration, 2 iterations unless
    the caller sets ``decon_iterations``."""

    def __init__(self, parameters=None, device=None, batch=None):
        super().__init__(parameters, device, batch)
        if 'decon_iterations' not in (parameters or {}):
            self.parameters['decon_iterations'] = 2
        self._deconIterations = self.parameters['decon_iterations']

    def preprocess_device(self, images: torch.Tensor, histogram: bool = True,
                          hist_group: Optional[int] = None
                          ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        if images.dtype != torch.uint16:
            raise NotImplementedError('camera frames must be uint16')
        flat, shape = _as_images(images)
        n, H, W = flat.shape
        if hist_group is None:
            hist_group = shape[1] if len(shape) == 4 else 1
        hist = None
        if histogram:
            hist = torch.zeros(((n + hist_group - 1) // hist_group, HIST_BINS),
                               dtype=torch.int64, device=flat.device)
        highPassFilterSize = int(2 * np.ceil(2 * self._highPassSigma) + 1)
        hp = high_pass_filter_device(flat, highPassFilterSize,
                                     self._highPassSigma)
        out = deconvolve_lucyrichardson_guo_device(
            hp, int(self.parameters['decon_filter_size']), self._deconSigma,
            self._deconIterations, out_dtype=torch.uint16, histogram=hist,
            hist_group=hist_group, batch=self._batch)
        return out.reshape(shape), hist
