"""CPU oracle for the MERFISH pixel-decode hot path.  TEST INFRASTRUCTURE ONLY.

This module is a plain numpy/scipy restatement of the reference algorithm
(emanuega/MERlin v0.1.6).  It is the *checker* for the HIP path: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  Nothing under ``merlin_amd/`` imports it and the
product path never falls back to it.

Reference files restated (file:line of /root/reference):

* ``merlin/util/decoding.py:17-22``     normalize
* ``merlin/util/decoding.py:46-131``    decode_pixels
* ``merlin/util/decoding.py:133-252``   extract_barcodes_with_index
* ``merlin/util/decoding.py:254-276``   _calculate_normalized_barcodes
* ``merlin/util/decoding.py:290-340``   extract_refactors
* ``merlin/util/decoding.py:342-377``   _extract_backgrounds
* ``merlin/analysis/optimize.py:140-155, 311-378`` scale-factor / background reduction
* ``merlin/analysis/globalalign.py:141-148``  fov_coordinate_array_to_global

Third-party arithmetic that is NOT under /root/reference and is restated from
its published algorithm:

* ``cv2.GaussianBlur`` on ``uint16`` (opencv-python>=3, unpinned in
  requirements.txt:6; call site decoding.py:96-97).  OpenCV 4.x runs its
  bit-exact fixed-point path for CV_16U: the kernel from
  getGaussianKernelBitExact is converted to 16.16 fixed point with error
  diffusion so that it sums to exactly 65536, the horizontal pass is an exact
  u32 sum, the vertical pass an exact u64 sum and the result is
  ``(acc + 2**31) >> 32``; border BORDER_REFLECT_101.  cv2 is absent from this
  container, so this stage is **parity unpinned** (pinned only by
  self-consistency and hand-derived known-answer tests).
* ``sklearn.neighbors.NearestNeighbors(1, 'ball_tree')`` (decoding.py:112-116):
  exact float64 Euclidean 1-NN.  Restated as brute force; exact ties resolve
  to the lowest codeword index (BallTree resolves them by traversal order).
* ``skimage.measure.label / regionprops`` (decoding.py:165-168): full
  connectivity (8 in 2-D, 26 in 3-D), labels numbered in raster order of each
  region's first pixel.  Restated with ``scipy.ndimage.label``.

Pinning status: everything after the blur is pinned by golden vectors generated
by running the reference's own ``decoding.py`` in this container (see
``oracle/make_goldens.py`` and ``tests/golden/README.md``).
"""
import math

import numpy as np
import pandas
from scipy import ndimage

COLUMN_NAMES = ['barcode_id', 'fov', 'mean_intensity', 'max_intensity',
                'area', 'mean_distance', 'min_distance', 'x', 'y', 'z',
                'global_x', 'global_y', 'global_z', 'cell_index']


# --------------------------------------------------------------------------
# Gaussian low-pass (cv2.GaussianBlur restatement)
# --------------------------------------------------------------------------

def filter_size_for_sigma(sigma: float) -> int:
    """decoding.py:94"""
    return int(2 * np.ceil(2 * sigma) + 1)


def gaussian_kernel_f64(ksize: int, sigma: float) -> np.ndarray:
    """OpenCV getGaussianKernelBitExact for sigma > 0 (double precision)."""
    if sigma <= 0:
        sigma = ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8
    scale2x = -0.5 / (sigma * sigma)
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    k = np.array([math.exp(scale2x * xi * xi) for xi in x], dtype=np.float64)
    s = 0.0
    for v in k:
        s += v
    return k * (1.0 / s)


def gaussian_kernel_fixed16(ksize: int, sigma: float) -> np.ndarray:
    """16.16 fixed-point kernel with error diffusion, sums to exactly 65536
    (OpenCV getGaussianKernelFixedPoint_ED, fractionBits = 16)."""
    kd = gaussian_kernel_f64(ksize, sigma)
    out = np.zeros(ksize, dtype=np.int64)
    half = ksize // 2
    err = 0.0
    total = 0
    for i in range(half):
        adj = kd[i] * 65536.0 + err
        v0 = int(np.rint(adj))          # cvRound: nearest, ties to even
        err = adj - float(v0)
        out[i] = v0
        out[ksize - 1 - i] = v0
        total += 2 * v0
    out[half] = 65536 - total
    return out


def gaussian_blur_u16(img: np.ndarray, ksize: int, sigma: float) -> np.ndarray:
    """Exact integer separable blur of one uint16 plane, BORDER_REFLECT_101."""
    assert img.dtype == np.uint16 and img.ndim == 2
    if ksize == 1:
        return img.copy()
    k = gaussian_kernel_fixed16(ksize, sigma).astype(np.uint64)
    r = ksize // 2
    h, w = img.shape
    if r > h - 1 or r > w - 1:
        raise ValueError('image smaller than blur radius')
    p = np.pad(img.astype(np.uint64), r, mode='reflect')
    hs = np.zeros((h + 2 * r, w), dtype=np.uint64)
    for j in range(ksize):
        hs += k[j] * p[:, j:j + w]
    vs = np.zeros((h, w), dtype=np.uint64)
    for j in range(ksize):
        vs += k[j] * hs[j:j + h, :]
    out = (vs + np.uint64(1 << 31)) >> np.uint64(32)
    return np.minimum(out, 65535).astype(np.uint16)


def gaussian_blur_f32(img: np.ndarray, ksize: int, sigma: float) -> np.ndarray:
    """Float path (cv2 on float32 input): separable float32 filter with the
    float32-rounded kernel, symmetric pairs summed first, centre tap first.
    Parity unpinned (cv2 absent); defined here so GPU and oracle agree."""
    assert img.dtype == np.float32 and img.ndim == 2
    if ksize == 1:
        return img.copy()
    k = gaussian_kernel_f64(ksize, sigma).astype(np.float32)
    r = ksize // 2
    h, w = img.shape
    p = np.pad(img, r, mode='reflect')
    hs = k[r] * p[:, r:r + w]
    for j in range(1, r + 1):
        hs = hs + k[r + j] * (p[:, r - j:r - j + w] + p[:, r + j:r + j + w])
    vs = k[r] * hs[r:r + h, :]
    for j in range(1, r + 1):
        vs = vs + k[r + j] * (hs[r - j:r - j + h, :] + hs[r + j:r + j + h, :])
    return vs.astype(np.float32)


# --------------------------------------------------------------------------
# Codebook
# --------------------------------------------------------------------------

def normalized_codebook(barcodes: np.ndarray) -> np.ndarray:
    """decoding.py:17-22, 254-276: each row divided by its L2 norm (float64);
    all-zero rows are left unchanged."""
    barcodes = np.asarray(barcodes)
    rows = []
    for x in barcodes:
        n = np.linalg.norm(x)
        rows.append(x / n if n > 0 else x.astype(np.float64))
    return np.array(rows, dtype=np.float64)


# --------------------------------------------------------------------------
# decode_pixels
# --------------------------------------------------------------------------

def scaled_traces(filtered: np.ndarray, scale_factors, backgrounds):
    """decoding.py:99-110 on a float32 [B, ...] stack.  Legacy-numpy value
    casting keeps (p - b) / s in float32 with b and s rounded to float32.
    Returns (magnitudes f32 [...], normalized traces f32 [B, ...])."""
    B = filtered.shape[0]
    sf = np.asarray(scale_factors, dtype=np.float64).astype(np.float32)
    bg = np.asarray(backgrounds, dtype=np.float64).astype(np.float32)
    scaled = np.empty(filtered.shape, dtype=np.float32)
    for b in range(B):
        scaled[b] = (filtered[b] - bg[b]) / sf[b]
    acc = np.zeros(filtered.shape[1:], dtype=np.float32)
    for b in range(B):                      # sequential f32 sum of squares
        acc = acc + scaled[b] * scaled[b]
    mag = np.sqrt(acc).astype(np.float32)
    mag[mag == 0] = 1
    norm = scaled / mag[None]
    return mag, norm.astype(np.float32)


def nearest_codeword(norm_traces: np.ndarray, decoding_matrix: np.ndarray,
                     chunk: int = 1 << 16, return_gap: bool = False):
    """decoding.py:112-116 as exact float64 brute force.
    norm_traces f32 [B, P] -> (distance f64 [P], index i64 [P]); with
    return_gap also the squared-distance gap to the runner-up codeword."""
    B, P = norm_traces.shape
    N = decoding_matrix.shape[0]
    dist = np.empty(P, dtype=np.float64)
    idx = np.empty(P, dtype=np.int64)
    gap = np.empty(P, dtype=np.float64)
    cm = np.asarray(decoding_matrix, dtype=np.float64)
    for s in range(0, P, chunk):
        x = norm_traces[:, s:s + chunk].astype(np.float64)     # [B, c]
        best = np.full(x.shape[1], np.inf)
        second = np.full(x.shape[1], np.inf)
        bidx = np.zeros(x.shape[1], dtype=np.int64)
        for n in range(N):
            d2 = np.zeros(x.shape[1], dtype=np.float64)
            for b in range(B):                                  # sequential
                t = x[b] - cm[n, b]
                d2 = d2 + t * t
            m = d2 < best                                       # lowest n wins
            second = np.where(m, best, np.minimum(second, d2))
            best[m] = d2[m]
            bidx[m] = n
        dist[s:s + chunk] = np.sqrt(best)
        idx[s:s + chunk] = bidx
        gap[s:s + chunk] = second - best
    if return_gap:
        return dist, idx, gap
    return dist, idx


def decode_pixels(image_data: np.ndarray, decoding_matrix: np.ndarray,
                  scale_factors=None, backgrounds=None,
                  distance_threshold: float = 0.5176,
                  magnitude_threshold: float = 1, lowpass_sigma: float = 1,
                  return_gap: bool = False):
    """decoding.py:46-131.  image_data [B, H, W] uint16 (or float32).
    Returns (decoded int16 [H,W], magnitudes f32 [H,W], traces f32 [B,H,W],
    distances f64 [H,W]) and, with return_gap, the squared-distance gap
    between the nearest and second-nearest codeword per pixel."""
    B = image_data.shape[0]
    if scale_factors is None:
        scale_factors = np.ones(B)
    if backgrounds is None:
        backgrounds = np.zeros(B)
    ksize = filter_size_for_sigma(lowpass_sigma)
    filtered = np.zeros(image_data.shape, dtype=np.float32)
    for b in range(B):
        if image_data.dtype == np.uint16:
            filtered[b] = gaussian_blur_u16(image_data[b], ksize, lowpass_sigma)
        else:
            filtered[b] = gaussian_blur_f32(
                image_data[b].astype(np.float32), ksize, lowpass_sigma)
    shape = filtered.shape[1:]
    mag, norm = scaled_traces(filtered.reshape(B, -1), scale_factors,
                              backgrounds)
    dist, idx, gap = nearest_codeword(norm, decoding_matrix, return_gap=True)
    decoded = np.where(dist <= distance_threshold, idx, -1).astype(np.int16)
    decoded[mag < magnitude_threshold] = -1
    out = (decoded.reshape(shape), mag.reshape(shape),
           norm.reshape((B,) + shape), dist.reshape(shape))
    if return_gap:
        out = out + (gap.reshape(shape),)
    return out


# --------------------------------------------------------------------------
# connected components + region records
# --------------------------------------------------------------------------

def label_regions(mask: np.ndarray):
    """skimage.measure.label(mask) with default (full) connectivity: labels
    1..n in raster order of each region's first pixel."""
    structure = np.ones((3,) * mask.ndim, dtype=bool)
    return ndimage.label(mask, structure=structure)


def _region_coords(labels: np.ndarray, n: int):
    """list of coordinate arrays [npix, ndim] in raster order per label."""
    nz = np.nonzero(labels)
    lab = labels[nz]
    order = np.argsort(lab, kind='stable')
    coords = np.stack(nz, axis=1)[order]
    counts = np.bincount(lab, minlength=n + 1)[1:]
    return np.split(coords, np.cumsum(counts)[:-1]) if n > 0 else []


def fov_coordinate_array_to_global(tform: np.ndarray, centroids: np.ndarray):
    """globalalign.py:141-148 with a given 3x3 transform."""
    to_global = np.ones(centroids.shape)
    to_global[:, [0, 1]] = centroids[:, [1, 2]]
    g = np.matmul(tform, to_global.T).T[:, [2, 0, 1]]
    g[:, 0] = centroids[:, 0]
    return g


def extract_barcodes_with_index(barcode_index: int, decoded: np.ndarray,
                                magnitudes: np.ndarray, traces: np.ndarray,
                                distances: np.ndarray, fov: int,
                                crop_width: int, z_index=None,
                                global_transform=None, minimum_area: int = 0
                                ) -> pandas.DataFrame:
    """decoding.py:133-252."""
    is3d = traces.ndim == 4
    B = traces.shape[1] if is3d else traces.shape[0]
    icols = ['intensity_{}'.format(i) for i in range(B)]
    labels, n = label_regions(decoded == barcode_index)
    if n == 0:
        return pandas.DataFrame(columns=COLUMN_NAMES + icols)
    regions = _region_coords(labels, n)

    centroids = np.zeros((n, 3))
    ia = np.zeros((n, 3))
    dd = np.zeros((n, 2))
    inten = np.zeros((n, B), dtype=np.float32)
    for i, c in enumerate(regions):
        if is3d:
            m = magnitudes[c[:, 0], c[:, 1], c[:, 2]]
            m64 = m.astype(np.float64)
            lo = c.min(axis=0)
            cen = ((c - lo) * m64[:, None]).sum(axis=0) / m64.sum() + lo
            centroids[i] = cen[[0, 2, 1]]           # decoding.py:188
            ia[i] = [np.mean(m), np.max(m), len(c)]
            dv = distances[c[:, 0], c[:, 1], c[:, 2]]
            tr = traces[c[:, 0], :, c[:, 1], c[:, 2]]
        else:
            m64 = magnitudes[c[:, 0], c[:, 1]].astype(np.float64)
            if len(c) > 1:
                w = m64 / m64.sum()
                cr = (c[:, 0] * w).sum()
                cc = (c[:, 1] * w).sum()
            else:
                cr, cc = c[0, 0], c[0, 1]
            centroids[i] = [z_index, cc, cr]        # decoding.py:209-211
            ia[i] = [m64.mean(), m64.max(), len(c)]
            dv = distances[c[:, 0], c[:, 1]]
            tr = traces[:, c[:, 0], c[:, 1]].T
        dv = np.asarray(dv, dtype=np.float64)
        dd[i] = [np.mean(dv), np.min(dv)] if len(c) > 1 else [dv[0], dv[0]]
        tr = np.ascontiguousarray(tr, dtype=np.float32)   # [npix, B]
        inten[i] = np.mean(tr, 0) if len(c) > 1 else tr[0]

    if global_transform is not None:
        gc = fov_coordinate_array_to_global(global_transform, centroids)
    else:
        gc = centroids

    df = pandas.DataFrame(np.zeros((n, len(COLUMN_NAMES))),
                          columns=COLUMN_NAMES)
    df['barcode_id'] = barcode_index
    df['fov'] = fov
    df[['mean_intensity', 'max_intensity', 'area']] = ia
    df[['mean_distance', 'min_distance']] = dd
    df[['x', 'y', 'z']] = centroids[:, [1, 2, 0]]
    df[['global_x', 'global_y', 'global_z']] = gc[:, [1, 2, 0]]
    df['cell_index'] = -1
    full = pandas.concat([df, pandas.DataFrame(inten, columns=icols)], axis=1)
    keep = ((full['x'] > crop_width)
            & (full['x'] < decoded.shape[0] - crop_width)
            & (full['y'] > crop_width)
            & (full['y'] < decoded.shape[1] - crop_width)
            & (full['area'] >= minimum_area))
    return full[keep]


def extract_all_barcodes(n_barcodes: int, *args, **kwargs) -> pandas.DataFrame:
    """analysis/decode.py:225-230: concat over every barcode index."""
    return pandas.concat([extract_barcodes_with_index(i, *args, **kwargs)
                          for i in range(n_barcodes)])


# --------------------------------------------------------------------------
# refactors (OptimizeIteration)
# --------------------------------------------------------------------------

def _regions_for(decoded, b, area_threshold):
    labels, n = label_regions(decoded == b)
    return [c for c in _region_coords(labels, n) if len(c) >= area_threshold]


def extract_backgrounds(decoded, magnitudes, traces, decoding_matrix):
    """decoding.py:342-377 (area threshold hard-coded to 5 at :362)."""
    N, B = decoding_matrix.shape
    sum_min = np.zeros((N, B))
    seen = np.zeros(N)
    for b in range(N):
        regs = _regions_for(decoded, b, 5)
        seen[b] = len(regs)
        for c in regs:
            tm = (traces[:, c[:, 0], c[:, 1]]
                  * magnitudes[c[:, 0], c[:, 1]][None]).astype(np.float32)
            sum_min[b, :] += tm.min(axis=1)
    off = sum_min.copy()
    off[decoding_matrix > 0] = np.nan
    with np.errstate(invalid='ignore', divide='ignore'):
        return np.nansum(off, axis=0) / np.sum(
            (decoding_matrix == 0) * seen[:, np.newaxis], axis=0)


def extract_refactors(decoded, magnitudes, traces, decoding_matrix,
                      refactor_area_threshold: int = 4,
                      extract_backgrounds_flag: bool = False):
    """decoding.py:290-340."""
    N, B = decoding_matrix.shape
    if extract_backgrounds_flag:
        bgr = extract_backgrounds(decoded, magnitudes, traces, decoding_matrix)
    else:
        bgr = np.zeros(B)
    sums = np.zeros((N, B))
    seen = np.zeros(N)
    for b in range(N):
        regs = _regions_for(decoded, b, refactor_area_threshold)
        seen[b] = len(regs)
        for c in regs:
            tm = (traces[:, c[:, 0], c[:, 1]]
                  * magnitudes[c[:, 0], c[:, 1]][None]).astype(np.float32)
            mean_tm = np.mean(np.ascontiguousarray(tm.T), axis=0) - bgr
            sums[b, :] += (mean_tm / np.linalg.norm(mean_tm)) / seen[b]
    sums[decoding_matrix == 0] = np.nan
    with np.errstate(invalid='ignore', divide='ignore'):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', category=RuntimeWarning)
            on_bit = np.nanmean(sums, axis=0)
        refactors = on_bit / np.mean(on_bit)
    return refactors, bgr, seen


# --------------------------------------------------------------------------
# OptimizeIteration reductions
# --------------------------------------------------------------------------

def initial_scale_factors(pixel_histograms: np.ndarray) -> np.ndarray:
    """optimize.py:140-155."""
    out = np.zeros(pixel_histograms.shape[0])
    for i in range(pixel_histograms.shape[0]):
        c = np.cumsum(pixel_histograms[i])
        c = c / c[-1]
        out[i] = np.argmin(np.abs(c - 0.9)) + 2
    return out


def reduce_scale_factors(refactors: np.ndarray, previous: np.ndarray):
    """optimize.py:328-345: refactors[F,B], previous scale factors[F,B]."""
    r = np.array(refactors, dtype=np.float64)
    r[r == 0] = 1
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', category=RuntimeWarning)
        return np.nanmedian(np.multiply(r, previous), axis=0)


def reduce_backgrounds(bg_refactors, previous_backgrounds, previous_factors):
    """optimize.py:358-378."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', category=RuntimeWarning)
        return np.nanmedian(np.add(previous_backgrounds, np.multiply(
            bg_refactors, previous_factors)), axis=0)
