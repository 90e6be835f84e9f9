"""MI355X drop-in for ``merlin.util.decoding`` (reference: merlin/util/decoding.py).

``PixelBasedDecoder`` keeps the reference's constructor, method names,
argument meaning, return tuple and error behaviour (decoding.py:25-377), so
``merlin.analysis.decode.Decode`` and ``optimize.OptimizeIteration`` can use
it unchanged.  All arithmetic runs in the HIP kernels of libmerfish_hip.so;
this module only moves tensors, caches device-resident results and assembles
the pandas tables.  There is no CPU fallback: without the library or a GPU the
calls raise.

Beyond the reference API the class offers device-resident variants
(``decode_pixels_device``, ``extract_barcodes``) that the FOV runner in
``merlin_amd.decode`` uses to keep a whole field of view on the GPU.
"""
import weakref
from typing import Optional, Tuple

import os

import numpy as np
import pandas
import torch

from merlin_amd import _lib

COLUMN_NAMES = ['barcode_id', 'fov', 'mean_intensity', 'max_intensity',
                'area', 'mean_distance', 'min_distance', 'x', 'y', 'z',
                'global_x', 'global_y', 'global_z', 'cell_index']


_NUMPY_DTYPES = {torch.int16: np.int16, torch.float32: np.float32,
                 torch.float64: np.float64}


def normalize(x):
    """decoding.py:17-22"""
    norm = np.linalg.norm(x)
    if norm > 0:
        return x / norm
    else:
        return x


# Frame rows are staged in HBM and moved to the host by the copy engine.
# False has the gather kernel write them over PCIe into mapped pinned memory
# instead, which costs the next FOV's decode kernel 5 % (measured; more with more
# workgroups storing into mapped host memory).
_FRAME_VIA_COPY = True


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _bucketed(shape, dtype, device, quantum=4 << 20):
    """Uninitialised device tensor of ``shape`` carved from an allocation whose
    size is rounded up to ``quantum`` bytes: table sizes differ a little from
    FOV to FOV, and exact sizes would make torch's caching allocator grow (a
    hipMalloc, 5-8 ms) whenever a FOV is slightly larger than every block it
    holds.  The view keeps the allocation alive."""
    n = 1
    for d in shape:
        n *= int(d)
    nbytes = n * torch.empty(0, dtype=dtype).element_size()
    cap = max(quantum, -(-nbytes // quantum) * quantum)
    raw = torch.empty(cap, dtype=torch.uint8, device=device)
    return raw[:nbytes].view(dtype).view(tuple(shape))


def _stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream


class DecodedVolume(object):
    """Device-resident output of one decode call: Z planes of one FOV.

    decoded int16 [Z,H,W], magnitudes f32 [Z,H,W], distances f32|f64 [Z,H,W],
    traces f32 [Z,B,H,W] (or None when not materialised).  ``traces`` may be a
    strided view: the FOV runner stores it pixel-major ([Z,H,W,B] in memory)
    and only at decoded pixels (``traces_decoded_only``); the entries of
    undecoded pixels are then undefined."""

    def __init__(self, decoded, magnitudes, distances, traces, bit_count):
        self.decoded = decoded
        self.magnitudes = magnitudes
        self.distances = distances
        self.traces = traces
        self.bit_count = bit_count
        self.traces_decoded_only = False
        # the barcodes are known to be below the codebook's size (a volume this
        # decoder produced); False for images that came back from the caller
        self.codes_in_codebook = True
        self._tables = {}

    @property
    def shape(self):
        return tuple(self.decoded.shape)


class RegionTable(object):
    """mf_region_table's output, rows in reference order.  The tensors stay on
    the GPU; the host copies are made on first use."""

    def __init__(self, rec, inten, mean_tm=None, min_tm=None):
        self.rec_d = rec
        self.inten_d = inten
        self.mean_tm_d = mean_tm
        self.min_tm_d = min_tm
        self._host = {}

    def _h(self, name):
        if name not in self._host:
            t = getattr(self, name + '_d')
            self._host[name] = None if t is None else t.cpu().numpy()
        return self._host[name]

    rec = property(lambda self: self._h('rec'))
    inten = property(lambda self: self._h('inten'))
    mean_tm = property(lambda self: self._h('mean_tm'))
    min_tm = property(lambda self: self._h('min_tm'))

    def __len__(self):
        return self.rec_d.shape[0]


class PixelBasedDecoder(object):

    def __init__(self, codebook, scaleFactors: np.ndarray = None,
                 backgrounds: np.ndarray = None, device=None):
        self._codebook = codebook
        self._decodingMatrix = self._calculate_normalized_barcodes()
        self._barcodeCount = self._decodingMatrix.shape[0]
        self._bitCount = self._decodingMatrix.shape[1]

        if scaleFactors is None:
            self._scaleFactors = np.ones(self._decodingMatrix.shape[1])
        else:
            self._scaleFactors = scaleFactors.copy()

        if backgrounds is None:
            self._backgrounds = np.zeros(self._decodingMatrix.shape[1])
        else:
            self._backgrounds = backgrounds.copy()

        self.refactorAreaThreshold = 4

        if self._bitCount > _lib.MF_MAX_BITS:
            raise NotImplementedError(
                'codebooks with more than {} bits are not supported'.format(
                    _lib.MF_MAX_BITS))
        self._device = torch.device(device) if device is not None else None
        self._cb_dev = None
        self._recent = []       # [(weakrefs of returned arrays, DecodedVolume)]

    # ------------------------------------------------------------ helpers --

    def _calculate_normalized_barcodes(self, ignoreBlanks=False):
        """decoding.py:254-276 (the includeErrors branch is never used)."""
        barcodeSet = np.asarray(
            self._codebook.get_barcodes(ignoreBlanks=ignoreBlanks))
        return np.array([normalize(x) for x in barcodeSet], dtype=np.float64)

    def _dev(self):
        if self._device is None:
            if not torch.cuda.is_available():
                raise _lib.MerfishHipError(
                    'no GPU visible: merlin_amd has no CPU fallback')
            self._device = torch.device('cuda', torch.cuda.current_device())
        return self._device

    def _codebook_device(self):
        """Device codebook table (mf_codebook_prepare), built once."""
        if self._cb_dev is None:
            import ctypes
            L = _lib.lib()
            N, B = self._decodingMatrix.shape
            nbytes = _lib.check(L.mf_codebook_table_bytes(N, B))
            dev = self._dev()
            with torch.cuda.device(dev):
                table = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                cb = np.ascontiguousarray(self._decodingMatrix,
                                          dtype=np.float64)
                self._cb_flags = _lib.check(L.mf_codebook_prepare(
                    cb.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), N, B,
                    table.data_ptr(), nbytes))
            self._cb_dev = table
        return self._cb_dev

    @staticmethod
    def _f32_params(values, count, name):
        v = np.asarray(values, dtype=np.float64).reshape(-1)
        if v.shape[0] != count:
            raise ValueError('{} must have one entry per bit ({} != {})'
                             .format(name, v.shape[0], count))
        # legacy numpy value casting keeps (p - b) / s in float32
        return np.ascontiguousarray(v.astype(np.float32))

    # -------------------------------------------------------- decode_pixels --

    def decode_pixels_device(self, imageData: torch.Tensor,
                             scaleFactors=None, backgrounds=None,
                             distanceThreshold: float = 0.5176,
                             magnitudeThreshold: float = 1,
                             lowPassSigma: float = 1,
                             returnTraces=True,
                             distanceDtype=torch.float64,
                             keepRunnerUp=False) -> DecodedVolume:
        """decode_pixels on a device-resident stack.

        imageData: uint16 / float32 tensor [B,H,W] or [B,Z,H,W] on the GPU
        (the layout preprocess.get_processed_image_set returns,
        preprocess.py:79-93).  Returns a DecodedVolume (tensors stay on the
        GPU)."""
        L = _lib.lib()
        dev = self._dev()
        if scaleFactors is None:
            scaleFactors = self._scaleFactors
        if backgrounds is None:
            backgrounds = self._backgrounds
        if imageData.dim() == 3:
            imageData = imageData.unsqueeze(1)
        if imageData.dim() != 4:
            raise ValueError('imageData must be [B,H,W] or [B,Z,H,W]')
        B, Z, H, W = imageData.shape
        if B != self._bitCount:
            raise ValueError('image stack has {} bits, codebook {}'.format(
                B, self._bitCount))
        if imageData.dtype == torch.uint16:
            dt = _lib.MF_DTYPE_U16
        elif imageData.dtype == torch.float32:
            dt = _lib.MF_DTYPE_F32
        else:
            raise TypeError('imageData must be uint16 or float32, got {}'
                            .format(imageData.dtype))
        if imageData.device != dev:
            imageData = imageData.to(dev)
        if imageData.stride(3) != 1 or imageData.stride(2) != W:
            imageData = imageData.contiguous()
        sf = self._f32_params(scaleFactors, B, 'scaleFactors')
        bg = self._f32_params(backgrounds, B, 'backgrounds')
        if int(2 * np.ceil(2 * lowPassSigma) + 1) // 2 > _lib.MF_MAX_RADIUS:
            # wider than the fused kernel's blur (sigma > 3): low-pass as a
            # pass of its own, then decode the blurred stack with sigma = 0
            imageData = self._lowpass_device(imageData, dt, lowPassSigma)
            lowPassSigma = 0

        with torch.cuda.device(dev):
            decoded = torch.empty((Z, H, W), dtype=torch.int16, device=dev)
            mags = torch.empty((Z, H, W), dtype=torch.float32, device=dev)
            dist = torch.empty((Z, H, W), dtype=distanceDtype, device=dev)
            flags = 0
            if returnTraces == 'decoded':
                store = torch.empty((Z, H, W, B), dtype=torch.float32,
                                    device=dev)
                traces = store.permute(0, 3, 1, 2)
                flags = _lib.MF_DECODE_TRACES_DECODED_ONLY
            elif returnTraces:
                traces = torch.empty((Z, B, H, W), dtype=torch.float32,
                                     device=dev)
            else:
                traces = None
            ddt = _lib.MF_DTYPE_F64 if distanceDtype == torch.float64 \
                else _lib.MF_DTYPE_F32
            if keepRunnerUp:        # MF_DECODE_KEEP_RUNNER_UP, see the header
                flags |= _lib.MF_DECODE_KEEP_RUNNER_UP
            cb = self._codebook_device()
            import ctypes
            fp = ctypes.POINTER(ctypes.c_float)
            has = traces is not None
            _lib.check(L.mf_decode_pixels(
                imageData.data_ptr(), dt, B, Z, H, W,
                imageData.stride(0), imageData.stride(1),
                cb.data_ptr(), self._barcodeCount, self._cb_flags,
                sf.ctypes.data_as(fp), bg.ctypes.data_as(fp),
                float(lowPassSigma), float(distanceThreshold),
                float(np.float32(magnitudeThreshold)),
                decoded.data_ptr(), mags.data_ptr(), dist.data_ptr(), ddt,
                _ptr(traces), traces.stride(0) if has else 0,
                traces.stride(1) if has else 0,
                traces.stride(3) if has else 1, flags, _stream_ptr(dev)))
        # keep the input alive until the stream has consumed it
        imageData.record_stream(torch.cuda.current_stream(dev))
        vol = DecodedVolume(decoded, mags, dist, traces, B)
        vol.traces_decoded_only = bool(
            flags & _lib.MF_DECODE_TRACES_DECODED_ONLY)
        return vol

    def _lowpass_device(self, imageData: torch.Tensor, dt: int,
                        lowPassSigma: float) -> torch.Tensor:
        """cv2.GaussianBlur of every (bit, z) plane of a contiguous-row stack
        (decoding.py:93-97) through mf_lowpass_images."""
        L = _lib.lib()
        dev = imageData.device
        imageData = imageData.contiguous()
        B, Z, H, W = imageData.shape
        with torch.cuda.device(dev):
            out = torch.empty_like(imageData)
            nb = _lib.check(L.mf_lowpass_workspace_bytes(dt, B * Z, H, W))
            ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
            _lib.check(L.mf_lowpass_images(
                imageData.data_ptr(), dt, B * Z, H, W, H * W,
                float(lowPassSigma), out.data_ptr(), ws.data_ptr(), nb,
                _stream_ptr(dev)))
            cur = torch.cuda.current_stream(dev)
            imageData.record_stream(cur)
            ws.record_stream(cur)
        return out

    def decode_pixels(self, imageData: np.ndarray,
                      scaleFactors: np.ndarray = None,
                      backgrounds: np.ndarray = None,
                      distanceThreshold: float = 0.5176,
                      magnitudeThreshold: float = 1,
                      lowPassSigma: float = 1):
        """Assign barcodes to the pixels in the provided image stack.

        Same contract as decoding.py:46-131: imageData [B,H,W]; returns
        (decodedImage int16 [H,W], pixelMagnitudes float32 [H,W],
        normalizedPixelTraces float32 [B,H,W], distances float64 [H,W]) as
        numpy arrays.  uint16 input takes the integer blur cv2 applies to
        CV_16U; any other dtype is blurred as float32."""
        if isinstance(imageData, torch.Tensor):
            img = imageData
        else:
            arr = np.asarray(imageData)
            if arr.ndim != 3:
                raise ValueError('imageData must be [bits, H, W]')
            if arr.dtype != np.uint16:
                arr = arr.astype(np.float32)
            img = torch.from_numpy(np.ascontiguousarray(arr))
        vol = self.decode_pixels_device(
            img.to(self._dev()), scaleFactors, backgrounds, distanceThreshold,
            magnitudeThreshold, lowPassSigma, True, torch.float64)
        def to_host(t):
            # into an array that numpy owns, so that a caller may make it
            # writeable again
            a = np.empty(tuple(t.shape), dtype=_NUMPY_DTYPES[t.dtype])
            torch.from_numpy(a).copy_(t)
            return a
        decoded = to_host(vol.decoded[0])
        mags = to_host(vol.magnitudes[0])
        traces = to_host(vol.traces[0])
        dist = to_host(vol.distances[0])
        # The arrays are handed out read-only: while they stay so, the
        # device-resident volume is known to hold the same contents and the
        # extraction calls reuse it.  A caller that wants to edit them sets
        # ``flags.writeable = True``; the extraction calls then upload the
        # arrays' current contents (see _lookup).
        for a in (decoded, mags, traces, dist):
            a.flags.writeable = False
        self._remember((decoded, mags, traces, dist), vol, None)
        return decoded, mags, traces, dist

    # ------------------------------------------------ device result caching --

    _CACHED_VOLUMES = 2       # device volumes kept for the extraction calls

    @staticmethod
    def _content_key(decodedImage, pixelMagnitudes, pixelTraces=None,
                     distances=None):
        """Checksum of the images an in-place edit would touch: all of
        decodedImage and pixelMagnitudes (``di[pm < t] = -1`` and the like) and
        a strided sample (every 61st element) of the traces and distances --
        a wholesale rewrite of those is caught, an edit of single trace values
        is not (documented in INTEGRATION.md)."""
        parts = [np.ascontiguousarray(decodedImage),
                 np.ascontiguousarray(pixelMagnitudes)]
        for a in (pixelTraces, distances):
            if a is not None:
                parts.append(np.ascontiguousarray(
                    np.asarray(a).reshape(-1)[::61]))
        try:
            import xxhash
            h = xxhash.xxh3_64()
            for a in parts:
                h.update(a.data)
            return h.intdigest()
        except ImportError:
            import zlib
            k = 0
            for a in parts:
                k = zlib.adler32(a.data, k)
            return k

    def _remember(self, arrays, vol, key):
        refs = tuple(weakref.ref(a) for a in arrays)
        self._recent.append((refs, vol, key))
        del self._recent[:-self._CACHED_VOLUMES]

    def _lookup(self, decodedImage, pixelMagnitudes, pixelTraces, distances):
        """DecodedVolume for the given images.

        A cached device volume is reused only when it is known to hold the
        arrays' CURRENT contents: the arrays are the very objects an earlier
        call saw (identity) and either they are still the read-only arrays
        decode_pixels returned, or a checksum of decodedImage and
        pixelMagnitudes (plus a strided sample of the traces and distances)
        taken at upload time still matches.  Anything else --
        other arrays, arrays made writeable, edited contents -- is uploaded
        again, so the result always follows the arrays as the reference's
        does.  Entries whose arrays are gone are dropped."""
        self._recent = [e for e in self._recent
                        if all(r() is not None for r in e[0])]
        for refs, vol, key in reversed(self._recent):
            d, m, t, ds = (r() for r in refs)
            if not (d is decodedImage and m is pixelMagnitudes
                    and t is pixelTraces
                    and (distances is None or ds is distances)):
                continue
            if key is None:
                if not any(a.flags.writeable for a in (d, m, t, ds)):
                    return vol
            elif key == self._content_key(decodedImage, pixelMagnitudes,
                                          pixelTraces, distances):
                return vol
            self._recent = [e for e in self._recent if e[1] is not vol]
            break
        dev = self._dev()
        dec = np.ascontiguousarray(decodedImage, dtype=np.int16)
        is3d = np.asarray(pixelTraces).ndim == 4
        if not is3d:
            dec = dec[None]
        def up(a, dt=None):
            a = np.ascontiguousarray(a) if dt is None \
                else np.ascontiguousarray(a, dtype=dt)
            import warnings
            with warnings.catch_warnings():
                # read-only arrays are only read here
                warnings.simplefilter('ignore', category=UserWarning)
                return torch.from_numpy(a).to(dev)
        mags = up(pixelMagnitudes, np.float32)
        tr = up(pixelTraces, np.float32)
        if distances is None:
            dist = None
        else:
            darr = np.asarray(distances)
            dist = up(darr, np.float32 if darr.dtype == np.float32
                      else np.float64)
        if not is3d:
            mags, tr = mags[None], tr[None]
            dist = None if dist is None else dist[None]
        vol = DecodedVolume(up(dec), mags, dist, tr, tr.shape[1])
        vol.codes_in_codebook = False
        keep = (decodedImage, pixelMagnitudes, pixelTraces,
                distances if distances is not None else pixelMagnitudes)
        if all(isinstance(a, np.ndarray) for a in keep):
            self._remember(keep, vol,
                           self._content_key(decodedImage, pixelMagnitudes,
                                             pixelTraces, distances))
        return vol

    # --------------------------------------------------------- region table --

    def _label_enqueue(self, vol: DecodedVolume, is3D: bool):
        """Queue the connected-component labelling of a volume and the copy of
        its (pixel, region) counts to the host; returns the pending state for
        _table_enqueue.  Does not block."""
        L = _lib.lib()
        dev = vol.decoded.device
        Z, H, W = vol.shape
        with torch.cuda.device(dev):
            sp = _stream_ptr(dev)
            labels = torch.empty((Z, H, W), dtype=torch.int32, device=dev)
            counts = torch.empty(_lib.MF_LABEL_COUNTS_LEN, dtype=torch.int64,
                                 device=dev)
            _lib.check(L.mf_label_components(
                vol.decoded.data_ptr(), Z, H, W, int(is3D),
                labels.data_ptr(), counts.data_ptr(), sp))
            hcounts = self._pinned((2,), torch.int64)
            hcounts.copy_(counts[:2], non_blocking=True)
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(dev))
        return (labels, counts, hcounts, done)

    def _table_enqueue(self, vol: DecodedVolume, pending, is3D: bool,
                       withRefactorStats: bool) -> RegionTable:
        """Wait for the counts of _label_enqueue (the one host sync of a FOV),
        size the tables and queue mf_region_table."""
        L = _lib.lib()
        labels, _, hcounts, done = pending
        dev = vol.decoded.device
        Z, H, W = vol.shape
        B = vol.bit_count
        done.synchronize()
        n_pix, n_reg = int(hcounts[0]), int(hcounts[1])
        with torch.cuda.device(dev):
            sp = _stream_ptr(dev)
            rec = _bucketed((n_reg, _lib.MF_REC_COLS), torch.float64, dev)
            inten = _bucketed((n_reg, B), torch.float32, dev)
            mean_tm = min_tm = None
            if withRefactorStats:
                mean_tm = _bucketed((n_reg, B), torch.float32, dev)
                min_tm = _bucketed((n_reg, B), torch.float32, dev)
            if n_reg > 0:
                ws_bytes = _lib.check(L.mf_region_workspace_bytes(
                    Z, H, W, n_pix, n_reg))
                ws = _bucketed((ws_bytes,), torch.uint8, dev, 16 << 20)
                dist = vol.distances if vol.distances is not None \
                    else vol.magnitudes
                ddt = _lib.MF_DTYPE_F64 if dist.dtype == torch.float64 \
                    else _lib.MF_DTYPE_F32
                tr = vol.traces
                _lib.check(L.mf_region_table(
                    vol.decoded.data_ptr(), vol.magnitudes.data_ptr(),
                    dist.data_ptr(), ddt, tr.data_ptr(), tr.stride(0),
                    tr.stride(1), tr.stride(3), labels.data_ptr(), B,
                    self._barcodeCount if vol.codes_in_codebook else 0,
                    Z, H, W, int(is3D),
                    n_pix, n_reg, rec.data_ptr(), inten.data_ptr(),
                    _ptr(mean_tm), _ptr(min_tm), ws.data_ptr(), ws_bytes, sp))
        return RegionTable(rec, inten, mean_tm, min_tm)

    def region_table(self, vol: DecodedVolume, is3D: bool = False,
                     withRefactorStats: bool = False,
                     pendingLabels=None) -> RegionTable:
        """CCL + region aggregation of every barcode at once on the GPU
        (replaces the per-index label/regionprops loop, decoding.py:165-221,
        323-334, 359-369).  Cached on the DecodedVolume."""
        key = (bool(is3D), bool(withRefactorStats))
        if key in vol._tables:
            return vol._tables[key]
        if (bool(is3D), True) in vol._tables:
            return vol._tables[(bool(is3D), True)]
        if vol.traces is None:
            raise ValueError('region extraction needs the pixel traces')
        if pendingLabels is None:
            pendingLabels = self._label_enqueue(vol, is3D)
        table = self._table_enqueue(vol, pendingLabels, is3D,
                                    withRefactorStats)
        vol._tables[key] = table
        vol.labels = pendingLabels[0]
        return table

    # ------------------------------------------ extract_barcodes_with_index --

    def _frame_from_rows(self, table: RegionTable, rows, is3D, bitCount,
                         fov, cropWidth, zIndex, globalAligner, minimumArea,
                         cropShape, zFromTable=False) -> pandas.DataFrame:
        """Assemble the reference's DataFrame (decoding.py:171-252) for the
        given region rows (one barcode index, or every index in reference
        order).  With zFromTable the z column is each row's own plane (a
        multi-plane 2-D volume), else zIndex."""
        intensityColumns = ['intensity_{}'.format(i) for i in range(bitCount)]
        rec = table.rec[rows]
        n = rec.shape[0]
        if n == 0:
            return pandas.DataFrame(columns=COLUMN_NAMES + intensityColumns)

        centroids = np.empty((n, 3))
        # reference order: [z, x (column), y (row)] (decoding.py:188, 209-211)
        if is3D:
            centroids[:, 0] = rec[:, 7]
        elif zFromTable:
            centroids[:, 0] = rec[:, 1]
        else:
            centroids[:, 0] = zIndex if zIndex is not None else np.nan
        centroids[:, 1] = rec[:, 9]
        centroids[:, 2] = rec[:, 8]
        if globalAligner is not None:
            globalCentroids = np.asarray(
                globalAligner.fov_coordinate_array_to_global(fov, centroids))
        else:
            globalCentroids = centroids

        # exclusive crop on the centroid, quirks included: x is compared with
        # shape[0] and y with shape[1] of the decoded image (decoding.py:244)
        x, y, area = centroids[:, 1], centroids[:, 2], rec[:, 2]
        keep = ((x > cropWidth) & (x < cropShape[0] - cropWidth)
                & (y > cropWidth) & (y < cropShape[1] - cropWidth)
                & (area >= minimumArea))
        # index = position inside the per-(plane, barcode) frame the reference
        # builds before concatenating (decode.py:225-230)
        gkey = rec[:, 1] * 65536.0 + rec[:, 0] if not is3D else rec[:, 0]
        starts = np.flatnonzero(np.r_[True, gkey[1:] != gkey[:-1]])
        pos = np.arange(n) - np.repeat(starts, np.diff(np.r_[starts, n]))
        sel = np.flatnonzero(keep)
        cols = {
            'barcode_id': rec[sel, 0].astype(np.int64),
            'fov': np.full(len(sel), fov),
            'mean_intensity': rec[sel, 3],
            'max_intensity': rec[sel, 4],
            'area': rec[sel, 2],
            'mean_distance': rec[sel, 5],
            'min_distance': rec[sel, 6],
            'x': centroids[sel, 1],
            'y': centroids[sel, 2],
            'z': centroids[sel, 0],
            'global_x': globalCentroids[sel, 1],
            'global_y': globalCentroids[sel, 2],
            'global_z': globalCentroids[sel, 0],
            'cell_index': np.full(len(sel), -1, dtype=np.int64),
        }
        inten = table.inten[rows][sel]
        for i, c in enumerate(intensityColumns):
            cols[c] = inten[:, i]
        return pandas.DataFrame(cols, index=pos[sel], copy=False)

    @staticmethod
    def _pinned(shape, dtype):
        """Fresh page-locked staging tensor from torch's caching host
        allocator (no hipHostMalloc after warm-up).  The numpy views handed to
        pandas keep it alive, so the DataFrame owns its memory without a
        second copy."""
        return torch.empty(tuple(shape), dtype=dtype, pin_memory=True)

    def _frame_enqueue(self, table: RegionTable, is3D, bitCount, cropWidth,
                       minimumArea, cropShape, stream=None, typed=False,
                       wantXY64=False):
        """Launch mf_region_frame for every row of a device-resident table:
        the crop / area filter and the column gather run on the GPU, which
        writes only the kept rows, column-major; three asynchronous copies
        bring them into page-locked buffers.
        Returns the pending state for _frame_finish.  With ``stream`` the work
        runs on that side stream (after what the current stream has queued),
        so it overlaps whatever the caller launches next.  With ``typed`` the
        rows come in the barcode database's column types
        (mf_region_frame_typed) for _frame_finish_typed; ``wantXY64`` adds the
        unrounded x, y a global aligner needs."""
        rec = table.rec_d
        n = rec.shape[0]
        if n == 0:
            return None
        dev = rec.device
        L = _lib.lib()
        h32 = self._pinned((bitCount, n), torch.float32)
        hcount = self._pinned((1,), torch.int64)
        if typed:
            hosts = [self._pinned((2, n), torch.uint16),
                     self._pinned((_lib.MF_TYPED_F32_COLS, n), torch.float32),
                     self._pinned((n,), torch.int32),
                     self._pinned((2, n), torch.float64) if wantXY64
                     else None]
        else:
            hosts = [self._pinned((_lib.MF_FRAME_COLS, n), torch.float64)]
        with torch.cuda.device(dev):
            cur = torch.cuda.current_stream(dev)
            run = cur if stream is None else stream
            if run is not cur:
                run.wait_stream(cur)
            with torch.cuda.stream(run):
                ws_bytes = _lib.check(L.mf_region_frame_workspace_bytes(n))
                ws = _bucketed((ws_bytes,), torch.uint8, dev)
                if _FRAME_VIA_COPY:
                    # rows into device buffers, then the copy engine moves
                    # them: no workgroup sits on PCIe beside the next decode
                    outs = [None if h is None
                            else _bucketed(h.shape, h.dtype, dev)
                            for h in hosts]
                    o32 = _bucketed(h32.shape, h32.dtype, dev)
                    ocount = torch.empty_like(hcount, device=dev)
                else:
                    outs, o32, ocount = hosts, h32, hcount
                common = (rec.data_ptr(), table.inten_d.data_ptr(), n,
                          bitCount, int(is3D), float(cropWidth),
                          float(cropShape[0]), float(cropShape[1]),
                          float(minimumArea))
                if typed:
                    _lib.check(L.mf_region_frame_typed(
                        *common, outs[0].data_ptr(), outs[1].data_ptr(),
                        outs[2].data_ptr(), o32.data_ptr(), _ptr(outs[3]),
                        ocount.data_ptr(), ws.data_ptr(), ws_bytes,
                        run.cuda_stream))
                else:
                    _lib.check(L.mf_region_frame(
                        *common, outs[0].data_ptr(), o32.data_ptr(),
                        ocount.data_ptr(), ws.data_ptr(), ws_bytes,
                        run.cuda_stream))
                if _FRAME_VIA_COPY:
                    hcount.copy_(ocount, non_blocking=True)
                    for h, o in zip(hosts + [h32], outs + [o32]):
                        if h is not None:
                            h.copy_(o, non_blocking=True)
                            o.record_stream(run)
                    ocount.record_stream(run)
                done = torch.cuda.Event()
                done.record(run)
        # `table` and `ws` stay referenced until the event has fired
        if typed:
            return (table, ws, hosts, h32, hcount, done)
        return (table, ws, hosts[0], h32, hcount, done)

    def _frame_finish_typed(self, pending, bitCount, fov, globalAligner):
        """Wait for a typed _frame_enqueue and wrap its columns as
        ``barcodedb.TypedBarcodes``: what ``BarcodeDB.write_barcodes`` would
        make of the reference's DataFrame (its astype to the column types,
        barcodedb.py:264-266), without the float64 frame in between.  fov and
        cell_index are constants; the global columns are the local ones unless
        a global aligner maps the unrounded centroids."""
        from merlin_amd.barcodedb import TypedBarcodes
        names = ['intensity_{}'.format(i) for i in range(bitCount)]
        if pending is None:
            m = 0
            u16 = np.zeros((2, 0), np.uint16)
            f32 = np.zeros((_lib.MF_TYPED_F32_COLS, 0), np.float32)
            idx = np.zeros(0, np.int32)
            a32 = np.zeros((bitCount, 0), np.float32)
            xy = None
        else:
            _, _, hosts, h32, hcount, done = pending
            done.synchronize()
            m = int(hcount[0])
            u16 = hosts[0].numpy()[:, :m]
            f32 = hosts[1].numpy()[:, :m]
            idx = hosts[2].numpy()[:m]
            xy = None if hosts[3] is None else hosts[3].numpy()[:, :m]
            a32 = h32.numpy()[:, :m]
        gx, gy, gz = f32[4], f32[5], f32[6]
        if globalAligner is not None and m:
            if xy is None:
                raise ValueError('a global aligner needs wantXY64')
            centroids = np.stack([f32[6].astype(np.float64), xy[0], xy[1]],
                                 axis=1)
            g = np.asarray(globalAligner.fov_coordinate_array_to_global(
                fov, centroids))
            gx, gy, gz = (g[:, 1].astype(np.float32),
                          g[:, 2].astype(np.float32),
                          g[:, 0].astype(np.float32))
        cols = {
            'barcode_id': u16[0],
            'fov': np.full(m, fov).astype(np.uint16),
            'mean_intensity': f32[0], 'max_intensity': f32[1],
            'area': u16[1], 'mean_distance': f32[2], 'min_distance': f32[3],
            'x': f32[4], 'y': f32[5], 'z': f32[6],
            'global_x': gx, 'global_y': gy, 'global_z': gz,
            'cell_index': np.full(m, -1, dtype=np.int32),
        }
        for i, c in enumerate(names):
            cols[c] = a32[i]
        return TypedBarcodes(cols, idx.astype(np.int64))

    def _frame_finish(self, pending, bitCount, fov, globalAligner,
                      zOverride=None) -> pandas.DataFrame:
        """Wait for _frame_enqueue's rows and hand them to pandas without a
        copy.  globalAligner is applied to the kept rows (it is a per-row
        affine map, globalalign.py:141-148)."""
        intensityColumns = ['intensity_{}'.format(i) for i in range(bitCount)]
        if pending is None:
            return pandas.DataFrame(columns=COLUMN_NAMES + intensityColumns)
        _, _, h64, h32, hcount, done = pending
        done.synchronize()
        m = int(hcount[0])
        a64 = h64.numpy()[:, :m]
        a32 = h32.numpy()[:, :m]
        xs, ys, zs = a64[6], a64[7], a64[8]
        if zOverride is not None:
            # a single plane decoded on its own carries the caller's zIndex
            # (decoding.py:209-211)
            zs = np.full(m, float(zOverride))
        if globalAligner is not None:
            centroids = np.stack([zs, xs, ys], axis=1)
            g = np.asarray(globalAligner.fov_coordinate_array_to_global(
                fov, centroids))
            gx, gy, gz = g[:, 1], g[:, 2], g[:, 0]
        else:
            gx, gy, gz = xs, ys, zs
        cols = {
            'barcode_id': a64[0].astype(np.int64),
            'fov': np.full(m, fov),
            'mean_intensity': a64[1], 'max_intensity': a64[2],
            'area': a64[3], 'mean_distance': a64[4], 'min_distance': a64[5],
            'x': xs, 'y': ys, 'z': zs,
            'global_x': gx, 'global_y': gy, 'global_z': gz,
            'cell_index': np.full(m, -1, dtype=np.int64),
        }
        for i, c in enumerate(intensityColumns):
            cols[c] = a32[i]
        return pandas.DataFrame(cols, index=a64[9].astype(np.int64),
                                copy=False)

    def _frame_on_device(self, table: RegionTable, is3D, bitCount, fov,
                         cropWidth, globalAligner, minimumArea, cropShape,
                         zOverride=None) -> pandas.DataFrame:
        """_frame_from_rows for every row of a device-resident table."""
        pending = self._frame_enqueue(table, is3D, bitCount, cropWidth,
                                      minimumArea, cropShape)
        return self._frame_finish(pending, bitCount, fov, globalAligner,
                                  zOverride)

    def extract_barcodes_with_index(
            self, barcodeIndex: int, decodedImage: np.ndarray,
            pixelMagnitudes: np.ndarray, pixelTraces: np.ndarray,
            distances: np.ndarray, fov: int, cropWidth: int,
            zIndex: int = None, globalAligner=None, minimumArea: int = 0
    ) -> pandas.DataFrame:
        """Same contract as decoding.py:133-252.  The first call for a set of
        images labels and aggregates every barcode on the GPU; later calls
        with the same arrays slice the cached table."""
        is3D = len(pixelTraces.shape) == 4
        vol = self._lookup(decodedImage, pixelMagnitudes, pixelTraces,
                           distances)
        table = self.region_table(vol, is3D)
        ids = table.rec[:, 0]
        rows = np.nonzero(ids == barcodeIndex)[0]
        bitCount = pixelTraces.shape[1] if is3D else pixelTraces.shape[0]
        return self._frame_from_rows(
            table, rows, is3D, bitCount, fov, cropWidth, zIndex,
            globalAligner, minimumArea, decodedImage.shape)

    def extract_barcodes(self, vol: DecodedVolume, fov: int, cropWidth: int,
                         zIndex=None, globalAligner=None,
                         minimumArea: int = 0, is3D: bool = False,
                         zPlane: Optional[int] = None) -> pandas.DataFrame:
        """All barcode indexes at once = pandas.concat over
        extract_barcodes_with_index(i, ...) for i in range(N)
        (analysis/decode.py:225-230), from a device-resident volume.
        In 2-D mode ``zPlane`` selects one plane of a multi-plane volume."""
        table = self.region_table(vol, is3D)
        Z, H, W = vol.shape
        cropShape = (Z, H, W) if is3D else (H, W)
        if is3D or (zPlane is None and zIndex is None and Z > 1):
            # every row; in 2-D mode the z column is each row's own plane
            # (a multi-plane 2-D volume has no counterpart in the reference)
            return self._frame_on_device(
                table, is3D, vol.bit_count, fov, cropWidth, globalAligner,
                minimumArea, cropShape)
        if Z == 1 and zPlane in (None, 0) and len(table):
            if zIndex is None:
                # the reference assigns None into the float centroid column:
                # NaN (decoding.py:209-211)
                zIndex = float('nan')
            # one plane decoded on its own (OptimizeIteration's fragments):
            # every row, z = the caller's zIndex
            return self._frame_on_device(
                table, is3D, vol.bit_count, fov, cropWidth, globalAligner,
                minimumArea, cropShape, zOverride=zIndex)
        rows = np.arange(len(table)) if zPlane is None \
            else np.nonzero(table.rec[:, 1] == zPlane)[0]
        return self._frame_from_rows(
            table, rows, is3D, vol.bit_count, fov, cropWidth, zIndex,
            globalAligner, minimumArea, cropShape)

    # ---------------------------------------------------- extract_refactors --

    def _refactor_sums(self, table: RegionTable, stat, Z, plane,
                       areaThreshold, offset, normalize):
        """mf_refactor_sums: (sums float64 [N,B], seen float64 [N]) on the
        host; one small device-to-host copy."""
        import ctypes
        L = _lib.lib()
        N, B = self._barcodeCount, self._bitCount
        dev = table.rec_d.device
        with torch.cuda.device(dev):
            out = torch.empty(N * B + N, dtype=torch.float64, device=dev)
            off = None
            if offset is not None:
                off = np.ascontiguousarray(offset, dtype=np.float64)
            _lib.check(L.mf_refactor_sums(
                table.rec_d.data_ptr(), stat.data_ptr(), len(table), B, N,
                int(Z), -1 if plane is None else int(plane),
                float(areaThreshold),
                None if off is None
                else off.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                int(normalize), out.data_ptr(), out[N * B:].data_ptr(),
                _stream_ptr(dev)))
            host = out.cpu().numpy()
        return host[:N * B].reshape(N, B), host[N * B:]

    def _refactors_from_table(self, table: RegionTable, Z, plane,
                              extractBackgrounds: bool):
        """extract_refactors' reductions (decoding.py:290-377) from the region
        table: the per-barcode sums over regions on the GPU
        (mf_refactor_sums), the [N,B] arithmetic that follows here."""
        N, B = self._barcodeCount, self._bitCount
        import warnings
        if len(table) == 0:
            seen = np.zeros(N)
            sumPixelTraces = np.zeros((N, B))
            sumMin = np.zeros((N, B))
            seenBg = np.zeros(N)
        if extractBackgrounds:
            # decoding.py:342-377, area threshold 5 hard-coded at :362
            if len(table):
                sumMin, seenBg = self._refactor_sums(
                    table, table.min_tm_d, Z, plane, 5, None, False)
            off = sumMin.copy()
            off[self._decodingMatrix > 0] = np.nan
            with np.errstate(invalid='ignore', divide='ignore'), \
                    warnings.catch_warnings():
                warnings.simplefilter('ignore', category=RuntimeWarning)
                backgroundRefactors = np.nansum(off, axis=0) / np.sum(
                    (self._decodingMatrix == 0) * seenBg[:, np.newaxis],
                    axis=0)
        else:
            backgroundRefactors = np.zeros(B)
        if len(table):
            sumPixelTraces, seen = self._refactor_sums(
                table, table.mean_tm_d, Z, plane, self.refactorAreaThreshold,
                backgroundRefactors, True)
            sumPixelTraces = sumPixelTraces.copy()
        barcodesSeen = np.array(seen, dtype=np.float64)
        sumPixelTraces[self._decodingMatrix == 0] = np.nan
        with warnings.catch_warnings(), np.errstate(invalid='ignore',
                                                    divide='ignore'):
            warnings.simplefilter('ignore', category=RuntimeWarning)
            onBitIntensity = np.nanmean(sumPixelTraces, axis=0)
            refactors = onBitIntensity / np.mean(onBitIntensity)
        return refactors, backgroundRefactors, barcodesSeen

    def extract_refactors(
            self, decodedImage, pixelMagnitudes, normalizedPixelTraces,
            extractBackgrounds=False
    ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """Same contract as decoding.py:290-340: (refactors[B],
        backgroundRefactors[B], barcodesSeen[N])."""
        vol = self._lookup(decodedImage, pixelMagnitudes,
                           normalizedPixelTraces, None)
        table = self.region_table(vol, False, True)
        return self._refactors_from_table(table, vol.shape[0], None,
                                          extractBackgrounds)

    def extract_refactors_device(self, vol: DecodedVolume,
                                 extractBackgrounds=False, zPlane=None):
        """extract_refactors on a device-resident volume; zPlane selects one
        plane of a multi-plane volume (None: every region of the volume)."""
        table = self.region_table(vol, False, True)
        return self._refactors_from_table(table, vol.shape[0], zPlane,
                                          extractBackgrounds)
