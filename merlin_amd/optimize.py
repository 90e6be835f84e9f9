"""Scale-factor / background optimisation loop: the MI355X counterpart of
merlin/analysis/optimize.py::OptimizeIteration (optimize.py:15-416, chromatic
correction excluded: default off, optimize.py:31-32).

In the reference every fragment (a random (fov, z) draw) runs in its own
process and the cross-fragment reduction goes through ``.npy`` files on a
shared file system (writers optimize.py:123-131, readers :328-340, :358-373,
:406-408).  Here fragments are sharded over the ranks of a
``torch.distributed`` group (one rank per GPU, RCCL over xGMI) and the one
exchange step per iteration is a sum-all-reduce of zero-initialised ``[F, .]``
matrices in which every rank fills only its own rows -- bit-exact because
``x + 0 == x`` (NaN rows stay NaN) -- followed by the reference's nanmedian on
every rank.  Parameter names and defaults are the task's (optimize.py:25-50).
"""
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

from merlin_amd.decoding import PixelBasedDecoder


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def shard_fragments(count: int, rank: int, world: int) -> List[int]:
    """Fragment i belongs to rank i mod world (fragments are independent:
    optimize.py:65-66, 78)."""
    return list(range(rank, count, world))


def all_gather_rows(local: np.ndarray, rows: Sequence[int], total: int,
                    device=None, group=None) -> np.ndarray:
    """Every rank contributes ``local[k]`` as row ``rows[k]`` of a [total, C]
    float64 matrix; returns the full matrix on every rank.  ONE sum-all-reduce
    over a zero-initialised buffer (exact: x + 0 == x, NaN rows stay NaN).  On
    the nccl (RCCL) backend the buffer is built and reduced on the GPU and
    crosses PCIe once each way; gloo reduces the host buffer."""
    local = np.asarray(local, dtype=np.float64)
    cols = local.shape[1:] if local.ndim > 1 else ()
    full = np.zeros((total,) + tuple(cols), dtype=np.float64)
    if len(rows):
        full[list(rows)] = local
    dist = _dist()
    if dist is None or dist.get_world_size(group) == 1:
        return full
    t = torch.from_numpy(full)
    if dist.get_backend(group) == 'nccl':
        t = t.pin_memory().to(
            device if device is not None
            else torch.device('cuda', torch.cuda.current_device()),
            non_blocking=True)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


def exchange_fragment_results(refactors: np.ndarray, bgRefactors: np.ndarray,
                              counts: np.ndarray, rows: Sequence[int],
                              total: int, device=None, group=None):
    """The one exchange step of an iteration (optimize.py:328-345, 358-373,
    406-408 read every fragment's files): this rank's scale refactors [n,B],
    background refactors [n,B] and barcode counts [n,N] packed side by side
    into one [total, 2B+N] matrix -> a single all-reduce -> the three full
    matrices on every rank."""
    B, N = refactors.shape[1], counts.shape[1]
    packed = np.concatenate([refactors, bgRefactors, counts], axis=1) \
        if len(rows) else np.zeros((0, 2 * B + N))
    full = all_gather_rows(packed, rows, total, device, group)
    return full[:, :B], full[:, B:2 * B], full[:, 2 * B:]


def broadcast_from_rank0(obj, group=None):
    """``obj`` of rank 0 on every rank (tiny Python objects: the fragment
    draw)."""
    dist = _dist()
    if dist is None or dist.get_world_size(group) == 1:
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0)
                               if group is not None else 0, group=group)
    return box[0]


def pixel_histogram(stack: torch.Tensor) -> torch.Tensor:
    """Per-bit histogram of a uint16 stack [B, ...] with the reference's bins
    (np.arange(0, 65535): 65534 bins, the last one closed; preprocess.py:
    110-131).  Returns float64 [B, 65534] on the stack's device."""
    B = stack.shape[0]
    out = torch.zeros((B, 65534), dtype=torch.float64, device=stack.device)
    for b in range(B):
        h = torch.bincount(stack[b].reshape(-1).to(torch.int32),
                           minlength=65536)
        out[b] = h[:65534].double()
        out[b, 65533] += h[65534].double()     # np.histogram's closed last bin
    return out


def initial_scale_factors(pixelHistograms: np.ndarray) -> np.ndarray:
    """optimize.py:140-155: argmin |cdf - 0.9| + 2 per bit."""
    h = np.asarray(pixelHistograms, dtype=np.float64)
    out = np.zeros(h.shape[0])
    for i in range(h.shape[0]):
        c = np.cumsum(h[i])
        c = c / c[-1]
        out[i] = np.argmin(np.abs(c - 0.9)) + 2
    return out


def summed_pixel_histogram(local: torch.Tensor, group=None) -> np.ndarray:
    """preprocess.py:23-33 (sum of the per-FOV histograms) across ranks: a
    genuine sum-all-reduce of a [B, 65534] float64 matrix."""
    dist = _dist()
    t = local.clone()
    if dist is not None and dist.get_world_size(group) > 1:
        if dist.get_backend(group) != 'nccl':
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


_WORKER_STREAMS = {}


def _worker_stream(dev, slot):
    """HIP stream of fragment worker ``slot`` on ``dev``, created once per
    process."""
    import torch
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (index, slot)
    if key not in _WORKER_STREAMS:
        with torch.cuda.device(dev):
            _WORKER_STREAMS[key] = torch.cuda.Stream(dev)
    return _WORKER_STREAMS[key]


class OptimizeIteration(object):
    """One iteration of the scale-factor optimisation.

    ``fragmentFunction(decoder, fovIndex, zIndex, scaleFactors, backgrounds)``
    returns ``(refactors[B], backgroundRefactors[B], barcodesSeen[N])`` for one
    fragment; the default decodes ``imageProvider(fov, z)`` (a uint16 [B,H,W]
    tensor) on the GPU exactly as optimize.py:100-112 does."""

    def __init__(self, codebook, parameters=None, fovs: Sequence[int] = (0,),
                 zCount: int = 1, previousIteration=None,
                 initialScaleFactors: Optional[np.ndarray] = None,
                 imageProvider: Optional[Callable] = None,
                 fragmentFunction: Optional[Callable] = None,
                 device=None, group=None, seed: Optional[int] = None):
        self.parameters = dict(parameters or {})
        p = self.parameters
        p.setdefault('fov_per_iteration', 50)
        p.setdefault('area_threshold', 5)
        p.setdefault('optimize_background', False)
        p.setdefault('optimize_chromatic_correction', False)
        p.setdefault('crop_width', 0)
        # fragments in flight on one GPU (worker threads, one stream each); not
        # a parameter of the reference, which runs one process per fragment
        # two fragments in flight: with the per-barcode sums on the device a fragment is
        # mostly GPU latency, and main + 2 worker streams stay within HIP's four
        # hardware queues (measured 1.2 ms per fragment; 3: 1.2, 4: 1.5, 6: 1.7)
        p.setdefault('fragment_workers', 2)
        if p['optimize_chromatic_correction']:
            raise NotImplementedError('chromatic correction is out of scope')
        if 'fov_index' in p:
            p['fov_per_iteration'] = len(p['fov_index'])
        else:
            # optimize.py:43-50.  Every rank must work on the same draw: rank 0
            # draws (from OS entropy when seed is None, like the reference) and
            # the list is broadcast.
            # (seed may be a RandomState shared by a chain of iterations: the
            # reference draws every iteration from numpy's one global stream)
            rng = seed if isinstance(seed, np.random.RandomState) \
                else np.random.RandomState(seed)
            p['fov_index'] = broadcast_from_rank0([
                [int(rng.choice(list(fovs))), int(rng.choice(list(range(zCount))))]
                for _ in range(p['fov_per_iteration'])], group)
        self._codebook = codebook
        self._previous = previousIteration
        self._initial = None if initialScaleFactors is None \
            else np.asarray(initialScaleFactors, dtype=np.float64)
        self._provider = imageProvider
        self._fragment_fn = fragmentFunction or self._decode_fragment
        self._device = device
        self._group = group
        self._bits = int(np.asarray(codebook.get_barcodes()).shape[1])
        self._codes = int(np.asarray(codebook.get_barcodes()).shape[0])
        self._complete = False
        self._decoder = None
        self.barcodes = {}        # fragment -> DataFrame (this rank's only)

    # ---------------------------------------------------------- helpers --

    def fragment_count(self):
        return self.parameters['fov_per_iteration']

    def get_codebook(self):
        return self._codebook

    def _get_previous_scale_factors(self) -> np.ndarray:
        """optimize.py:157-165"""
        if self._previous is None:
            if self._initial is None:
                raise ValueError('the first iteration needs '
                                 'initialScaleFactors (optimize.py:140-155)')
            return self._initial.copy()
        return self._previous.get_scale_factors()

    def _get_previous_backgrounds(self) -> np.ndarray:
        """optimize.py:167-175"""
        if self._previous is None:
            return np.zeros(self._bits)
        return self._previous.get_backgrounds()

    def _decode_fragment(self, decoder, fovIndex, zIndex, scaleFactors,
                         backgrounds):
        """optimize.py:100-122 for one fragment on the GPU."""
        if self._provider is None:
            raise ValueError('imageProvider is required')
        img = self._provider(fovIndex, zIndex)
        out, vol = self._fragment_on_device(decoder, img, scaleFactors,
                                            backgrounds)
        self._last_volume = vol
        return out

    def _fragment_on_device(self, decoder, img, scaleFactors, backgrounds):
        # region statistics read the traces of decoded pixels only
        vol = decoder.decode_pixels_device(img, scaleFactors, backgrounds,
                                           returnTraces='decoded')
        out = decoder.extract_refactors_device(
            vol, self.parameters['optimize_background'])
        return out, vol

    def _fragment_barcodes(self, vol, fovIndex, zIndex):
        """optimize.py:116-122: the fragment's barcodes"""
        p = self.parameters
        return self._decoder.extract_barcodes(
            vol, fovIndex, p['crop_width'], zIndex, None, p['area_threshold'],
            zPlane=0)

    def _run_fragments_threaded(self, mine, scaleFactors, backgrounds, workers,
                                rf, bgr, seen):
        """The default fragment function with ``workers`` fragments in flight.
        A fragment is < 1 ms of GPU work and ~4 ms of host work (the ordered
        float64 reductions of extract_refactors and the DataFrame), most of it
        inside numpy / torch calls that release the interpreter lock, so
        worker threads -- each with its own HIP stream -- overlap one
        fragment's host work with the others' kernels and copies.  Fragments are
        independent and every reduction inside one is ordered, so the results
        do not depend on the scheduling.  The image provider is called from the
        calling thread, in fragment order; a worker's stream waits for whatever
        the provider queued on the caller's stream."""
        import threading
        from concurrent.futures import ThreadPoolExecutor
        import torch
        p = self.parameters
        dev = self._decoder._dev()
        local = threading.local()

        slots = iter(range(workers))
        slot_lock = threading.Lock()

        def work(k, frag, img, ready):
            if not hasattr(local, 'stream'):
                # worker i of every run uses the same stream: new streams per
                # run would pile up on the four hardware queues HIP maps them to
                with slot_lock:
                    local.stream = _worker_stream(dev, next(slots))
            fovIndex, zIndex = p['fov_index'][frag]
            with torch.cuda.device(dev), torch.cuda.stream(local.stream):
                local.stream.wait_event(ready)
                out, vol = self._fragment_on_device(
                    self._decoder, img, scaleFactors, backgrounds)
                barcodes = self._fragment_barcodes(vol, fovIndex, zIndex)
                # the volume's buffers were allocated on this stream; it is
                # idle again before they go back to the allocator
                local.stream.synchronize()
            return k, frag, out, barcodes

        with ThreadPoolExecutor(max_workers=workers) as pool:
            pending = []
            for k, frag in enumerate(mine):
                fovIndex, zIndex = p['fov_index'][frag]
                img = self._provider(fovIndex, zIndex)
                with torch.cuda.device(dev):
                    ready = torch.cuda.Event()
                    ready.record(torch.cuda.current_stream(dev))
                pending.append(pool.submit(work, k, frag, img, ready))
                # bound the images kept alive by fragments not yet started
                while len(pending) > 2 * workers:
                    self._store_fragment(pending.pop(0).result(), rf, bgr, seen)
            for f in pending:
                self._store_fragment(f.result(), rf, bgr, seen)

    def _store_fragment(self, result, rf, bgr, seen):
        k, frag, out, barcodes = result
        rf[k], bgr[k], seen[k] = out
        self.barcodes[frag] = barcodes

    # -------------------------------------------------------------- run --

    def run(self):
        """Run this rank's fragments, then the exchange step."""
        p = self.parameters
        dist = _dist()
        rank = dist.get_rank(self._group) if dist else 0
        world = dist.get_world_size(self._group) if dist else 1
        F = p['fov_per_iteration']
        mine = shard_fragments(F, rank, world)

        scaleFactors = self._get_previous_scale_factors()
        backgrounds = self._get_previous_backgrounds()
        if self._decoder is None and self._fragment_fn == self._decode_fragment:
            self._decoder = PixelBasedDecoder(self._codebook,
                                              device=self._device)
        if self._decoder is not None:
            self._decoder.refactorAreaThreshold = p['area_threshold']

        rf = np.zeros((len(mine), self._bits))
        bgr = np.zeros((len(mine), self._bits))
        seen = np.zeros((len(mine), self._codes))
        workers = int(p['fragment_workers'])
        if self._fragment_fn == self._decode_fragment and workers > 1 \
                and len(mine) > 1:
            if self._provider is None:
                raise ValueError('imageProvider is required')
            self._decoder._codebook_device()   # built before the threads start
            self._run_fragments_threaded(mine, scaleFactors, backgrounds,
                                         workers, rf, bgr, seen)
            mine_serial = []
        else:
            mine_serial = list(enumerate(mine))
        for k, frag in mine_serial:
            fovIndex, zIndex = p['fov_index'][frag]
            self._last_volume = None
            rf[k], bgr[k], seen[k] = self._fragment_fn(
                self._decoder, fovIndex, zIndex, scaleFactors, backgrounds)
            if self._last_volume is not None and self._decoder is not None:
                self.barcodes[frag] = self._fragment_barcodes(
                    self._last_volume, fovIndex, zIndex)
                self._last_volume = None

        # one exchange per iteration (every rank used the same previous
        # factors, so previous_scale_factors[F,B] is a broadcast row)
        self._refactors, self._bg_refactors, self._counts = \
            exchange_fragment_results(rf, bgr, seen, mine, F, self._device,
                                      self._group)
        self._previous_factors = np.tile(scaleFactors, (F, 1))
        self._previous_backgrounds = np.tile(backgrounds, (F, 1))
        self._complete = True
        return self

    # ---------------------------------------------------------- results --

    def is_complete(self):
        return self._complete

    def get_scale_factors(self) -> np.ndarray:
        """optimize.py:311-345"""
        if not self._complete:
            raise Exception('Analysis is still running. Unable to get scale '
                            + 'factors.')
        refactors = self._refactors.copy()
        refactors[refactors == 0] = 1       # bits that were never seen
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', category=RuntimeWarning)
            return np.nanmedian(
                np.multiply(refactors, self._previous_factors), axis=0)

    def get_backgrounds(self) -> np.ndarray:
        """optimize.py:347-378"""
        if not self._complete:
            raise Exception('Analysis is still running. Unable to get '
                            + 'backgrounds.')
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', category=RuntimeWarning)
            return np.nanmedian(np.add(
                self._previous_backgrounds,
                np.multiply(self._bg_refactors, self._previous_factors)),
                axis=0)

    def get_scale_factor_history(self) -> np.ndarray:
        """optimize.py:380-395"""
        if self._previous is None:
            return np.array([self.get_scale_factors()])
        return np.append(self._previous.get_scale_factor_history(),
                         [self.get_scale_factors()], axis=0)

    def get_barcode_count_history(self) -> np.ndarray:
        """optimize.py:397-416"""
        countsMean = np.mean(self._counts, axis=0)
        if self._previous is None:
            return np.array([countsMean])
        return np.append(self._previous.get_barcode_count_history(),
                         [countsMean], axis=0)


def run_optimize_chain(codebook, imageProvider, fovs, zCount,
                       initialScaleFactors, iterations=3, parameters=None,
                       device=None, group=None, seed=None):
    """``iterations`` chained OptimizeIteration passes, the way an analysis
    JSON chains them through ``previous_iteration`` (optimize.py:61-62,
    158-165; the reference's test run chains two,
    test/auxiliary_files/test_analysis_parameters.json:17-41): every pass draws
    its own ``fov_per_iteration`` random (fov, z) fragments (optimize.py:43-50)
    from ONE random stream, as the reference's constructors draw from numpy's
    global one (``seed``: what ``np.random.seed`` would be given; None = OS
    entropy), shards them over the ranks of ``group`` and ends in the one
    exchange step.  Returns the list of
    completed iterations; the last one's get_scale_factors() /
    get_backgrounds() are what Decode uses (decode.py:107-110)."""
    done = []
    prev = None
    rng = np.random.RandomState(seed)
    for k in range(iterations):
        it = OptimizeIteration(
            codebook, parameters, fovs=fovs, zCount=zCount,
            previousIteration=prev,
            initialScaleFactors=initialScaleFactors if prev is None else None,
            imageProvider=imageProvider, device=device, group=group,
            seed=rng)
        it.run()
        done.append(it)
        prev = it
    return done
