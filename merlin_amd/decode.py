"""FOV-level decode runner: the MI355X counterpart of
merlin/analysis/decode.py::Decode._run_analysis (decode.py:92-175).

The reference task pulls images through MERlin's dataset/task framework (out
of scope); this runner takes the same inputs as tensors and returns
the same products: the barcode table the task writes to the barcode database
(decode.py:225-230) and the decoded / magnitude / distance images
(decode.py:113-127).  Parameter names and defaults are the task's
(decode.py:46-66)."""
import os

import numpy as np
import torch

from merlin_amd.decoding import PixelBasedDecoder, DecodedVolume


def shard_fovs(fovs, rank: int, world: int):
    """FOV number i of the list belongs to rank i mod world."""
    if not (0 <= rank < world):
        raise ValueError('rank {} outside 0..{}'.format(rank, world - 1))
    return [f for i, f in enumerate(fovs) if i % world == rank]


# (device index, role) -> torch.cuda.Stream, see Decode._shared_stream
_SIDE_STREAMS = {}


class Decode(object):
    """Parameters as analysis/decode.py:46-66.  What the reference's task gets
    from its data set is passed in instead:

    * ``database`` (run_fov / run_fovs): a ``merlin_amd.barcodedb.BarcodeDB``;
      the FOV's records then take the typed device path into it
      (``get_barcode_database().write_barcodes``, decode.py:225-230) and
      ``remove_z_duplicated_barcodes`` works on the database's table exactly
      as decode.py:170-175 does (read back, filter, empty, rewrite);
    * ``outputDirectory``: where ``write_decoded_images`` puts
      ``decoded_<fov>.tif`` (decode.py:165-168, 196-213; nothing is written
      without a directory);
    * ``memory_map`` (decode.py:131-139) asks the reference to keep the
      ``decode_3d`` pixel traces in a memory-mapped file instead of RAM.  Here
      those traces never reach host memory (they stay in HBM), so the
      parameter is accepted and changes nothing, results included."""

    def __init__(self, codebook, parameters=None, device=None,
                 outputDirectory=None, overlap=True):
        self.parameters = dict(parameters or {})
        p = self.parameters
        p.setdefault('crop_width', 100)
        p.setdefault('write_decoded_images', True)
        p.setdefault('minimum_area', 0)
        p.setdefault('distance_threshold', 0.5167)
        p.setdefault('lowpass_sigma', 1)
        p.setdefault('decode_3d', False)
        p.setdefault('memory_map', False)
        p.setdefault('remove_z_duplicated_barcodes', False)
        if p['remove_z_duplicated_barcodes']:
            p.setdefault('z_duplicate_zPlane_threshold', 1)
            p.setdefault('z_duplicate_xy_pixel_threshold', np.sqrt(2))
        self.cropWidth = p['crop_width']
        self._codebook = codebook
        self.decoder = PixelBasedDecoder(codebook, device=device)
        self.outputDirectory = outputDirectory
        # overlap=False: every stage of run_fovs on one stream (measurement)
        self.overlap = bool(overlap)
        # depth of the run_fovs pipeline (see there)
        self.decodeAhead = 1
        self.frameLag = 2
        # worker threads of the decoded-image output (imagewriter.AsyncImageWriter)
        self.imageWriters = 4

    def get_codebook(self):
        return self._codebook

    def _shared_stream(self, dev, role):
        """The side streams are per device and process, not per Decode object:
        HIP maps streams onto GPU_MAX_HW_QUEUES = 4 hardware queues, and the
        loop's four (main, labelling, table, rows) fill them.  With three new
        streams per Decode object the second object's streams share hardware
        queues with one another -- stages that should overlap then run one
        after the other: 1.75 ms per FOV instead of 1.46 (dev/streams.py)."""
        if not self.overlap:
            return None
        index = dev.index if dev.index is not None else torch.cuda.current_device()
        key = (index, role)
        if key not in _SIDE_STREAMS:
            with torch.cuda.device(dev):
                _SIDE_STREAMS[key] = torch.cuda.Stream(dev)
        return _SIDE_STREAMS[key]

    def _side_stream(self, dev):
        """Stream for mf_region_frame's PCIe-bound gather.  (A CU-masked
        stream was measured too: slower than an ordinary one.)"""
        return self._shared_stream(dev, 'rows')

    def _table_stream(self, dev):
        """Second compute stream for the region-table stage of the pipelined
        FOV loop."""
        return self._shared_stream(dev, 'table')

    def _label_stream(self, dev):
        """Third compute stream: the labelling of the pipelined FOV loop."""
        return self._shared_stream(dev, 'label')

    def reserve_memory(self, stackShape, fovsInFlight=None,
                       expectedBarcodes=None, typed=False):
        """Give torch's caching allocator the working set of the FOV loop up
        front: one block per stream of the pipeline (the allocator keeps a pool
        per stream), allocated and released at once, from which the per-FOV
        images, label volumes, tables and staging buffers are then carved.
        Without it the pool grows by a hipMalloc (5-8 ms) whenever a FOV needs a
        slightly larger table than any before it.  Optional; sizes are upper
        estimates for ``stackShape`` = (B, Z, H, W) and are capped at half the
        free memory.

        With ``expectedBarcodes`` (regions per FOV, an estimate) the page-locked
        blocks that carry the frame rows to the host are set up as well: a
        hipHostMalloc costs about 0.1 ms, but the first CPU access to each page
        of such a block faults (about 1 us a page here), which makes the
        DataFrame hand-over of the first twenty or so FOVs 1.3 ms slower each
        until every block of torch's pinned pool has been read once
        (dev/first_fovs.py).  The blocks are allocated in the shapes
        ``_frame_enqueue`` asks for (``typed``: those of the database path),
        read once and released to that pool."""
        B, Z, H, W = (int(v) for v in stackShape)
        px = Z * H * W
        n = fovsInFlight or (self.decodeAhead + self.frameLag + 2)
        dev = self.decoder._dev()
        sizes = [(None, px * (2 + 4 + 8 + 4 * B) * n),
                 (self._label_stream(dev), (px * 4 + (1 << 20)) * n),
                 (self._table_stream(dev), px * 24 * (self.frameLag + 1)),
                 (self._side_stream(dev), px * 4 * (self.frameLag + 1))]
        with torch.cuda.device(dev):
            free = torch.cuda.mem_get_info(dev)[0]
            scale = min(1.0, 0.5 * free / max(sum(b for _, b in sizes), 1))
            for stream, nbytes in sizes:
                nbytes = int(nbytes * scale)
                if stream is None:
                    if nbytes:
                        torch.empty(nbytes, dtype=torch.uint8, device=dev)
                elif nbytes:
                    with torch.cuda.stream(stream):
                        torch.empty(nbytes, dtype=torch.uint8, device=dev)
        if expectedBarcodes:
            from merlin_amd import _lib
            m = int(expectedBarcodes)
            if typed:
                shapes = [((2, m), torch.uint16),
                          ((_lib.MF_TYPED_F32_COLS, m), torch.float32),
                          ((m,), torch.int32), ((2, m), torch.float64)]
            else:
                shapes = [((_lib.MF_FRAME_COLS, m), torch.float64)]
            shapes.append(((B, m), torch.float32))
            # torch's pinned pool hands out blocks of 2^k bytes: ask for whole
            # blocks, twice the FOVs in flight (a block returns to the pool
            # only after the consumer has dropped the FOV's DataFrame)
            blocks = []
            for sh, dt in shapes:
                nbytes = int(np.prod(sh)) * torch.empty((), dtype=dt).element_size()
                nbytes = 1 << max(nbytes - 1, 1).bit_length()
                blocks += [self.decoder._pinned((nbytes,), torch.uint8)
                           for _ in range(2 * n)]
            for t in blocks:
                t[::4096].sum()     # one byte of every page
            del blocks

    def flush_images(self):
        """Wait until every decoded-image file queued so far is on disk."""
        w = getattr(self, '_image_writer', None)
        if w is not None:
            w.flush()

    def decode_fov_device(self, imageStack: torch.Tensor, scaleFactors,
                          backgrounds, returnTraces='decoded'
                          ) -> DecodedVolume:
        """Blur + normalise + nearest codeword for every z plane of one FOV
        in one launch.  imageStack: uint16 [B,Z,H,W] on the GPU.  By default
        the normalised traces are kept only at decoded pixels, pixel-major
        (all that barcode extraction reads); returnTraces=True keeps every
        pixel's trace in the reference's layout."""
        p = self.parameters
        return self.decoder.decode_pixels_device(
            imageStack, scaleFactors, backgrounds,
            distanceThreshold=p['distance_threshold'],
            lowPassSigma=p['lowpass_sigma'], returnTraces=returnTraces,
            distanceDtype=torch.float32 if p['decode_3d'] else torch.float64)

    def _finish_fov(self, pending, vol, fov, globalAligner, zPositions,
                    database=None):
        p = self.parameters
        zpos = zPositions if zPositions is not None \
            else list(range(vol.shape[0]))
        if database is not None:
            # decode.py:225-230 then :170-175, on the database's typed table
            database.write_barcodes(
                self.decoder._frame_finish_typed(pending, vol.bit_count, fov,
                                                 globalAligner), fov=fov)
            if p['remove_z_duplicated_barcodes']:
                from merlin_amd import barcodefilters
                bc = barcodefilters.remove_zplane_duplicates_all_barcodeids(
                    database.get_barcodes(fov=fov),
                    p['z_duplicate_zPlane_threshold'],
                    p['z_duplicate_xy_pixel_threshold'], zpos)
                database.empty_database(fov)
                database.write_barcodes(bc, fov=fov)
            return database.get_barcodes(fov=fov)
        bc = self.decoder._frame_finish(pending, vol.bit_count, fov,
                                        globalAligner)
        if p['remove_z_duplicated_barcodes']:
            # (the reference filters the database's float32 columns; without a
            # database the float64 frame is filtered as it is)
            from merlin_amd import barcodefilters
            bc = barcodefilters.remove_zplane_duplicates_all_barcodeids(
                bc, p['z_duplicate_zPlane_threshold'],
                p['z_duplicate_xy_pixel_threshold'], zpos)
        return bc

    def _queue_images(self, vol, fov):
        """decode.py:165-168, 196-213: decoded_<fov>.tif, written behind the
        FOV loop (imagewriter.AsyncImageWriter); complete after flush_images(),
        which run_fov and the end of run_fovs call."""
        if not self.parameters['write_decoded_images'] \
                or self.outputDirectory is None or fov is None:
            return
        from merlin_amd import imagewriter
        if getattr(self, '_image_writer', None) is None:
            os.makedirs(self.outputDirectory, exist_ok=True)
            self._image_writer = imagewriter.AsyncImageWriter(
                vol.decoded.device, workers=self.imageWriters)
        self._image_writer.submit(
            os.path.join(self.outputDirectory, 'decoded_{}.tif'.format(fov)),
            vol)

    def _enqueue_decode(self, imageStack, scaleFactors, backgrounds, lbl=None,
                        fov=None):
        """Stage 1 of a FOV: decode kernel + labelling (+ the image output of
        ``fov``), nothing blocks."""
        timing = getattr(self, 'kernel_events', None)
        if timing is not None:
            # measurement hook (bench.py): HIP events around the decode launch
            # on the stream it is queued on
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        vol = self.decode_fov_device(imageStack, scaleFactors, backgrounds)
        if timing is not None:
            e1.record()
            timing.append((e0, e1))
        self._queue_images(vol, fov)
        is3D = bool(self.parameters['decode_3d'])
        if lbl is None:
            lab = self.decoder._label_enqueue(vol, is3D)
        else:
            # the labelling on its own stream: the main one carries nothing
            # but decode kernels
            lbl.wait_stream(torch.cuda.current_stream(vol.decoded.device))
            with torch.cuda.stream(lbl):
                lab = self.decoder._label_enqueue(vol, is3D)
            vol.decoded.record_stream(lbl)
        return vol, lab

    def _enqueue_table(self, vol, lab, side=None, typed=False, wantXY64=False):
        """Stage 2: region table + frame rows.  Blocks once, on the pixel /
        region counts that size the tables."""
        p = self.parameters
        is3D = bool(p['decode_3d'])
        Z, H, W = vol.shape
        shape = (Z, H, W) if is3D else (H, W)
        tbl = self._table_stream(vol.decoded.device) if side is not None \
            else None
        if tbl is None:
            table = self.decoder.region_table(vol, is3D, pendingLabels=lab)
            return self.decoder._frame_enqueue(
                table, is3D, vol.bit_count, self.cropWidth,
                p['minimum_area'], shape, stream=side, typed=typed,
                wantXY64=wantXY64)
        # the sort / region statistics of this FOV on their own stream, beside
        # the decode kernel of the next FOV (already queued on the main one)
        tbl.wait_event(lab[3])
        with torch.cuda.stream(tbl):
            table = self.decoder.region_table(vol, is3D, pendingLabels=lab)
            pending = self.decoder._frame_enqueue(
                table, is3D, vol.bit_count, self.cropWidth,
                p['minimum_area'], shape, stream=side, typed=typed,
                wantXY64=wantXY64)
        for t in (vol.decoded, vol.magnitudes, vol.distances, vol.traces,
                  lab[0]):
            if t is not None:
                t.record_stream(tbl)
        return pending

    def run_fov(self, imageStack: torch.Tensor, scaleFactors, backgrounds,
                fov: int, globalAligner=None, zPositions=None, database=None):
        """decode.py:92-175 for one FOV.  Returns (barcodes DataFrame,
        DecodedVolume); the volume's decoded/magnitudes/distances tensors are
        the images _save_decoded_images would write.

        2-D mode labels every plane in one pass and emits the rows in the
        order the reference writes them (plane, barcode index, label order);
        decode_3d labels the volume with 26-connectivity."""
        vol, lab = self._enqueue_decode(imageStack, scaleFactors, backgrounds,
                                        fov=fov)
        pending = self._enqueue_table(vol, lab, typed=database is not None,
                                      wantXY64=globalAligner is not None)
        bc = self._finish_fov(pending, vol, fov, globalAligner, zPositions,
                              database)
        self.flush_images()
        return bc, vol

    def run_fovs(self, imageStacks, scaleFactors, backgrounds, fovs=None,
                 globalAligner=None, zPositions=None, database=None):
        """run_fov over a sequence of FOVs (the loop the reference spreads
        over fragment processes, decode.py:68-75), yielding (barcodes,
        DecodedVolume) per FOV in order.

        Software pipeline, three FOVs in flight: the decode kernel and the
        labelling of FOV k+1 are queued before the host waits for FOV k's
        region counts, so that wait costs the GPU nothing; the labelling and
        the sort / region statistics of FOV k run on two more compute streams
        beside that decode kernel, its frame rows cross PCIe on a copy stream
        and FOV k-1's DataFrame is assembled on the host meanwhile
        (``overlap=False`` at construction: everything on one stream)."""
        dev = self.decoder._dev()
        side = self._side_stream(dev)
        lbl = None
        typed = database is not None
        xy64 = globalAligner is not None
        if side is not None and self._table_stream(dev) is not None:
            lbl = self._label_stream(dev)
        from collections import deque
        decoded = deque()   # (vol, labels pending, fov): stage 1 queued
        tabled = deque()    # (vol, frame pending, fov): stage 2 queued

        def table(d):
            pending = self._enqueue_table(d[0], d[1], side, typed, xy64)
            tabled.append((d[0], pending, d[2]))

        def finish(t):
            return self._finish_fov(t[1], t[0], t[2], globalAligner,
                                    zPositions, database), t[0]

        # The host blocks twice per FOV: on the region counts that size FOV
        # k's tables (inherent) and on FOV j's frame rows.  decodeAhead decode
        # kernels are queued beyond the FOV whose counts are awaited and
        # frameLag FOVs may have their tables and rows in flight, so neither
        # wait leaves the main stream without a queued decode kernel.
        ahead = self.decodeAhead if side is not None else 1
        lag = self.frameLag if side is not None else 1
        for k, stack in enumerate(imageStacks):
            fov = k if fovs is None else fovs[k]
            decoded.append(self._enqueue_decode(stack, scaleFactors,
                                                backgrounds, lbl, fov) + (fov,))
            if len(decoded) > ahead:
                table(decoded.popleft())
            while len(tabled) > lag:
                yield finish(tabled.popleft())
        while decoded:
            table(decoded.popleft())
            while len(tabled) > lag:
                yield finish(tabled.popleft())
        while tabled:
            if len(tabled) == 1:
                self.flush_images()      # every file on disk before the last FOV
            yield finish(tabled.popleft())
        self.flush_images()

    def run_fovs_sharded(self, stackProvider, scaleFactors, backgrounds, fovs,
                         rank=None, world=None, group=None,
                         globalAligner=None, zPositions=None, database=None):
        """This rank's share of the fragment loop (core/analysistask.py:
        285-297 runs ``_run_analysis(i)`` for every fragment, one process
        each; analysis/decode.py:71-72: a fragment is a FOV).  FOV number i of
        ``fovs`` belongs to rank ``i mod world``; FOVs are independent, so
        there is no collective.  ``stackProvider(fov)`` returns the uint16
        [B,Z,H,W] device tensor of a FOV and is only called for this rank's
        FOVs.  rank / world default to the torch.distributed group's.
        Yields (fov, barcodes, DecodedVolume) in this rank's FOV order."""
        if rank is None or world is None:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                rank = dist.get_rank(group)
                world = dist.get_world_size(group)
            else:
                rank, world = 0, 1
        mine = shard_fovs(fovs, rank, world)
        stacks = (stackProvider(f) for f in mine)
        results = self.run_fovs(stacks, scaleFactors, backgrounds, fovs=mine,
                                globalAligner=globalAligner,
                                zPositions=zPositions, database=database)
        for f, (bc, vol) in zip(mine, results):
            yield f, bc, vol

    def run_fovs_from_host(self, hostStacks, scaleFactors, backgrounds,
                           fovs=None, globalAligner=None, zPositions=None,
                           preprocess=None, depth=3):
        """run_fovs for stacks that start in HOST memory (what
        ``get_processed_image_set`` returns, or -- with ``preprocess`` set to a
        ``merlin_amd.preprocess.DeconvolutionPreprocess`` -- the warped camera
        frames it would read): uint16 [B,Z,H,W] numpy arrays or torch CPU
        tensors.  Uploads run on a copy stream through pinned staging buffers,
        one FOV ahead of the kernels, so PCIe (0.94 GB per 7-z FOV) overlaps
        compute.  Yields (barcodes, DecodedVolume) per FOV in order."""
        dev = self.decoder._dev()
        up = HostStackUploader(dev, depth=depth)
        stacks = up.stream(hostStacks)
        if preprocess is not None:
            stacks = (preprocess.preprocess_device(s, histogram=False)[0]
                      for s in stacks)
        return self.run_fovs(stacks, scaleFactors, backgrounds, fovs=fovs,
                             globalAligner=globalAligner, zPositions=zPositions)


class HostStackUploader(object):
    """Pinned, multi-buffered host-to-device upload of image stacks on its own
    HIP stream.

    ``stream(hostStacks)`` yields device tensors in order; the copy of stack
    k+1 is queued before stack k is handed out, so it crosses PCIe while the
    consumer's kernels for stack k run.  A device buffer is reused ``depth``
    stacks later, after the kernels the consumer queued for its previous
    content have finished (an event recorded when the consumer asks for the
    next stack): the consumer must have queued every kernel that reads stack k
    on the current stream before it asks for stack k+1, as run_fovs does.
    Stacks already in page-locked memory (``torch.Tensor.is_pinned()``) are
    copied straight from there; anything else goes through a pinned staging
    buffer first (a multi-threaded host copy)."""

    def __init__(self, device, depth=3):
        if depth < 2:
            raise ValueError('depth must be >= 2')
        self.device = torch.device(device)
        self.depth = depth
        with torch.cuda.device(self.device):
            self.copy_stream = torch.cuda.Stream(self.device)
        self._dev = [None] * depth
        self._stage = [None, None]
        self._stage_free = [None, None]

    def _as_tensor(self, h):
        if isinstance(h, torch.Tensor):
            if h.device.type != 'cpu':
                raise ValueError('host stack expected, got a device tensor')
            return h.contiguous()
        return torch.from_numpy(np.ascontiguousarray(h))

    def _issue(self, k, h, consumed):
        """Queue the upload of host stack h as stack k; returns (device
        tensor, event that marks the copy complete)."""
        t = self._as_tensor(h)
        slot = k % self.depth
        buf = self._dev[slot]
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
            buf = torch.empty(t.shape, dtype=t.dtype, device=self.device)
            self._dev[slot] = buf
        if not t.is_pinned():
            s = k % 2
            if self._stage_free[s] is not None:
                self._stage_free[s].synchronize()      # its last upload has left
            st = self._stage[s]
            if st is None or st.shape != t.shape or st.dtype != t.dtype:
                st = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                self._stage[s] = st
            st.copy_(t)
            t = st
        with torch.cuda.stream(self.copy_stream):
            if consumed[slot] is not None:
                self.copy_stream.wait_event(consumed[slot])
            buf.copy_(t, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.copy_stream)
        if t is self._stage[k % 2]:
            self._stage_free[k % 2] = done
        return buf, done

    def stream(self, hostStacks):
        consumed = [None] * self.depth
        it = iter(hostStacks)
        pending = None
        k = 0
        for h in it:
            nxt = self._issue(k, h, consumed)
            if pending is not None:
                buf, done = pending
                torch.cuda.current_stream(self.device).wait_event(done)
                yield buf
                # the consumer has queued its kernels for that stack by now
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                consumed[(k - 1) % self.depth] = ev
            pending = nxt
            k += 1
        if pending is not None:
            buf, done = pending
            torch.cuda.current_stream(self.device).wait_event(done)
            yield buf
