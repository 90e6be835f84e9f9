"""Decoded-image writer: counterpart of Decode._save_decoded_images
(merlin/analysis/decode.py:196-213) and DataSet.analysis_tiff_description /
writer_for_analysis_images (merlin/core/dataset.py:205-230).

The reference writes, per FOV, a float32 ImageJ hyperstack whose pages are
interleaved per z plane: decoded index, pixel magnitude, distance (``slices`` =
z planes, ``frames`` = 3).  Here the three device-resident images of a
DecodedVolume are cast and interleaved on the GPU, cross PCIe once into a
page-locked buffer and are written as one contiguous block behind the page
directory of an uncompressed little-endian TIFF (classic, or BigTIFF above
4 GB).  No TIFF library is needed for writing; ``read_hyperstack`` is the
matching minimal reader (used by the tests, and a way to get the arrays back).
"""
import struct
from typing import Tuple

import numpy as np

_TYPE_SIZES = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 16: 8}


def analysis_tiff_description(sliceCount: int, frameCount: int) -> dict:
    """dataset.py:222-230"""
    return {'ImageJ': '1.47a\n', 'images': sliceCount * frameCount,
            'channels': 1, 'slices': sliceCount, 'frames': frameCount,
            'hyperstack': True, 'loop': False}


def _description_bytes(meta: dict) -> bytes:
    """ImageJ's ImageDescription text for the metadata dictionary."""
    lines = ['ImageJ=' + str(meta.get('ImageJ', '1.47a')).strip()]
    for k, v in meta.items():
        if k == 'ImageJ':
            continue
        if isinstance(v, bool):
            v = 'true' if v else 'false'
        lines.append('{}={}'.format(k, v))
    return ('\n'.join(lines) + '\n').encode('ascii') + b'\0'


def write_hyperstack(path: str, pages: np.ndarray, metadata: dict) -> None:
    """pages: float32 [n, H, W] C-contiguous; one TIFF page each
    (photometric MINISBLACK, one strip per page, uncompressed)."""
    pages = np.ascontiguousarray(pages, dtype='<f4')
    if pages.ndim != 3:
        raise ValueError('pages must be [n, H, W]')
    n, H, W = pages.shape
    page_bytes = H * W * 4
    desc = _description_bytes(metadata)
    big = 16 + n * (8 + 11 * 20 + 8) + len(desc) + n * page_bytes >= (1 << 32) - (1 << 20)
    # tag, type, count, value (inline) -- the same for every page except offsets
    if big:
        head, cnt_fmt, ent_fmt, nxt_fmt, off_type = 16, '<Q', '<HHQQ', '<Q', 16
    else:
        head, cnt_fmt, ent_fmt, nxt_fmt, off_type = 8, '<H', '<HHII', '<I', 4
    n_tags = 11
    ifd_size = struct.calcsize(cnt_fmt) + n_tags * struct.calcsize(ent_fmt) \
        + struct.calcsize(nxt_fmt)
    desc_off = head + n * ifd_size
    data_off = (desc_off + len(desc) + 15) & ~15
    with open(path, 'wb') as f:
        if big:
            f.write(struct.pack('<2sHHHQ', b'II', 43, 8, 0, head))
        else:
            f.write(struct.pack('<2sHI', b'II', 42, head))
        for i in range(n):
            tags = [
                (254, 4, 1, 0),                      # NewSubfileType
                (256, 4, 1, W),                      # ImageWidth
                (257, 4, 1, H),                      # ImageLength
                (258, 3, 1, 32),                     # BitsPerSample
                (259, 3, 1, 1),                      # Compression: none
                (262, 3, 1, 1),                      # Photometric: MINISBLACK
                (270, 2, len(desc), desc_off),       # ImageDescription
                (273, off_type, 1, data_off + i * page_bytes),   # StripOffsets
                (277, 3, 1, 1),                      # SamplesPerPixel
                (278, 4, 1, H),                      # RowsPerStrip
                (279, off_type, 1, page_bytes),      # StripByteCounts
                (339, 3, 1, 3),                      # SampleFormat: IEEE float
            ]
            if i > 0:        # the description belongs to the first page only
                tags = [t for t in tags if t[0] != 270]
            else:
                tags = [t for t in tags if t[0] != 254]
            assert len(tags) == n_tags
            f.write(struct.pack(cnt_fmt, n_tags))
            for tag, typ, count, value in tags:
                f.write(struct.pack(ent_fmt, tag, typ, count, value))
            nxt = head + (i + 1) * ifd_size if i + 1 < n else 0
            f.write(struct.pack(nxt_fmt, nxt))
        f.write(desc)
        f.write(b'\0' * (data_off - desc_off - len(desc)))
        f.write(memoryview(pages).cast('B'))


def read_hyperstack(path: str) -> Tuple[np.ndarray, str]:
    """(float32 [n, H, W], ImageDescription text) of a file written by
    write_hyperstack (uncompressed, one strip per page)."""
    with open(path, 'rb') as f:
        buf = f.read()
    if buf[:2] != b'II':
        raise ValueError('not a little-endian TIFF')
    magic = struct.unpack_from('<H', buf, 2)[0]
    if magic == 43:
        off = struct.unpack_from('<Q', buf, 8)[0]
        cnt_fmt, ent_fmt, nxt_fmt, inline = '<Q', '<HHQQ', '<Q', 8
    elif magic == 42:
        off = struct.unpack_from('<I', buf, 4)[0]
        cnt_fmt, ent_fmt, nxt_fmt, inline = '<H', '<HHII', '<I', 4
    else:
        raise ValueError('not a TIFF')
    pages, desc = [], ''
    while off:
        cnt = struct.unpack_from(cnt_fmt, buf, off)[0]
        pos = off + struct.calcsize(cnt_fmt)
        tags = {}
        for _ in range(cnt):
            tag, typ, count, value = struct.unpack_from(ent_fmt, buf, pos)
            if _TYPE_SIZES[typ] * count <= inline and typ == 3:
                value &= 0xffff
            tags[tag] = (typ, count, value)
            pos += struct.calcsize(ent_fmt)
        off = struct.unpack_from(nxt_fmt, buf, pos)[0]
        W, H = tags[256][2], tags[257][2]
        if tags[258][2] != 32 or tags[339][2] != 3 or tags[259][2] != 1:
            raise ValueError('only uncompressed float32 pages are read')
        if 270 in tags:
            _, count, o = tags[270]
            desc = buf[o:o + count].rstrip(b'\0').decode('ascii')
        o = tags[273][2]
        pages.append(np.frombuffer(buf, dtype='<f4', count=H * W,
                                   offset=o).reshape(H, W))
    return np.array(pages), desc


def decoded_image_stack(vol, pinned: bool = True) -> np.ndarray:
    """float32 [Z*3, H, W] in the reference's page order (per z: decoded,
    magnitude, distance; decode.py:202-213) from a DecodedVolume: cast and
    interleaved on the GPU, one device-to-host copy."""
    import torch
    dec, mag, dist = vol.decoded, vol.magnitudes, vol.distances
    if dec.dim() == 2:
        dec, mag, dist = dec[None], mag[None], dist[None]
    Z, H, W = dec.shape
    stack = torch.empty((Z, 3, H, W), dtype=torch.float32, device=dec.device)
    stack[:, 0] = dec
    stack[:, 1] = mag
    stack[:, 2] = dist
    host = torch.empty((Z * 3, H, W), dtype=torch.float32, pin_memory=pinned)
    host.copy_(stack.view(Z * 3, H, W), non_blocking=False)
    return host.numpy()


def save_decoded_images(path: str, vol) -> None:
    """Decode._save_decoded_images for one FOV."""
    pages = decoded_image_stack(vol)
    write_hyperstack(path, pages,
                     analysis_tiff_description(pages.shape[0] // 3, 3))


class AsyncImageWriter(object):
    """Decode._save_decoded_images off the critical path of a FOV loop.

    ``submit(path, vol)`` queues, on a stream of its own, the float32 cast +
    page interleave of the volume's three images into a device staging buffer
    and its copy into a page-locked host buffer (352 MB for a 7-z 2048 x 2048
    FOV: 6.4 ms of PCIe), and hands the file write to a worker thread that
    waits for the copy's event; nothing blocks the caller unless every staging
    slot is still busy (``slots`` FOVs in flight; file writes release the
    interpreter lock, so several run side by side).  ``flush()`` returns once
    every file queued so far is on disk and re-raises a worker's error.  The
    files are byte-identical to ``save_decoded_images``'s."""

    def __init__(self, device, workers: int = 4, slots: int = None):
        import torch
        from concurrent.futures import ThreadPoolExecutor
        self.device = torch.device(device)
        self.slots = slots or workers + 1
        with torch.cuda.device(self.device):
            self.stream = torch.cuda.Stream(self.device)
        self._pool = ThreadPoolExecutor(max_workers=workers)
        self._free = []          # (device staging, pinned host) of finished writes
        self._busy = []          # (future, buffers) in submission order
        self._made = 0

    def _buffers(self, shape):
        import torch
        while True:
            for i, (d, h) in enumerate(self._free):
                if tuple(d.shape) == tuple(shape):
                    return self._free.pop(i)
            if self._made < self.slots:
                self._free = []                      # other shapes: let them go
                self._made += 1
                return (torch.empty(shape, dtype=torch.float32, device=self.device),
                        torch.empty(shape, dtype=torch.float32, pin_memory=True))
            fut, bufs = self._busy.pop(0)            # oldest write still running
            fut.result()
            self._free.append(bufs)

    def submit(self, path: str, vol) -> None:
        import torch
        dec, mag, dist = vol.decoded, vol.magnitudes, vol.distances
        if dec.dim() == 2:
            dec, mag, dist = dec[None], mag[None], dist[None]
        Z, H, W = dec.shape
        self._reap()
        stage, host = self._buffers((Z, 3, H, W))
        cur = torch.cuda.current_stream(self.device)
        with torch.cuda.device(self.device):
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                stage[:, 0] = dec
                stage[:, 1] = mag
                stage[:, 2] = dist
                host.copy_(stage, non_blocking=True)
                done = torch.cuda.Event()
                done.record(self.stream)
            for t in (dec, mag, dist):
                t.record_stream(self.stream)
        meta = analysis_tiff_description(Z, 3)

        def work():
            done.synchronize()
            write_hyperstack(path, host.numpy().reshape(Z * 3, H, W), meta)

        self._busy.append((self._pool.submit(work), (stage, host)))

    def _reap(self):
        while self._busy and self._busy[0][0].done():
            fut, bufs = self._busy.pop(0)
            fut.result()
            self._free.append(bufs)

    def flush(self) -> None:
        while self._busy:
            fut, bufs = self._busy.pop(0)
            fut.result()
            self._free.append(bufs)

    def close(self) -> None:
        self.flush()
        self._pool.shutdown()
