"""Reference-side fixture for the decoded-image writer.

Run in the build container only:

    /opt/conda/bin/python3.9 oracle/make_tiff_golden.py

Writes a small (z, 3, H, W) float32 stack the way the reference does --
``DataSet.writer_for_analysis_images`` opens ``tifffile.TiffWriter(path,
imagej=True)`` (merlin/core/dataset.py:205-220) and
``Decode._save_decoded_images`` calls ``save(image.astype(np.float32),
photometric='MINISBLACK', metadata=analysis_tiff_description(z, 3))`` once per
page (merlin/analysis/decode.py:196-213) -- with tifffile 2021.7.2, the
version this container's conda interpreter has, and stores the FILE BYTES
together with what tifffile reads back from them (pages, ImageDescription of
every page, is_imagej, imagej_metadata).  tifffile is a third-party library,
not part of the reference; the statements above are the reference's call
sequence restated.  tifffile 2021.7.2 refuses that page-by-page sequence for an
ImageJ file ("non-contiguous series"), so the fixture is what the same writer
makes of the same pages and metadata in ONE save() call (recorded in the
fixture as ``per_page_save_ok``).

Output: ``tests/golden/tiff_decoded_reference.npz`` (data only).
"""
import os
import tempfile

import numpy as np
import tifffile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')


def analysis_tiff_description(sliceCount, frameCount):
    # merlin/core/dataset.py:222-230 (a dictionary literal: data, restated)
    return {'ImageJ': '1.47a\n', 'images': sliceCount * frameCount,
            'channels': 1, 'slices': sliceCount, 'frames': frameCount,
            'hyperstack': True, 'loop': False}


def main():
    rng = np.random.default_rng(5)
    Z, H, W = 2, 6, 9
    decoded = rng.integers(-1, 140, (Z, H, W)).astype(np.int16)
    magnitude = rng.uniform(0, 30, (Z, H, W)).astype(np.float32)
    distance = rng.uniform(0, 1.2, (Z, H, W)).astype(np.float32)
    meta = analysis_tiff_description(Z, 3)
    path = os.path.join(tempfile.mkdtemp(), 'decoded_0.tif')
    stack = np.empty((Z * 3, H, W), dtype=np.float32)
    stack[0::3], stack[1::3], stack[2::3] = decoded, magnitude, distance
    per_page_ok = True
    try:
        with tifffile.TiffWriter(path, imagej=True) as out:
            for page in stack:
                out.save(page, photometric='MINISBLACK', metadata=meta)
    except ValueError as e:
        # tifffile 2021.7.2: "ImageJ does not support non-contiguous series" on
        # the second save() -- the reference's page-by-page sequence needs the
        # tifffile of its time.  Same writer, same pages, same metadata, one
        # call: the ImageJ hyperstack that sequence stood for.
        per_page_ok = False
        print('per-page save refused by tifffile {}: {}'.format(
            tifffile.__version__, e))
        with tifffile.TiffWriter(path, imagej=True) as out:
            out.save(stack, photometric='MINISBLACK', metadata=meta)
    raw = np.frombuffer(open(path, 'rb').read(), dtype=np.uint8)
    with tifffile.TiffFile(path) as tf:
        pages = np.array([p.asarray() for p in tf.pages])
        descs = [p.tags['ImageDescription'].value if 'ImageDescription' in p.tags
                 else '' for p in tf.pages]
        tags = [sorted(int(t.code) for t in p.tags.values()) for p in tf.pages]
        print('tifffile', tifffile.__version__, 'is_imagej', tf.is_imagej,
              'pages', len(tf.pages), 'series', len(tf.series),
              [s.shape for s in tf.series])
        print('imagej_metadata', tf.imagej_metadata)
        print('first description', repr(descs[0]))
        print('tags per page', tags[0], tags[1])
        ij = tf.imagej_metadata or {}
        is_ij = bool(tf.is_imagej)
    # the reference reads a page back by index (dataset.py:200-203)
    with tifffile.TiffFile(path) as tf:
        assert np.array_equal(tf.asarray(key=4), magnitude[1])
    # The file merlin_amd.imagewriter writes for the same pages (with the
    # reference's metadata dictionary, which the tifffile of the reference's
    # time passed through: slices = z, frames = 3), checked HERE with tifffile:
    # an ImageJ hyperstack with the same pages.  Its bytes go into the fixture;
    # the test on the GPU box, where tifffile is absent, compares the writer's
    # output with them.
    import sys
    sys.path.insert(0, os.path.dirname(HERE))
    from merlin_amd import imagewriter
    ours = os.path.join(os.path.dirname(path), 'ours.tif')
    imagewriter.write_hyperstack(ours, stack, meta)
    with tifffile.TiffFile(ours) as tf:
        assert tf.is_imagej and len(tf.pages) == Z * 3
        got = np.array([p.asarray() for p in tf.pages])
        assert got.dtype == np.float32 and np.array_equal(got, stack)
        assert np.array_equal(tf.asarray(key=4), magnitude[1])      # dataset.py:200-203
        ours_meta = tf.imagej_metadata
        ours_series = [tuple(s.shape) for s in tf.series]
        ours_axes = [s.axes for s in tf.series]
        print('ours: imagej_metadata', ours_meta, 'series', ours_series, ours_axes)
        assert ours_meta['slices'] == Z and ours_meta['frames'] == 3
        assert ours_meta['images'] == Z * 3 and ours_meta['hyperstack'] is True
    ours_raw = np.frombuffer(open(ours, 'rb').read(), dtype=np.uint8)
    np.savez_compressed(
        os.path.join(OUT, 'tiff_decoded_reference.npz'),
        file_bytes=raw, decoded=decoded, magnitude=magnitude, distance=distance,
        pages=pages, descriptions=np.array(descs),
        first_page_tags=np.array(tags[0]), other_page_tags=np.array(tags[1]),
        is_imagej=is_ij, tifffile_version=tifffile.__version__,
        per_page_save_ok=per_page_ok, our_file_bytes=ours_raw,
        our_series_shape=np.array(ours_series[0]), our_series_axes=ours_axes[0],
        imagej_keys=np.array(sorted(ij.keys())),
        imagej_values=np.array([str(ij[k]) for k in sorted(ij.keys())]))
    print('wrote', len(raw), 'file bytes')


if __name__ == '__main__':
    main()
