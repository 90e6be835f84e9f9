"""Decoded-image writer (merlin_amd/imagewriter.py): the TIFF container on CPU,
the device-side interleave on the GPU.  tifffile -- what the reference writes
and reads these files with (core/dataset.py:196-220) -- is not importable where
the tests run; tests/golden/tiff_decoded_reference.npz (made by
oracle/make_tiff_golden.py under the build container's tifffile 2021.7.2)
holds (a) the bytes of a file tifffile wrote for a small stack, with what
tifffile reads back from it, and (b) the bytes of the file this module's
writer makes for the same stack, which that script checked WITH tifffile
(ImageJ hyperstack, slices / frames of the reference's description, pages
identical, page-by-index read as dataset.py:200-203)."""
import os
import struct

import numpy as np
import pytest

from merlin_amd import imagewriter as iw


def test_description_matches_reference_metadata():
    d = iw.analysis_tiff_description(7, 3)      # dataset.py:222-230
    assert d == {'ImageJ': '1.47a\n', 'images': 21, 'channels': 1, 'slices': 7,
                 'frames': 3, 'hyperstack': True, 'loop': False}
    text = iw._description_bytes(d)
    assert text == (b'ImageJ=1.47a\nimages=21\nchannels=1\nslices=7\nframes=3\n'
                    b'hyperstack=true\nloop=false\n\0')


@pytest.mark.parametrize('shape', [(21, 50, 70), (3, 1, 1), (6, 33, 17)])
def test_round_trip(tmp_path, shape):
    rng = np.random.default_rng(1)
    pages = rng.normal(size=shape).astype(np.float32)
    pages[0, 0, 0] = -1.0
    path = str(tmp_path / 'decoded.tif')
    iw.write_hyperstack(path, pages, iw.analysis_tiff_description(shape[0] // 3, 3))
    got, desc = iw.read_hyperstack(path)
    assert got.dtype == np.float32 and np.array_equal(got, pages)
    assert desc.startswith('ImageJ=1.47a\nimages={}\n'.format(shape[0]))
    raw = open(path, 'rb').read()
    assert raw[:4] == b'II*\0'
    first = struct.unpack_from('<I', raw, 4)[0]
    n_tags = struct.unpack_from('<H', raw, first)[0]
    tags = [struct.unpack_from('<HHII', raw, first + 2 + 12 * i) for i in range(n_tags)]
    assert [t[0] for t in tags] == sorted(t[0] for t in tags)       # ascending, as TIFF asks
    byid = {t[0]: t for t in tags}
    assert byid[256][3] == shape[2] and byid[257][3] == shape[1]
    assert byid[258][3] == 32 and byid[339][3] == 3 and byid[262][3] == 1
    assert byid[279][3] == shape[1] * shape[2] * 4


def test_rejects_bad_input(tmp_path):
    with pytest.raises(ValueError):
        iw.write_hyperstack(str(tmp_path / 'x.tif'), np.zeros((4, 4), np.float32), {})
    p = tmp_path / 'junk.tif'
    p.write_bytes(b'MM\0*....')
    with pytest.raises(ValueError):
        iw.read_hyperstack(str(p))


@pytest.mark.gpu
def test_save_decoded_images(gpu_device, tmp_path):
    """Page order of Decode._save_decoded_images (decode.py:202-213): per z
    plane decoded, magnitude, distance, all as float32."""
    import torch
    from conftest import ArrayCodebook, load_golden, synth_stack
    from merlin_amd.decode import Decode
    cb = load_golden('g2_synth192')['codebook']
    img = synth_stack(cb, (3, 64, 96), 40, seed=3)
    task = Decode(ArrayCodebook(cb), {'crop_width': 2}, device=gpu_device)
    bc, vol = task.run_fov(torch.from_numpy(img).to(gpu_device), np.full(16, 400.0),
                           np.zeros(16), fov=0)
    path = str(tmp_path / 'decoded_0.tif')
    iw.save_decoded_images(path, vol)
    pages, desc = iw.read_hyperstack(path)
    assert pages.shape == (9, 64, 96) and 'slices=3\nframes=3' in desc
    for z in range(3):
        assert np.array_equal(pages[3 * z], vol.decoded[z].cpu().numpy().astype(np.float32))
        assert np.array_equal(pages[3 * z + 1], vol.magnitudes[z].cpu().numpy())
        assert np.array_equal(pages[3 * z + 2],
                              vol.distances[z].cpu().numpy().astype(np.float32))


@pytest.fixture(scope='module')
def tiff_fixture():
    from conftest import GOLDEN_DIR
    g = np.load(os.path.join(GOLDEN_DIR, 'tiff_decoded_reference.npz'))
    return {k: g[k] for k in g.files}


def _stack(fx):
    Z = fx['decoded'].shape[0]
    stack = np.empty((Z * 3,) + fx['decoded'].shape[1:], dtype=np.float32)
    stack[0::3], stack[1::3], stack[2::3] = fx['decoded'], fx['magnitude'], fx['distance']
    return stack


def test_reader_reads_a_tifffile_written_stack(tiff_fixture, tmp_path):
    """read_hyperstack on the file tifffile itself wrote: same pages, same
    ImageDescription as tifffile reads from it."""
    fx = tiff_fixture
    path = str(tmp_path / 'ref.tif')
    open(path, 'wb').write(fx['file_bytes'].tobytes())
    pages, desc = iw.read_hyperstack(path)
    assert pages.dtype == np.float32 and np.array_equal(pages, fx['pages'])
    assert np.array_equal(pages, _stack(fx))
    assert desc.strip() == str(fx['descriptions'][0]).strip()      # tifffile strips the text
    # page by index, as DataSet.get_analysis_image does (dataset.py:200-203)
    assert np.array_equal(pages[1 * 3 + 1], fx['magnitude'][1])


def test_writer_output_is_the_file_tifffile_validated(tiff_fixture, tmp_path):
    """write_hyperstack for the fixture's stack and the reference's metadata
    dictionary gives, byte for byte, the file that oracle/make_tiff_golden.py
    opened with tifffile (is_imagej, slices = z, frames = 3, pages equal)."""
    fx = tiff_fixture
    Z = fx['decoded'].shape[0]
    path = str(tmp_path / 'ours.tif')
    iw.write_hyperstack(path, _stack(fx), iw.analysis_tiff_description(Z, 3))
    assert open(path, 'rb').read() == fx['our_file_bytes'].tobytes()
    assert str(fx['our_series_axes']) == 'TZYX'
    assert fx['our_series_shape'].tolist() == [3, Z] + list(fx['decoded'].shape[1:])
    # the baseline tags tifffile writes per page are all present in ours
    raw = fx['our_file_bytes'].tobytes()
    first = struct.unpack_from('<I', raw, 4)[0]
    n_tags = struct.unpack_from('<H', raw, first)[0]
    ours = {struct.unpack_from('<H', raw, first + 2 + 12 * i)[0] for i in range(n_tags)}
    assert set(fx['first_page_tags'].tolist()) - {305} <= ours      # 305 = Software
