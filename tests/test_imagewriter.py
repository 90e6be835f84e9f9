"""Decoded-image writer (merlin_amd/imagewriter.py): the TIFF container on CPU,
the device-side interleave on the GPU.  The container was also read back with
tifffile 2021 in the build container (is_imagej, 21 pages, pixel-identical);
tifffile is not importable where the tests run, so the checks here use the
module's own reader plus direct inspection of the directory entries."""
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
