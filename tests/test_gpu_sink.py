"""GPU: what sits either side of the decode kernels in the Decode task
(merlin/analysis/decode.py:92-175, 196-237): the typed record sink
(util/barcodedb.py:46-65, 250-266) fed straight from the device, z-duplicate
removal on the database's table, barcode filtering on top of it
(analysis/filterbarcodes.py), the decoded-image file, the device-volume cache
of the drop-in decoder under in-place edits, and the sharded fragment loop.
Checker: the oracle's DataFrame cast to the database's column types."""
import warnings

import numpy as np
import pandas
import pytest

import merfish_oracle as oracle
from conftest import (ArrayCodebook, AffineAligner, load_golden, synth_stack)
from test_gpu_fov import _oracle_fov

pytestmark = pytest.mark.gpu

INT_COLUMNS = ('barcode_id', 'fov', 'area', 'cell_index')


class NamedCodebook(ArrayCodebook):
    """every third codeword is a blank (filterbarcodes needs the split)"""

    def get_blank_indexes(self):
        return list(range(0, self._b.shape[0], 3))

    def get_coding_indexes(self):
        return [i for i in range(self._b.shape[0]) if i % 3]


def _assert_typed_equal(got: pandas.DataFrame, exp: pandas.DataFrame, schema,
                        rtol=1.2e-7):
    assert list(got.columns) == list(schema)
    assert len(got) == len(exp)
    for c, dt in schema.items():
        assert got[c].dtype == np.dtype(dt), c
        a, b = got[c].to_numpy(), exp[c].to_numpy()
        if c in INT_COLUMNS:
            assert np.array_equal(a, b), c
        else:
            # float64 sums differ from numpy's pairwise order in the last
            # bits, so the float32 cast may land one ulp apart
            np.testing.assert_allclose(a, b, rtol=rtol, atol=0, err_msg=c)


@pytest.mark.parametrize('decode3d', [False, True])
def test_typed_records_into_database(gpu_device, decode3d):
    import torch
    from merlin_amd import barcodedb
    from merlin_amd.decode import Decode
    cb = load_golden('g2_synth192')['codebook']
    # (3-D: the crop compares x with shape[0] = Z, the reference's quirk)
    shape = (24, 40, 72) if decode3d else (3, 80, 136)
    img = synth_stack(cb, shape, 90, seed=21)
    rng = np.random.default_rng(3)
    sf = rng.uniform(300, 600, 16)
    bg = rng.uniform(90, 110, 16)
    tform = np.float32([[0.108, 0, 3.5], [0, 0.108, -2.25], [0, 0, 1]])
    params = {'crop_width': 4, 'minimum_area': 2, 'decode_3d': decode3d}
    book = NamedCodebook(cb)
    task = Decode(book, params, device=gpu_device)
    db = barcodedb.BarcodeDB(book)
    stack = torch.from_numpy(img).to(gpu_device)
    got, vol = task.run_fov(stack, sf, bg, fov=5,
                            globalAligner=AffineAligner(tform), database=db)
    schema = barcodedb.bc_column_types(16)
    exp64 = _oracle_fov(cb, img, sf, bg, task.parameters, 5, tform, decode3d)
    exp = exp64.astype(schema)
    assert len(exp) > 10
    # (3-D: regionprops' float32 means vs ordered sums, 1e-6 as in test_gpu_fov)
    _assert_typed_equal(got, exp, schema, 1e-6 if decode3d else 1.2e-7)
    assert np.array_equal(got.index.to_numpy(), exp.index.to_numpy())
    assert got is not None and len(db.get_barcodes()) == len(exp)
    # the float64 frame of the plain path, cast by the database, is the same table
    bc64, _ = task.run_fov(stack, sf, bg, fov=5,
                           globalAligner=AffineAligner(tform))
    db2 = barcodedb.BarcodeDB(book)
    db2.write_barcodes(bc64, fov=5)
    via_host = db2.get_barcodes(fov=5)
    for c in schema:
        assert np.array_equal(via_host[c].to_numpy(), got[c].to_numpy()), c

    # barcode filtering on top of the device-made table (filterbarcodes.py:
    # 52-63, 203-235) == the same on the oracle-made table
    from merlin_amd import filterbarcodes
    odb = barcodedb.BarcodeDB(book)
    odb.write_barcodes(exp64, fov=5)
    thr_i = float(np.median(exp['mean_intensity']))
    a = filterbarcodes.filter_barcodes(db, 5, 3, thr_i, 0.5)
    b = filterbarcodes.filter_barcodes(odb, 5, 3, thr_i, 0.5)
    assert len(a) == len(b) and 0 < len(a) < len(exp)
    assert np.array_equal(a['barcode_id'].to_numpy(), b['barcode_id'].to_numpy())
    hists = []
    for d in (db, odb):
        t = filterbarcodes.AdaptiveThreshold(book, distance_threshold=0.5167)
        t.set_intensity_bins_from_maximum(
            float(d.get_barcodes(fov=5)['mean_intensity'].max()))
        t.add_barcodes(d.get_barcodes(fov=5))
        hists.append((t.get_blank_count_histogram(),
                      t.get_coding_count_histogram()))
    # (3-D regions mostly exceed the default area bins)
    assert hists[0][0].sum() + hists[0][1].sum() > (0 if decode3d else 10)
    # a record whose float32 statistic sits one ulp apart may change bin
    assert np.abs(hists[0][0] - hists[1][0]).sum() <= 2
    assert np.abs(hists[0][1] - hists[1][1]).sum() <= 2


def test_z_duplicate_removal_on_the_database(gpu_device):
    """decode.py:170-175: read the FOV back from the database (float32
    columns), filter, empty, rewrite -- against the same steps on the oracle's
    frame."""
    import torch
    from merlin_amd import barcodedb, barcodefilters
    from merlin_amd.decode import Decode
    cb = load_golden('g2_synth192')['codebook']
    img = synth_stack(cb, (4, 72, 104), 60, seed=8)      # spots span planes
    sf, bg = np.full(16, 400.0), np.zeros(16)
    params = {'crop_width': 3, 'remove_z_duplicated_barcodes': True}
    book = ArrayCodebook(cb)
    task = Decode(book, params, device=gpu_device)
    db = barcodedb.BarcodeDB(book)
    zpos = [0.0, 1.5, 3.0, 4.5]
    got, _ = task.run_fov(torch.from_numpy(img).to(gpu_device), sf, bg, fov=2,
                          zPositions=zpos, database=db)
    exp64 = _oracle_fov(cb, img, sf, bg, task.parameters, 2, None, False)
    odb = barcodedb.BarcodeDB(book)
    odb.write_barcodes(exp64, fov=2)
    before = len(odb.get_barcodes(fov=2))
    exp = barcodefilters.remove_zplane_duplicates_all_barcodeids(
        odb.get_barcodes(fov=2), 1, np.sqrt(2), zpos)
    assert 0 < len(exp) < before
    _assert_typed_equal(got.reset_index(drop=True), exp.reset_index(drop=True),
                        barcodedb.bc_column_types(16))
    assert len(db.get_barcodes(fov=2)) == len(exp)


def test_decoded_images_written_and_memory_map_accepted(gpu_device, tmp_path):
    import torch
    from merlin_amd import imagewriter
    from merlin_amd.decode import Decode
    cb = load_golden('g2_synth192')['codebook']
    img = torch.from_numpy(synth_stack(cb, (2, 64, 96), 40, seed=4)).to(gpu_device)
    sf, bg = np.full(16, 400.0), np.zeros(16)
    out = tmp_path / 'images'
    task = Decode(ArrayCodebook(cb), {'crop_width': 2}, device=gpu_device,
                  outputDirectory=str(out))
    assert task.parameters['write_decoded_images'] is True
    bc, vol = task.run_fov(img, sf, bg, fov=7)
    pages, desc = imagewriter.read_hyperstack(str(out / 'decoded_7.tif'))
    assert pages.shape == (6, 64, 96) and pages.dtype == np.float32
    assert 'slices=2' in desc and 'frames=3' in desc
    for z in range(2):
        assert np.array_equal(pages[3 * z], vol.decoded[z].cpu().numpy().astype(np.float32))
        assert np.array_equal(pages[3 * z + 1], vol.magnitudes[z].cpu().numpy())
        assert np.array_equal(pages[3 * z + 2],
                              vol.distances[z].cpu().numpy().astype(np.float32))
    # write_decoded_images false: no file; memory_map: accepted, same table
    quiet = Decode(ArrayCodebook(cb), {'crop_width': 2, 'memory_map': True,
                                       'write_decoded_images': False},
                   device=gpu_device, outputDirectory=str(tmp_path / 'none'))
    bc2, _ = quiet.run_fov(img, sf, bg, fov=7)
    assert not (tmp_path / 'none').exists()
    assert bc.equals(bc2)


def test_run_fovs_writes_images_behind_the_loop(gpu_device, tmp_path):
    """run_fovs with an output directory: the files are written by worker
    threads behind the pipeline (more FOVs than staging slots), complete when
    the generator is exhausted, and byte-identical to the synchronous
    save_decoded_images of the same volumes; the barcode tables do not
    change."""
    import torch
    from merlin_amd import imagewriter
    from merlin_amd.decode import Decode
    cb = load_golden('g2_synth192')['codebook']
    stacks = [torch.from_numpy(synth_stack(cb, (2, 64, 96), 40, seed=60 + k)).to(gpu_device)
              for k in range(7)]
    stacks[3] = torch.zeros_like(stacks[3])
    sf, bg = np.full(16, 400.0), np.zeros(16)
    out = tmp_path / 'images'
    task = Decode(ArrayCodebook(cb), {'crop_width': 2}, device=gpu_device,
                  outputDirectory=str(out))
    task.imageWriters = 2                   # 3 staging slots for 7 FOVs
    fovs = [20 + k for k in range(7)]
    got = list(task.run_fovs(iter(stacks), sf, bg, fovs=fovs))
    plain = Decode(ArrayCodebook(cb), {'crop_width': 2}, device=gpu_device)
    for k, (bc, vol) in enumerate(got):
        path = out / 'decoded_{}.tif'.format(fovs[k])
        assert path.exists()
        ref = tmp_path / 'sync_{}.tif'.format(k)
        imagewriter.save_decoded_images(str(ref), vol)
        assert path.read_bytes() == ref.read_bytes()
        ebc, _ = plain.run_fov(stacks[k], sf, bg, fovs[k])
        assert bc.equals(ebc)
    # a second pass over the same task reuses the staging slots
    got2 = list(task.run_fovs(iter(stacks[:2]), sf, bg, fovs=[90, 91]))
    assert (out / 'decoded_91.tif').read_bytes() == (out / 'decoded_21.tif').read_bytes()
    assert len(got2) == 2


def test_extraction_follows_in_place_edits(gpu_device):
    """The drop-in decoder keeps the decoded volume on the GPU between
    decode_pixels and the extraction calls; it must never serve a stale copy
    once the caller has edited the arrays (the reference reads the arrays it is
    given, decoding.py:165-168)."""
    from merlin_amd.decoding import PixelBasedDecoder
    g = load_golden('g2_synth192')
    cb, p = g['codebook'], g['p']
    dec = PixelBasedDecoder(ArrayCodebook(cb), device=gpu_device)
    di, pm, npt, d = dec.decode_pixels(g['image'], g['sf'], g['bg'], p['dthr'],
                                       p['mthr'], p['sigma'])
    assert not di.flags.writeable and not pm.flags.writeable
    with pytest.raises(ValueError):
        di[0, 0] = -1
    code = int(np.bincount(di[di >= 0]).argmax())

    def ours(*arrays):
        return dec.extract_barcodes_with_index(code, *arrays, 0, 0, 0)

    def theirs(*arrays):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            return oracle.extract_barcodes_with_index(code, *arrays, 0, 0, 0)

    first = ours(di, pm, npt, d)
    assert len(first) == len(theirs(di, pm, npt, d)) > 0
    # the caller takes ownership and re-thresholds in place
    di.flags.writeable = True
    cut = float(np.median(pm[di == code]))
    di[pm < cut] = -1
    second = ours(di, pm, npt, d)
    want = theirs(di, pm, npt, d)
    assert len(second) == len(want)
    assert np.array_equal(second['area'].to_numpy(), want['area'].to_numpy())
    assert second['area'].sum() < first['area'].sum()
    # caller-owned arrays (uploaded, cached by checksum): same rule
    di2, pm2 = di.copy(), pm.copy()
    a = ours(di2, pm2, npt, d)
    assert len(a) == len(want)
    di2[pm2 < 1.5 * cut] = -1
    b = ours(di2, pm2, npt, d)
    wb = theirs(di2, pm2, npt, d)
    assert len(b) == len(wb)
    assert np.array_equal(b['area'].to_numpy(), wb['area'].to_numpy())
    assert b['area'].sum() < a['area'].sum()
    # at most two device volumes are held
    assert len(dec._recent) <= 2


def test_single_plane_z_column_follows_arguments(gpu_device):
    """z = zIndex, NaN when it is None (decoding.py:209-211), whichever
    internal path assembles the frame."""
    import torch
    from merlin_amd.decoding import PixelBasedDecoder
    cb = load_golden('g2_synth192')['codebook']
    img = torch.from_numpy(synth_stack(cb, (64, 96), 30, seed=2)).to(gpu_device)
    dec = PixelBasedDecoder(ArrayCodebook(cb), device=gpu_device)
    vol = dec.decode_pixels_device(img, np.full(16, 400.0), np.zeros(16))
    a = dec.extract_barcodes(vol, 0, 0, zIndex=None)
    b = dec.extract_barcodes(vol, 0, 0, zIndex=3)
    assert len(a) == len(b) > 0
    assert np.isnan(a['z'].to_numpy()).all()
    assert (b['z'] == 3).all()


def test_sharded_fragment_loop_union(gpu_device):
    """run_fovs_sharded: FOV number i on rank i mod world; the union over the
    ranks equals the single-rank run (FOVs are independent, decode.py:71-72)."""
    import torch
    from merlin_amd.decode import Decode
    cb = load_golden('g2_synth192')['codebook']
    fovs = [4, 9, 10, 17, 23]
    stacks = {f: torch.from_numpy(synth_stack(cb, (2, 64, 96), 40, seed=f))
              .to(gpu_device) for f in fovs}
    sf, bg = np.full(16, 400.0), np.zeros(16)
    task = Decode(ArrayCodebook(cb), {'crop_width': 2}, device=gpu_device)
    asked = []

    def provider(f):
        asked.append(f)
        return stacks[f]

    single = {f: bc for f, bc, _ in task.run_fovs_sharded(
        provider, sf, bg, fovs, 0, 1)}
    assert sorted(single) == fovs
    union = {}
    for rank in range(2):
        asked.clear()
        part = {f: bc for f, bc, _ in task.run_fovs_sharded(
            provider, sf, bg, fovs, rank, 2)}
        assert sorted(part) == sorted(asked) == fovs[rank::2]
        union.update(part)
    for f in fovs:
        assert union[f].equals(single[f]) and (single[f]['fov'] == f).all()
