"""CPU: the barcode record sink (merlin_amd/barcodedb.py) with the cases of the
reference's test/test_barcode_database.py, and barcode filtering
(merlin_amd/filterbarcodes.py) against goldens produced by the reference's own
filterbarcodes.py (oracle/make_filter_goldens.py), incl. the world-size-2
histogram all-reduce over gloo."""
import os
import random
import socket

import numpy as np
import pandas
import pytest
import torch.multiprocessing as mp

from conftest import GOLDEN_DIR
from merlin_amd import barcodedb, filterbarcodes
from merlin_amd.codebook import Codebook


class Bits16:
    def get_bit_count(self):
        return 16


def generate_random_barcode(fov):
    """test_barcode_database.py:30-50"""
    b = {'barcode': random.getrandbits(32), 'barcode_id': random.randint(0, 200),
         'fov': fov, 'mean_intensity': random.uniform(5, 15),
         'max_intensity': random.uniform(5, 15), 'area': random.randint(0, 10),
         'mean_distance': random.random(), 'min_distance': random.random(),
         'x': random.uniform(0, 2048), 'y': random.uniform(0, 2048),
         'z': random.uniform(0, 5), 'global_x': random.uniform(0, 200000),
         'global_y': random.uniform(0, 200000), 'global_z': random.uniform(0, 5),
         'cell_index': random.randint(0, 5000)}
    for i in range(16):
        b['intensity_' + str(i)] = random.uniform(5, 15)
    return b


# the two known records of test_barcode_database.py:53-115
barcode1 = dict(barcode=290, barcode_id=0, fov=0, mean_intensity=5.0, max_intensity=7.0,
                area=5, mean_distance=0.1, min_distance=0.05, x=10, y=5, z=15, global_x=87,
                global_y=29, global_z=14, cell_index=8,
                **{'intensity_%d' % i: (54 if i == 15 else 89) for i in range(16)})
barcode2 = dict(barcode=390, barcode_id=1, fov=0, mean_intensity=5.2, max_intensity=7.2,
                area=4, mean_distance=0.2, min_distance=0.07, x=11, y=6, z=12, global_x=81,
                global_y=28, global_z=15, cell_index=7,
                **{'intensity_%d' % i: v for i, v in enumerate(
                    [88, 88, 28, 38, 48, 58, 68, 78, 97, 17, 27, 37, 47, 57, 67, 77])})


@pytest.fixture(params=['memory', 'parquet'])
def barcode_db(request, tmp_path):
    random.seed(7)
    d = None if request.param == 'memory' else str(tmp_path / 'bc')
    db = barcodedb.BarcodeDB(Bits16(), fovs=[0, 1], directory=d)
    yield db
    db.empty_database()
    assert len(db.get_barcodes()) == 0


@pytest.fixture
def barcode_db_with_barcodes(barcode_db):
    written = pandas.concat([
        pandas.DataFrame([generate_random_barcode(0) for _ in range(20)]),
        pandas.DataFrame([generate_random_barcode(1) for _ in range(20)])])
    barcode_db.write_barcodes(written)
    return barcode_db, written


def _same(a, b):
    return np.isclose(a.values.astype(float), b.values.astype(float)).all()


def test_column_types():
    t = barcodedb.bc_column_types(3)
    assert list(t)[:14] == ['barcode_id', 'fov', 'mean_intensity', 'max_intensity', 'area',
                            'mean_distance', 'min_distance', 'x', 'y', 'z', 'global_x',
                            'global_y', 'global_z', 'cell_index']
    assert t['barcode_id'] is np.uint16 and t['area'] is np.uint16
    assert t['cell_index'] is np.int32 and t['x'] is np.float32
    assert list(t)[14:] == ['intensity_0', 'intensity_1', 'intensity_2']


def test_write_and_read_one_fov(barcode_db):
    assert len(barcode_db.get_barcodes()) == 0
    assert list(barcode_db.get_barcodes().columns) == list(barcodedb.bc_column_types(16))
    written = pandas.DataFrame([barcode1, barcode2])
    barcode_db.write_barcodes(written, fov=0)
    read = barcode_db.get_barcodes()[written.columns]
    assert _same(written, read)
    assert read['barcode_id'].dtype == np.uint16 and read['x'].dtype == np.float32
    barcode_db.empty_database(fov=0)
    assert len(barcode_db.get_barcodes()) == 0


def test_multiple_write_one_fov(barcode_db):
    sets = [pandas.DataFrame([generate_random_barcode(0) for _ in range(10)]) for _ in range(3)]
    for s in sets:
        barcode_db.write_barcodes(s, fov=0)
    read = barcode_db.get_barcodes()[sets[0].columns]
    combined = pandas.concat(sets)
    by = list(read.columns)[1:]
    assert _same(read.sort_values(by=by), combined.sort_values(by=by))


def test_write_and_read_multiple_fov(barcode_db):
    combined = pandas.concat([
        pandas.DataFrame([generate_random_barcode(0) for _ in range(10)]),
        pandas.DataFrame([generate_random_barcode(1) for _ in range(10)])])
    barcode_db.write_barcodes(combined)
    read = barcode_db.get_barcodes()[combined.columns]
    by = list(read.columns)[1:]
    assert _same(read.sort_values(by=by), combined.sort_values(by=by))
    assert len(barcode_db.get_barcodes(fov=1)) == 10
    assert set(barcode_db.get_barcodes(fov=1)['fov']) == {1}


def test_read_select_columns(barcode_db_with_barcodes):
    db, written = barcode_db_with_barcodes
    cols = ['mean_intensity', 'x', 'intensity_0']
    assert np.isclose(written[cols].values, db.get_barcodes(columnList=cols)[cols].values).all()
    assert list(db.get_barcodes(fov=5, columnList=cols).columns) == cols


def test_read_filtered_barcodes(barcode_db_with_barcodes):
    db, written = barcode_db_with_barcodes
    for area in range(0, 11, 2):
        for intensity in np.arange(0, 20, 5.1):
            read = db.get_filtered_barcodes(area, intensity)
            want = written[(written['area'] >= area) & (written['mean_intensity'] >= intensity)]
            assert len(read) == len(want)
            if len(read):
                by = list(read.columns)[1:]
                assert np.isclose(read.sort_values(by=by).values.astype(float),
                                  want.sort_values(by=by)[read.columns].values.astype(float)).all()
    # distance threshold and the FilterBarcodes defaults
    read = db.get_filtered_barcodes(0, 0, distanceThreshold=0.5, fov=0)
    want = written[(written['fov'] == 0) & (written['min_distance'].astype(np.float32) <= 0.5)]
    assert len(read) == len(want)
    assert len(filterbarcodes.filter_barcodes(db, fov=0)) == 0      # intensity >= 200: none


def test_get_barcode_values(barcode_db_with_barcodes):
    db, written = barcode_db_with_barcodes
    for area in range(11):
        assert np.isclose(sorted(db.get_intensities_for_barcodes_with_area(area)),
                          sorted(written[written['area'] == area]['mean_intensity'])).all()
    for fn, col in (('get_barcode_intensities', 'mean_intensity'), ('get_barcode_areas', 'area'),
                    ('get_barcode_distances', 'mean_distance')):
        assert np.isclose(sorted(getattr(db, fn)()), sorted(written[col])).all()


# ------------------------------------------------------------ adaptive filter --

def _golden():
    g = np.load(os.path.join(GOLDEN_DIR, 'fb_adaptive.npz'))
    names = ['code%d' % i for i in range(int(g['n_codes']))]
    for b in g['blanks']:
        names[int(b)] = 'Blank-%d' % b
    cb = Codebook(np.zeros((len(names), 16), dtype=np.int64), names)
    df = pandas.DataFrame({k: g[k] for k in ('barcode_id', 'fov', 'mean_intensity',
                                             'min_distance', 'area')})
    return g, cb, df


def _threshold_task(g, cb, df, fovs):
    t = filterbarcodes.AdaptiveThreshold(cb, distance_threshold=0.5167)
    t.set_intensity_bins_from_maximum(float(df['mean_intensity'].max()))
    for f in fovs:
        t.add_barcodes(df[df['fov'] == f])
    return t


def test_adaptive_threshold_matches_reference():
    g, cb, df = _golden()
    t = _threshold_task(g, cb, df, [0, 1, 2])
    assert np.array_equal(t.get_area_bins(), g['area_bins'])
    assert np.array_equal(t.get_distance_bins(), g['distance_bins'])
    assert np.array_equal(t.get_intensity_bins(), g['intensity_bins'])
    assert np.array_equal(t.get_blank_count_histogram(), g['blank_counts'])
    assert np.array_equal(t.get_coding_count_histogram(), g['coding_counts'])
    assert np.array_equal(t.get_blank_fraction_histogram(), g['fraction'])
    for p, r, c in zip(g['probe'], g['rates'], g['counts']):
        assert t.calculate_misidentification_rate_for_threshold(p) == r
        assert t.calculate_barcode_count_for_threshold(p) == c
    for target, thr, kept in zip(g['targets'], g['thresholds'], g['kept']):
        got = t.calculate_threshold_for_misidentification_rate(target)
        assert got == thr
        sel = t.extract_barcodes_with_threshold(got, df)
        assert np.array_equal(np.isin(np.arange(len(df)), sel.index), kept)
        assert sel.equals(t.adaptive_filter(df, misidentification_rate=target))
    edge = pandas.DataFrame({'barcode_id': g['edge_barcode_id'],
                             'mean_intensity': g['edge_mean_intensity'],
                             'min_distance': g['edge_min_distance'], 'area': g['edge_area']})
    sel = t.extract_barcodes_with_threshold(float(g['thresholds'][0]), edge)
    assert np.array_equal(np.isin(np.arange(len(edge)), sel.index), g['edge_kept'])


def test_count_cube_is_histogramdd():
    g, cb, df = _golden()
    t = _threshold_task(g, cb, df, [])
    rng = np.random.default_rng(0)
    pts = np.stack([rng.uniform(-0.5, 7, 5000), rng.uniform(-0.05, 0.6, 5000),
                    rng.integers(0, 40, 5000).astype(float)], axis=1)
    pts[:50, 0] = t.get_intensity_bins()[-1]          # exactly on the closing edges
    pts[50:100, 1] = t.get_distance_bins()[-1]
    pts[100:150, 2] = t.get_area_bins()[-1]
    pts[150:200, 0] = t.get_intensity_bins()[3]
    want = np.histogramdd(pts, bins=(t.get_intensity_bins(), t.get_distance_bins(),
                                     t.get_area_bins()))[0]
    assert np.array_equal(t._count_cube(pts), want)
    assert t._count_cube(pts[:0]).sum() == 0


def test_adaptive_threshold_needs_bins():
    g, cb, df = _golden()
    t = filterbarcodes.AdaptiveThreshold(cb)
    with pytest.raises(ValueError):
        t.add_barcodes(df[:10])


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g, cb, df = _golden()
        t = _threshold_task(g, cb, df, [f for f in range(3) if f % world == rank])
        t.all_reduce()
        q.put((rank, t.get_blank_count_histogram(), t.get_coding_count_histogram(),
               t.calculate_threshold_for_misidentification_rate(0.05)))
    finally:
        dist.destroy_process_group()


def test_gloo_world2_histogram_all_reduce():
    """FOVs sharded over two ranks, count histograms summed by all-reduce:
    every rank ends with the reference's histograms and threshold."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = np.load(os.path.join(GOLDEN_DIR, 'fb_adaptive.npz'))
    for _, blank, coding, thr in res:
        assert np.array_equal(blank, g['blank_counts'])
        assert np.array_equal(coding, g['coding_counts'])
        assert thr == g['thresholds'][0]
