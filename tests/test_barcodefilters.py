"""CPU: z-plane duplicate removal against the reference's own known-answer
cases (test/test_zplane_duplicate_removal.py:41-123; the barcodes below are
that file's fixture data)."""
import random

import numpy as np
import pandas as pd

from merlin_amd import barcodefilters


def generate_barcode(fov, barcode_id, x, y, z, mean_intensity):
    bc = {'barcode': random.getrandbits(32), 'barcode_id': barcode_id,
          'fov': fov, 'mean_intensity': mean_intensity,
          'max_intensity': random.uniform(5, 15),
          'area': random.randint(0, 10), 'mean_distance': random.random(),
          'min_distance': random.random(), 'x': x, 'y': y, 'z': z,
          'global_x': random.uniform(0, 200000),
          'global_y': random.uniform(0, 200000),
          'global_z': random.uniform(0, 5),
          'cell_index': random.randint(0, 5000)}
    for i in range(16):
        bc['intensity_' + str(i)] = random.uniform(5, 15)
    return bc


b1 = generate_barcode(100, 5, 402.21, 787.11, 2, 14.23)
b2 = generate_barcode(100, 5, 502.21, 687.11, 3, 12.23)
b3 = generate_barcode(100, 17, 402.21, 787.11, 2, 10.23)
b1_above_dimmer = generate_barcode(100, 5, 402.21, 787.11, 3, 11.23)
b1_closeby_above_brighter = generate_barcode(100, 5, 403.21, 787.11, 3, 15.23)
b2_above_brighter = generate_barcode(100, 5, 502.31, 687.11, 4, 14.23)
b1_closeby_below_brighter = generate_barcode(100, 5, 403.21, 787.11, 1, 15.0)
b1_closeby_toofar_brighter = generate_barcode(100, 5, 403.21, 787.11, 0, 15.0)
ZPOS = [0, 1.5, 3, 4.5, 6, 7.5, 9]


def _kept(bcs, zplanes=1, xy=np.sqrt(2)):
    out = barcodefilters.remove_zplane_duplicates_all_barcodeids(
        pd.DataFrame(bcs), zplanes, xy, ZPOS)
    return set(out['barcode'].values), out


def test_multiple_comparisons_barcodes():
    kept, _ = _kept([b1, b2, b3, b1_above_dimmer, b1_closeby_above_brighter,
                     b2_above_brighter, b1_closeby_below_brighter,
                     b1_closeby_toofar_brighter])
    for ex in (b1_closeby_above_brighter, b2_above_brighter, b3):
        assert ex['barcode'] in kept
    for ne in (b1, b2, b1_above_dimmer, b1_closeby_below_brighter,
               b1_closeby_toofar_brighter):
        assert ne['barcode'] not in kept


def test_all_compatible_barcodes():
    bcs = [b1, b2, b3, b1_closeby_toofar_brighter]
    kept, out = _kept(bcs)
    assert kept == {x['barcode'] for x in bcs} and len(out) == len(bcs)
    assert list(out['barcode_id']) == sorted(out['barcode_id'])


def test_farther_zrange():
    kept, _ = _kept([b1, b2, b3, b1_closeby_toofar_brighter], zplanes=2)
    assert kept == {x['barcode'] for x in
                    (b2, b3, b1_closeby_toofar_brighter)}


def test_farther_xyrange():
    kept, _ = _kept([b1, b2, b3], xy=np.sqrt(20001))
    assert kept == {b1['barcode'], b3['barcode']}


def test_empty_and_mixed_id_errors():
    empty = pd.DataFrame([b1]).iloc[0:0]
    assert len(barcodefilters.remove_zplane_duplicates_all_barcodeids(
        empty, 1, 1.5, ZPOS)) == 0
    import pytest
    with pytest.raises(ValueError):
        barcodefilters.remove_zplane_duplicates_single_barcodeid(
            pd.DataFrame([b1, b3]), 1, 1.5, ZPOS)
