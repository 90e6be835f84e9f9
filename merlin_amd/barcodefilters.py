"""z-plane duplicate removal for decoded barcodes: counterpart of
merlin/util/barcodefilters.py:8-107 (used by Decode when
``remove_z_duplicated_barcodes`` is set, analysis/decode.py:170-175, 232-237).

A "next" row of the scope table: a post-filter over the <= 1e5
records of one FOV, host work.  The reference builds a networkx graph per
barcode id with one cKDTree query per pair of neighbouring planes; here one
KD-tree per (barcode id, plane) and scipy's connected_components do the same
grouping, and the brightest barcode of every group is kept.
"""
from typing import List

import numpy as np
import pandas
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components
from scipy.spatial import cKDTree


def remove_zplane_duplicates_single_barcodeid(barcodes: pandas.DataFrame,
                                              zPlanes: int, maxDist: float,
                                              allZPos: List) -> pandas.DataFrame:
    """barcodefilters.py:48-107 for one barcode id."""
    barcodes = barcodes.reset_index(drop=True)
    if not len(barcodes['barcode_id'].unique()) == 1:
        raise ValueError(
            'The method remove_zplane_duplicates_single_barcodeid should be '
            'given a dataframe containing molecules that all have the same '
            'barcode id. Please use remove_zplane_duplicates_all_barcodeids '
            'to handle dataframes containing multiple barcode ids')
    n = len(barcodes)
    z = barcodes['z'].to_numpy()
    xy = barcodes[['x', 'y']].to_numpy(dtype=np.float64)
    nz = len(allZPos)
    members = {k: np.flatnonzero(z == k) for k in range(nz)}
    src, dst = [], []
    for k in range(nz):
        cur = members[k]
        if len(cur) == 0:
            continue
        tree = cKDTree(xy[cur])
        for other in range(max(0, k - zPlanes), min(nz, k + zPlanes + 1)):
            q = members[other]
            if other == k or len(q) == 0:
                continue
            dist, idx = tree.query(xy[q], k=1, distance_upper_bound=maxDist)
            hit = np.isfinite(dist)
            src.append(cur[idx[hit]])
            dst.append(q[hit])
    if src:
        src, dst = np.concatenate(src), np.concatenate(dst)
    else:
        src = dst = np.zeros(0, dtype=np.int64)
    graph = coo_matrix((np.ones(len(src)), (src, dst)), shape=(n, n))
    _, comp = connected_components(graph, directed=False)
    # brightest member of every component (first one on exact ties, as a
    # stable descending sort picks it)
    inten = barcodes['mean_intensity'].to_numpy()
    order = np.lexsort((np.arange(n), -inten, comp))
    firsts = order[np.r_[True, comp[order][1:] != comp[order][:-1]]]
    return barcodes.loc[np.sort(firsts), :]


def remove_zplane_duplicates_all_barcodeids(barcodes: pandas.DataFrame,
                                            zPlanes: int, maxDist: float,
                                            allZPos: List) -> pandas.DataFrame:
    """barcodefilters.py:8-45."""
    if len(barcodes) == 0:
        return barcodes
    kept = [remove_zplane_duplicates_single_barcodeid(g, zPlanes, maxDist,
                                                      allZPos)
            for _, g in barcodes.groupby('barcode_id')]
    merged = pandas.concat(kept, axis=0).reset_index(drop=True)
    return merged.sort_values(by=['barcode_id', 'z'])
