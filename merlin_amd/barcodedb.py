"""Barcode record sink: counterpart of merlin/util/barcodedb.py:46-65, 160-266.

The reference appends each FOV's barcode DataFrame to a per-FOV PyTables HDF5
store after casting it to a fixed column-type map (``_get_bc_column_types``).
Here the same typed table is kept per FOV in memory and, when a directory is
given, as one Parquet file per FOV (pyarrow; PyTables is not required).  The
method names, argument meaning and empty-result behaviour follow the
reference's ``BarcodeDB`` / ``PyTablesBarcodeDB`` so that ``Decode``,
``FilterBarcodes`` and the adaptive filter can use it unchanged.
"""
import os
from typing import Dict, List, Optional

import numpy as np
import pandas


def bc_column_types(bitCount: int) -> Dict[str, type]:
    """barcodedb.py:46-65"""
    columnInformation = {'barcode_id': np.uint16,
                         'fov': np.uint16,
                         'mean_intensity': np.float32,
                         'max_intensity': np.float32,
                         'area': np.uint16,
                         'mean_distance': np.float32,
                         'min_distance': np.float32,
                         'x': np.float32,
                         'y': np.float32,
                         'z': np.float32,
                         'global_x': np.float32,
                         'global_y': np.float32,
                         'global_z': np.float32,
                         'cell_index': np.int32}
    for i in range(bitCount):
        columnInformation['intensity_' + str(i)] = np.float32
    return columnInformation


class BarcodeDB(object):

    def __init__(self, codebook, fovs: Optional[List[int]] = None,
                 directory: Optional[str] = None):
        """codebook: anything with get_bit_count(); fovs: the data set's FOV
        list (grown on write when not given); directory: where the per-FOV
        Parquet files live (None = memory only)."""
        self._codebook = codebook
        self._fovs = list(fovs) if fovs is not None else []
        self._dir = directory
        self._mem: Dict[int, pandas.DataFrame] = {}
        if directory is not None:
            os.makedirs(directory, exist_ok=True)

    def _get_bc_column_types(self):
        return bc_column_types(self._codebook.get_bit_count())

    def _path(self, fov):
        return os.path.join(self._dir, 'barcode_data_{}.parquet'.format(int(fov)))

    def _known_fovs(self):
        fovs = set(self._fovs) | set(self._mem.keys())
        if self._dir is not None:
            for f in os.listdir(self._dir):
                if f.startswith('barcode_data_') and f.endswith('.parquet'):
                    fovs.add(int(f[len('barcode_data_'):-len('.parquet')]))
        return sorted(fovs)

    # ------------------------------------------------------------ writing --

    def empty_database(self, fov: int = None) -> None:
        """barcodedb.py:164-170"""
        if fov is None:
            for f in self._known_fovs():
                self.empty_database(f)
            return
        self._mem.pop(int(fov), None)
        if self._dir is not None and os.path.exists(self._path(fov)):
            os.remove(self._path(fov))

    def write_barcodes(self, barcodeInformation: pandas.DataFrame,
                       fov: int = None) -> None:
        """barcodedb.py:250-266: cast to the column-type map and append to the
        FOV's table (extra columns are kept as they are)."""
        if len(barcodeInformation) <= 0:
            return
        if fov is None:
            for f in barcodeInformation.fov.unique():
                self.write_barcodes(
                    barcodeInformation.loc[barcodeInformation['fov'] == f],
                    fov=f)
            return
        fov = int(fov)
        typed = barcodeInformation.astype(self._get_bc_column_types())
        old = self._load(fov)
        if old is not None and len(old):
            typed = pandas.concat([old, typed], sort=False)
        if self._dir is not None:
            typed.reset_index(drop=True).to_parquet(self._path(fov))
            self._mem.pop(fov, None)
        else:
            self._mem[fov] = typed
        if fov not in self._fovs:
            self._fovs.append(fov)

    # ------------------------------------------------------------ reading --

    def _load(self, fov, columnList=None):
        fov = int(fov)
        if fov in self._mem:
            df = self._mem[fov]
            return df if columnList is None else df[columnList]
        if self._dir is not None and os.path.exists(self._path(fov)):
            return pandas.read_parquet(self._path(fov), columns=columnList)
        return None

    def get_barcodes(self, fov: int = None, columnList: List[str] = None,
                     chunksize: int = None) -> pandas.DataFrame:
        """barcodedb.py:172-207"""
        if fov is None:
            parts = [self.get_barcodes(fov=x, columnList=columnList)
                     for x in self._known_fovs()]
            parts = [p for p in parts if len(p)]
            barcodes = pandas.concat(parts, sort=False) if parts \
                else pandas.DataFrame()
        else:
            barcodes = self._load(fov, columnList)
            if barcodes is None:
                barcodes = pandas.DataFrame()
        if len(barcodes) == 0:
            if columnList:
                barcodes = pandas.DataFrame(columns=columnList)
            else:
                barcodes = pandas.DataFrame(
                    columns=self._get_bc_column_types().keys())
        return barcodes

    def get_filtered_barcodes(
            self, areaThreshold: int, intensityThreshold: float,
            distanceThreshold: float = None, fov: int = None,
            chunksize: int = None) -> pandas.DataFrame:
        """barcodedb.py:209-223"""
        allBarcodes = self.get_barcodes(fov)
        if distanceThreshold is None:
            return allBarcodes[
                (allBarcodes['area'] >= areaThreshold)
                & (allBarcodes['mean_intensity'] >= intensityThreshold)]
        return allBarcodes[
            (allBarcodes['area'] >= areaThreshold)
            & (allBarcodes['mean_intensity'] >= intensityThreshold)
            & (allBarcodes['min_distance'] <= distanceThreshold)]

    def get_intensities_for_barcodes_with_area(
            self, area: int) -> pandas.Series:
        """barcodedb.py:245-248"""
        allBarcodes = self.get_barcodes(columnList=['area', 'mean_intensity'])
        return allBarcodes[allBarcodes['area'] == area]['mean_intensity']

    def get_barcode_intensities(self) -> pandas.Series:
        return self.get_barcodes(
            columnList=['mean_intensity'])['mean_intensity']

    def get_barcode_areas(self) -> pandas.Series:
        return self.get_barcodes(columnList=['area'])['area']

    def get_barcode_distances(self) -> pandas.Series:
        return self.get_barcodes(columnList=['mean_distance'])['mean_distance']
