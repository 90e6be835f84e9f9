"""Barcode record sink: counterpart of merlin/util/barcodedb.py.

The reference appends each FOV's barcode DataFrame to a per-FOV PyTables HDF5
store after casting it to a fixed column-type map (``_get_bc_column_types``,
barcodedb.py:46-65; the cast is write_barcodes' ``astype``, :264-266).  Here a
FOV's records live as one typed column table: either cast from a DataFrame the
way the reference does, or taken as they come from the GPU
(``TypedBarcodes``, filled by ``mf_region_frame_typed`` -- the kernel already
wrote uint16 / float32 / int32 columns, so there is no host-side cast).  Tables
are kept in memory and, when a directory is given, as one Parquet file per FOV
(pyarrow; PyTables is not required).

The public methods keep the names, arguments and empty-result behaviour of the
reference's ``BarcodeDB`` / ``PyTablesBarcodeDB`` (barcodedb.py:67-266), which
``Decode``, ``FilterBarcodes`` and the adaptive filter call.
"""
import os
from typing import Dict, Iterable, List, Optional

import numpy as np
import pandas

_FIXED_COLUMNS = (
    ('barcode_id', np.uint16), ('fov', np.uint16),
    ('mean_intensity', np.float32), ('max_intensity', np.float32),
    ('area', np.uint16), ('mean_distance', np.float32),
    ('min_distance', np.float32), ('x', np.float32), ('y', np.float32),
    ('z', np.float32), ('global_x', np.float32), ('global_y', np.float32),
    ('global_z', np.float32), ('cell_index', np.int32))


def bc_column_types(bitCount: int) -> Dict[str, type]:
    """The database schema (barcodedb.py:46-65): column -> numpy dtype."""
    schema = dict(_FIXED_COLUMNS)
    schema.update(('intensity_{}'.format(i), np.float32)
                  for i in range(bitCount))
    return schema


class TypedBarcodes(object):
    """One FOV's records already in the schema's dtypes: ``columns`` maps every
    schema column to a 1-D array of its dtype, ``index`` is the DataFrame index
    the reference would carry (position inside the per-barcode frame)."""

    def __init__(self, columns: Dict[str, np.ndarray], index: np.ndarray):
        self.columns = columns
        self.index = index

    def __len__(self):
        return len(self.index)

    def check(self, schema: Dict[str, type]) -> None:
        for name, dt in schema.items():
            got = self.columns[name]
            if got.dtype != np.dtype(dt) or got.shape != self.index.shape:
                raise TypeError('column {} is {}{}, the schema wants {}[{}]'
                                .format(name, got.dtype, got.shape,
                                        np.dtype(dt), len(self.index)))

    def to_frame(self, copy: bool = False) -> pandas.DataFrame:
        """The records as a DataFrame.  The columns of a GPU hand-over are
        views of page-locked staging buffers sized for every region of the
        FOV, not just the kept rows: ``copy=True`` moves the kept rows into
        ordinary numpy memory (what a table that outlives the FOV must hold,
        or pinned memory grows with the FOV count)."""
        if copy:
            return pandas.DataFrame(
                {k: np.array(v) for k, v in self.columns.items()},
                index=np.array(self.index), copy=False)
        return pandas.DataFrame(self.columns, index=self.index, copy=False)


class BarcodeDB(object):

    def __init__(self, codebook, fovs: Optional[List[int]] = None,
                 directory: Optional[str] = None):
        """codebook: anything with get_bit_count(); fovs: the data set's FOV
        list (grown on write when not given); directory: where the per-FOV
        Parquet files live (None = memory only)."""
        self._codebook = codebook
        self._fovs = list(fovs) if fovs is not None else []
        self._dir = directory
        self._tables: Dict[int, pandas.DataFrame] = {}
        if directory is not None:
            os.makedirs(directory, exist_ok=True)

    def _get_bc_column_types(self):
        return bc_column_types(self._codebook.get_bit_count())

    # ------------------------------------------------------------- storage --

    def _file(self, fov: int) -> str:
        return os.path.join(self._dir, 'barcode_data_{}.parquet'.format(fov))

    def _stored_fovs(self) -> List[int]:
        found = set(self._fovs) | set(self._tables)
        if self._dir is not None:
            for name in os.listdir(self._dir):
                stem, ext = os.path.splitext(name)
                if ext == '.parquet' and stem.startswith('barcode_data_'):
                    found.add(int(stem[len('barcode_data_'):]))
        return sorted(found)

    def _read_table(self, fov: int, columns=None) -> Optional[pandas.DataFrame]:
        if fov in self._tables:
            table = self._tables[fov]
            return table if columns is None else table[columns]
        if self._dir is not None and os.path.exists(self._file(fov)):
            return pandas.read_parquet(self._file(fov), columns=columns)
        return None

    def _append_table(self, fov: int, typed: pandas.DataFrame) -> None:
        existing = self._read_table(fov)
        if existing is not None and len(existing):
            typed = pandas.concat([existing, typed], sort=False)
        if self._dir is None:
            self._tables[fov] = typed
        else:
            typed.reset_index(drop=True).to_parquet(self._file(fov))
            self._tables.pop(fov, None)
        if fov not in self._fovs:
            self._fovs.append(fov)

    def _empty_result(self, columns: Optional[Iterable[str]]):
        return pandas.DataFrame(
            columns=list(columns) if columns
            else list(self._get_bc_column_types()))

    # ------------------------------------------------------------- writing --

    def empty_database(self, fov: int = None) -> None:
        """Drop one FOV's records, or every FOV's (barcodedb.py:185-191)."""
        targets = self._stored_fovs() if fov is None else [int(fov)]
        for f in targets:
            self._tables.pop(f, None)
            if self._dir is not None and os.path.exists(self._file(f)):
                os.remove(self._file(f))

    def write_barcodes(self, barcodeInformation, fov: int = None) -> None:
        """Append records (barcodedb.py:250-266).  A DataFrame is cast to the
        schema as the reference does (extra columns stay as they are); a
        ``TypedBarcodes`` from the GPU is stored as it is.  Without ``fov``
        the rows are split by their fov column."""
        if len(barcodeInformation) == 0:
            return
        if isinstance(barcodeInformation, TypedBarcodes):
            if fov is None:
                raise ValueError('typed records are written one FOV at a time')
            barcodeInformation.check(self._get_bc_column_types())
            # zero-copy only on the way into a Parquet file; a table kept in
            # memory owns ordinary memory, not the FOV's pinned staging buffers
            self._append_table(int(fov), barcodeInformation.to_frame(
                copy=self._dir is None))
            return
        if fov is None:
            for f, rows in barcodeInformation.groupby('fov', sort=False):
                self.write_barcodes(rows, fov=f)
            return
        self._append_table(
            int(fov), barcodeInformation.astype(self._get_bc_column_types()))

    # ------------------------------------------------------------- reading --

    def get_barcodes(self, fov: int = None, columnList: List[str] = None,
                     chunksize: int = None) -> pandas.DataFrame:
        """One FOV's table, or every FOV's stacked in FOV order; a frame with
        the right columns and no rows when nothing is stored
        (barcodedb.py:193-227)."""
        if fov is not None:
            found = self._read_table(int(fov), columnList)
            tables = [] if found is None else [found]
        else:
            tables = [t for t in (self._read_table(f, columnList)
                                  for f in self._stored_fovs())
                      if t is not None]
        tables = [t for t in tables if len(t)]
        if not tables:
            return self._empty_result(columnList)
        return tables[0] if len(tables) == 1 \
            else pandas.concat(tables, sort=False)

    def get_filtered_barcodes(
            self, areaThreshold: int, intensityThreshold: float,
            distanceThreshold: float = None, fov: int = None,
            chunksize: int = None) -> pandas.DataFrame:
        """Rows with area >= areaThreshold, mean_intensity >=
        intensityThreshold and, when given, min_distance <= distanceThreshold
        (barcodedb.py:229-243)."""
        table = self.get_barcodes(fov)
        passing = (table['area'] >= areaThreshold) \
            & (table['mean_intensity'] >= intensityThreshold)
        if distanceThreshold is not None:
            passing = passing & (table['min_distance'] <= distanceThreshold)
        return table[passing]

    def _column(self, name: str) -> pandas.Series:
        return self.get_barcodes(columnList=[name])[name]

    def get_intensities_for_barcodes_with_area(
            self, area: int) -> pandas.Series:
        """barcodedb.py:245-248"""
        pair = self.get_barcodes(columnList=['area', 'mean_intensity'])
        return pair.loc[pair['area'] == area, 'mean_intensity']

    def get_barcode_intensities(self) -> pandas.Series:
        return self._column('mean_intensity')

    def get_barcode_areas(self) -> pandas.Series:
        return self._column('area')

    def get_barcode_distances(self) -> pandas.Series:
        return self._column('mean_distance')
