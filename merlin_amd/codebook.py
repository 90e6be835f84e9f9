"""Minimal codebook container for the decode path.

The decoder only needs ``get_barcodes()`` (merlin/util/decoding.py:270); the
accessors mirror merlin/data/codebook.py:91-149 so that a MERlin ``Codebook``
and this class are interchangeable for PixelBasedDecoder.
"""
import itertools

import numpy as np
import pandas


class Codebook(object):

    def __init__(self, barcodes, names=None, bitNames=None):
        self._barcodes = np.asarray(barcodes).astype(np.int64)
        if self._barcodes.ndim != 2:
            raise ValueError('barcodes must be [codewords, bits]')
        n, b = self._barcodes.shape
        self._names = list(names) if names is not None else \
            ['code{}'.format(i) for i in range(n)]
        self._bitNames = list(bitNames) if bitNames is not None else \
            ['bit{}'.format(i + 1) for i in range(b)]

    @classmethod
    def from_csv(cls, path):
        """New-style MERlin codebook csv: name,id,bit1,...,bitB
        (merlin/data/codebook.py:38-41)."""
        df = pandas.read_csv(path)
        bits = [c for c in df.columns if c not in ('name', 'id')]
        return cls(df[bits].to_numpy(), list(df['name']), bits)

    def get_barcodes(self, ignoreBlanks: bool = False) -> np.ndarray:
        if ignoreBlanks:
            keep = ['BLANK' not in n.upper() for n in self._names]
            return self._barcodes[np.array(keep, dtype=bool)]
        return self._barcodes

    def get_barcode(self, index: int) -> np.ndarray:
        return self._barcodes[index]

    def get_barcode_count(self) -> int:
        return self._barcodes.shape[0]

    def get_bit_count(self) -> int:
        return self._barcodes.shape[1]

    def get_bit_names(self):
        return self._bitNames

    def get_name_for_barcode_index(self, index: int) -> str:
        return self._names[index]

    def get_coding_indexes(self):
        return [i for i, n in enumerate(self._names)
                if 'blank' not in n.lower()]

    def get_blank_indexes(self):
        return [i for i, n in enumerate(self._names) if 'blank' in n.lower()]


def constant_weight_codebook(bits: int, weight: int = 4, minDistance: int = 4,
                             count: int = None, seed: int = None) -> Codebook:
    """Constant-weight binary code with the given minimum Hamming distance
    (MHD4 for weight 4 / distance 4), built greedily in lexicographic order;
    with ``seed`` the candidate order is shuffled (stress codebooks)."""
    words = []
    cands = itertools.combinations(range(bits), weight)
    if seed is not None:
        cands = list(cands)
        np.random.default_rng(seed).shuffle(cands)
    share = weight - minDistance // 2       # max on-bits two words may share
    for c in cands:
        s = set(c)
        if all(len(s & w) <= share for w in words):
            words.append(s)
            if count is not None and len(words) >= count:
                break
    mat = np.zeros((len(words), bits), dtype=np.int64)
    for i, w in enumerate(words):
        mat[i, sorted(w)] = 1
    return Codebook(mat)


def random_weight_codebook(bits: int, count: int, weight: int = 4,
                           seed: int = 0) -> Codebook:
    """Distinct random codewords of fixed weight, no distance guarantee
    (SURVEY 8d config 5 stress)."""
    from math import comb
    if not 0 < weight <= bits or count > comb(bits, weight):
        raise ValueError('no {} distinct codewords of weight {} in {} bits'.format(
            count, weight, bits))
    rng = np.random.default_rng(seed)
    seen = set()
    while len(seen) < count:
        seen.add(tuple(sorted(rng.choice(bits, weight, replace=False))))
    mat = np.zeros((count, bits), dtype=np.int64)
    for i, w in enumerate(sorted(seen)):
        mat[i, list(w)] = 1
    return Codebook(mat)
