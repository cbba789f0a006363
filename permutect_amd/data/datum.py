"""Field offsets of the per-variant int16 / float16 arrays and a minimal `Datum` container.

Only the layout constants the hot path reads are mirrored here (reference permutect/data/datum.py:51-89 and :35);
the text/VCF construction paths of the reference Datum are out of scope.
"""
from __future__ import annotations

import enum

import numpy as np

NUMBER_OF_BYTES_IN_PACKED_READ = 7  # bit-packed binary read features, MSB first (np.packbits)


class Data(enum.Enum):
    # int array columns
    REF_COUNT = ("int", 0)
    ALT_COUNT = ("int", 1)
    LABEL = ("int", 2)
    VARIANT_TYPE = ("int", 3)
    SOURCE = ("int", 4)
    ORIGINAL_DEPTH = ("int", 5)
    ORIGINAL_ALT_COUNT = ("int", 6)
    ORIGINAL_NORMAL_DEPTH = ("int", 7)
    ORIGINAL_NORMAL_ALT_COUNT = ("int", 8)
    CONTIG = ("int", 9)
    # float array columns
    SEQ_ERROR_LOG_LK = ("float", 0)
    NORMAL_SEQ_ERROR_LOG_LK = ("float", 1)
    ALLELE_FREQUENCY = ("float", 2)
    MAF = ("float", 3)
    NORMAL_MAF = ("float", 4)
    CACHED_ARTIFACT_LOGIT = ("float", 5)

    def __init__(self, kind: str, idx: int):
        self.kind = kind
        self.idx = idx


NUM_SCALAR_INT_ELEMENTS = 16
HAPLOTYPES_START_IDX = 16
NUM_SCALAR_FLOAT_ELEMENTS = 6
INFO_START_IDX = 6


class Datum:
    """One variant: int16 array [16 + H], float16 array [6 + I], reads uint8 [(n_ref + n_alt), 7 + nf] (ref rows
    first) or float16 [(n_ref + n_alt), F]."""

    def __init__(self, int_array: np.ndarray, float_array: np.ndarray, reads_re: np.ndarray, compressed: bool = True):
        assert int_array.ndim == 1 and len(int_array) >= NUM_SCALAR_INT_ELEMENTS
        assert float_array.ndim == 1 and len(float_array) >= NUM_SCALAR_FLOAT_ELEMENTS
        self.int_array = int_array.astype(np.int16)
        self.float_array = float_array.astype(np.float16)
        self.reads_re = reads_re
        assert reads_re.dtype == (np.uint8 if compressed else np.float16)

    def get(self, field: Data):
        return (self.int_array if field.kind == "int" else self.float_array)[field.idx]

    def set(self, field: Data, value):
        (self.int_array if field.kind == "int" else self.float_array)[field.idx] = value

    def get_int_array(self):
        return self.int_array

    def get_float_array(self):
        return self.float_array

    def get_ref_reads_re(self):
        return self.reads_re[: -int(self.get(Data.ALT_COUNT))]

    def get_alt_reads_re(self):
        return self.reads_re[-int(self.get(Data.ALT_COUNT)):]

    def num_read_features(self) -> int:
        if self.reads_re.dtype == np.uint8:
            return 8 * NUMBER_OF_BYTES_IN_PACKED_READ + self.reads_re.shape[1] - NUMBER_OF_BYTES_IN_PACKED_READ
        return self.reads_re.shape[1]
