"""`BitMatrix`: the m x N bit matrix of an index as the layers above the storage contract see it.

API-compatible with the reference's class of the same name (bigsi/matrix/bitmatrix.py:1-75) -- rows in and out as
bit rows, the two dimension records, column read / insert -- but written for a backend whose matrix is device
resident: dimensions are read through properties, column operations go to `storage.insert_column` / `get_column`
(one kernel over all m rows) when the backend has them, and only fall back to the per-row read-modify-write loop of
the generic contract (`set_bits`, bigsi/storage/base.py:111-122) for backends that do not.
"""
from ..bitrow import BitRow, row_bytes_of

NUM_ROWS_KEY, NUM_COLS_KEY = "number_of_rows", "number_of_cols"


def _trim(row, width):
    return row[:width]


class BitMatrix(object):
    def __init__(self, storage):
        self.storage = storage
        # both records must exist: opening an empty store is an error, as in the reference (bitmatrix.py:16-17)
        self._rows = storage.get_integer(NUM_ROWS_KEY)
        self._cols = storage.get_integer(NUM_COLS_KEY)

    # ---- dimensions
    @property
    def num_rows(self):
        return self._rows

    @property
    def num_cols(self):
        return self._cols

    @num_cols.setter
    def num_cols(self, value):
        self.set_num_cols(value)

    def set_num_cols(self, num_cols):
        self._cols = int(num_cols)
        self.storage.set_integer(NUM_COLS_KEY, self._cols)

    @classmethod
    def create(cls, storage, rows, num_rows, num_cols):
        """Store `rows` (an iterable of num_rows bit rows) and the two dimension records; returns the opened matrix.
        The row count is written first so that a device backend can size its matrix before the rows arrive."""
        storage.set_integer(NUM_ROWS_KEY, int(num_rows))
        storage.set_bitarrays(range(num_rows), rows)
        storage.set_integer(NUM_COLS_KEY, int(num_cols))
        storage.sync()
        return cls(storage)

    # ---- rows
    def get_row(self, row_index):
        return _trim(self.storage.get_bitarray(row_index), self._cols)

    def get_rows(self, row_indexes, remove_trailing_zeros=True):
        """Generator over the requested rows; `remove_trailing_zeros` cuts the byte padding off (bitmatrix.py:30-37)."""
        fetched = self.storage.get_bitarrays(row_indexes)
        if not remove_trailing_zeros:
            return fetched
        width = self._cols
        return (_trim(r, width) for r in fetched)

    def set_row(self, row_index, bitarray):
        self.storage.set_bitarray(row_index, bitarray)

    def set_rows(self, row_indexes, bitarrays):
        self.storage.set_bitarrays(row_indexes, bitarrays)

    # ---- columns
    def get_column(self, column_index):
        reader = getattr(self.storage, "get_column", None)
        if reader is not None:
            return BitRow.frombytes(reader(column_index), self._rows)
        every_row = list(range(self._rows))
        return BitRow(list(self.storage.get_bits(every_row, [column_index] * self._rows)))

    def get_columns(self, column_indexes):
        return (self.get_column(c) for c in column_indexes)

    def insert_column(self, bitarray, column_index):
        """Write one sample's Bloom filter into column `column_index`; appending (index == num_cols) grows the matrix."""
        data, nbits = row_bytes_of(bitarray)
        writer = getattr(self.storage, "insert_column", None)
        if writer is not None:
            writer(column_index, data)
        else:
            bits = BitRow.frombytes(data, nbits).tolist()
            self.storage.set_bits(list(range(nbits)), [column_index] * nbits, bits)
        if column_index >= self._cols:
            self.set_num_cols(self._cols + 1)
