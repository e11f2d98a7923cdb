"""Row-oriented view of the bit matrix held by a storage backend (bigsi/matrix/bitmatrix.py:1-75).

Knows nothing about k-mers.  With the hip-hbm backend the column operations run as device kernels instead of m
read-modify-writes of whole rows (the reference's insert_column, bitmatrix.py:67-75)."""
from ..bitrow import BitRow, row_bytes_of

NUM_ROWS_KEY = "number_of_rows"
NUM_COLS_KEY = "number_of_cols"


class BitMatrix(object):
    def __init__(self, storage):
        self.storage = storage
        self.num_rows = storage.get_integer(NUM_ROWS_KEY)     # KeyError on an empty store, like the reference
        self.num_cols = storage.get_integer(NUM_COLS_KEY)

    @classmethod
    def create(cls, storage, rows, num_rows, num_cols):
        storage.set_integer(NUM_ROWS_KEY, num_rows)           # first, so the device matrix can be sized
        storage.set_bitarrays(range(num_rows), rows)
        storage.set_integer(NUM_COLS_KEY, num_cols)
        storage.sync()
        return cls(storage)

    def get_row(self, row_index):
        return self.storage.get_bitarray(row_index)[: self.num_cols]

    def get_rows(self, row_indexes, remove_trailing_zeros=True):
        rows = self.storage.get_bitarrays(row_indexes)
        return (r[: self.num_cols] for r in rows) if remove_trailing_zeros else rows

    def set_row(self, row_index, bitarray):
        return self.storage.set_bitarray(row_index, bitarray)

    def set_rows(self, row_indexes, bitarrays):
        return self.storage.set_bitarrays(row_indexes, bitarrays)

    def set_num_cols(self, num_cols):
        self.num_cols = num_cols
        self.storage.set_integer(NUM_COLS_KEY, num_cols)

    def get_column(self, column_index):
        if hasattr(self.storage, "get_column"):
            return BitRow.frombytes(self.storage.get_column(column_index), self.num_rows)
        rows = range(self.num_rows)
        return BitRow(list(self.storage.get_bits(list(rows), [column_index] * self.num_rows)))

    def get_columns(self, column_indexes):
        for c in column_indexes:
            yield self.get_column(c)

    def insert_column(self, bitarray, column_index):
        data, nbits = row_bytes_of(bitarray)
        if hasattr(self.storage, "insert_column"):
            self.storage.insert_column(column_index, data)
        else:
            bits = BitRow.frombytes(data, nbits).tolist()
            self.storage.set_bits(list(range(nbits)), [column_index] * nbits, bits)
        if column_index >= self.num_cols:
            self.set_num_cols(self.num_cols + 1)
