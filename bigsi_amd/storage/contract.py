"""Typed key/value contract every BIGSI storage backend satisfies.

This is the reference's plugin boundary (bigsi/storage/base.py:9-151) expressed over two primitives a backend
supplies -- `_get_raw(key: bytes) -> bytes` (KeyError on a miss) and `_put_raw(key: bytes, value: bytes)` --
plus `delete_all()`.  For source compatibility a backend may instead expose a mapping as `self.storage`, the
reference's convention (base.py:13-21); the primitives default to it.

Key grammar (must match the reference byte for byte, it is the on-disk format of an index):
    f"{key}:int"       -> decimal ASCII                      base.py:29-30, 48-52, 61-75
    f"{key}:string"    -> UTF-8 text                         base.py:32-33, 77-84
    f"{key}:bitarray"  -> row bytes, big-endian bit order    base.py:35-36, 85-109
"""
import gc

from ..bitrow import BitRow, row_bytes_of

_INT, _STR, _ROW = "int", "string", "bitarray"


def typed_key(key, kind):
    return ("%s:%s" % (key, kind)).encode("utf-8")


class BaseStorage(object):
    # ---- primitives -----------------------------------------------------------------------------
    def _get_raw(self, key):
        return self.storage[key]

    def _put_raw(self, key, value):
        self.storage[key] = value

    def _get_many_raw(self, keys):
        """Values in key order.  Backends with a native multi-get override this (cf. rocksdb.py:71-74)."""
        return [self._get_raw(k) for k in keys]

    def _put_many_raw(self, keys, values):
        for k, v in zip(keys, values):
            self._put_raw(k, v)

    def delete_all(self):
        raise NotImplementedError("Implemented in subclass")

    def sync(self):
        pass

    def close(self):
        self.storage = None
        gc.collect()

    # ---- untyped access (storage["k"] = b"..") ---------------------------------------------------
    @staticmethod
    def convert_key_to_bytes(key):
        return key if isinstance(key, bytes) else key.encode("utf-8")

    def __getitem__(self, key):
        return self._get_raw(self.convert_key_to_bytes(key))

    def __setitem__(self, key, value):
        self._put_raw(self.convert_key_to_bytes(key), value)

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def batch_get(self, keys):
        return self._get_many_raw([self.convert_key_to_bytes(k) for k in keys])

    def batch_set(self, keys, values):
        self._put_many_raw([self.convert_key_to_bytes(k) for k in keys], list(values))

    # names the reference's callers and tests use for the key grammar
    def convert_to_integer_key(self, key):
        return typed_key(key, _INT).decode("utf-8")

    def convert_to_string_key(self, key):
        return typed_key(key, _STR).decode("utf-8")

    def convert_to_bitarray_key(self, key):
        return typed_key(key, _ROW).decode("utf-8")

    # ---- integers --------------------------------------------------------------------------------
    def get_integer(self, key):
        return int(self._get_raw(typed_key(key, _INT)).decode("utf-8"))

    def set_integer(self, key, value):
        self._put_raw(typed_key(key, _INT), str(value).encode("utf-8"))

    def get_integers(self, keys):
        return [int(v.decode("utf-8")) for v in self._get_many_raw([typed_key(k, _INT) for k in keys])]

    def set_integers(self, keys, values):
        self._put_many_raw([typed_key(k, _INT) for k in keys], [str(v).encode("utf-8") for v in values])

    def incr(self, key):
        """Read-modify-write counter starting at 1 (base.py:135-144)."""
        try:
            nxt = self.get_integer(key) + 1
        except KeyError:
            nxt = 1
        self.set_integer(key, nxt)
        return nxt

    # ---- strings ---------------------------------------------------------------------------------
    def get_string(self, key):
        return self._get_raw(typed_key(key, _STR)).decode("utf-8")

    def set_string(self, key, value):
        assert isinstance(value, str)
        self._put_raw(typed_key(key, _STR), value.encode("utf-8"))

    # ---- bit rows --------------------------------------------------------------------------------
    @staticmethod
    def load_bitarray(raw):
        return BitRow.frombytes(raw)

    def get_bitarray(self, key):
        return BitRow.frombytes(self._get_raw(typed_key(key, _ROW)))

    def set_bitarray(self, key, value):
        self._put_raw(typed_key(key, _ROW), row_bytes_of(value)[0])

    def get_bitarrays(self, keys):
        return (BitRow.frombytes(raw) for raw in self._get_many_raw([typed_key(k, _ROW) for k in keys]))

    def set_bitarrays(self, keys, values):
        self._put_many_raw([typed_key(k, _ROW) for k in keys], [row_bytes_of(v)[0] for v in values])

    def get_bit(self, key, pos):
        return self.get_bitarray(key)[pos]

    def set_bit(self, key, pos, bit):
        """Row read-modify-write; a position one past the end appends (base.py:111-117)."""
        row = self.get_bitarray(key)
        if pos < len(row):
            row[pos] = bit
        else:
            row.append(bit)
        self.set_bitarray(key, row)

    def get_bits(self, keys, positions):
        return (self.get_bit(k, p) for k, p in zip(keys, positions))

    def set_bits(self, keys, positions, bits):
        for k, p, b in zip(keys, positions, bits):
            self.set_bit(k, p, b)
