"""BitRow: the small subset of `bitarray` the reference's API surface exposes.

The reference hands rows around as `bitarray` objects (bigsi/storage/base.py:96-99: big-endian bit order,
`tobytes()` zero-pads the last byte).  `bitarray` is not installable here, and the hot path never touches
host-side rows anyway, so this is a minimal stand-in for API parity of `lookup()`, `BIGSI.bloom()`,
`insert()` and the storage contract: construction from a '01' string / iterable / bytes, `tobytes`, `to01`,
`&`, `==`, `len`, indexing, slicing, `count`, `tolist`, `append`, `extend`, `setall`.
A real `bitarray` is accepted anywhere a BitRow is (duck-typed through `tobytes()` + `len()`).
"""
import numpy as np


class BitRow:
    __slots__ = ("_bits",)

    def __init__(self, init=0):
        if isinstance(init, BitRow):
            self._bits = init._bits.copy()
        elif isinstance(init, str):
            if init.strip("01"):
                raise ValueError("BitRow string may only contain '0' and '1'")
            self._bits = (np.frombuffer(init.encode("ascii"), dtype=np.uint8) - ord("0")).astype(np.uint8)
        elif isinstance(init, (int, np.integer)):
            self._bits = np.zeros(int(init), dtype=np.uint8)      # zero-initialised (unlike bitarray(n))
        elif hasattr(init, "tobytes") and hasattr(init, "__len__") and not isinstance(init, np.ndarray):
            self._bits = np.unpackbits(np.frombuffer(init.tobytes(), dtype=np.uint8))[: len(init)].copy()
        else:
            self._bits = np.asarray(list(init) if not isinstance(init, np.ndarray) else init).astype(bool).astype(np.uint8)

    # -- bytes <-> bits -------------------------------------------------------------------------
    @classmethod
    def frombytes(cls, data, nbits=None):
        r = cls.__new__(cls)
        bits = np.unpackbits(np.frombuffer(bytes(data), dtype=np.uint8))
        r._bits = (bits if nbits is None else bits[:nbits]).copy()
        return r

    def tobytes(self):
        return np.packbits(self._bits).tobytes()

    def to01(self):
        return (self._bits + ord("0")).astype(np.uint8).tobytes().decode("ascii")

    def tolist(self):
        return [bool(b) for b in self._bits]

    def count(self, value=1):
        ones = int(self._bits.sum())
        return ones if value else len(self._bits) - ones

    def length(self):
        return len(self._bits)

    def setall(self, value):
        self._bits[:] = 1 if value else 0

    def append(self, value):
        self._bits = np.append(self._bits, np.uint8(1 if value else 0))

    def extend(self, other):
        self._bits = np.concatenate([self._bits, BitRow(other)._bits])

    def copy(self):
        return BitRow(self)

    # -- operators ------------------------------------------------------------------------------
    def __len__(self):
        return len(self._bits)

    def __getitem__(self, i):
        if isinstance(i, slice):
            r = BitRow.__new__(BitRow)
            r._bits = self._bits[i].copy()
            return r
        return bool(self._bits[i])

    def __setitem__(self, i, value):
        self._bits[i] = 1 if value else 0      # IndexError past the end, like bitarray (base.py:113-116 relies on it)

    def __iter__(self):
        return (bool(b) for b in self._bits)

    def __and__(self, other):
        o = BitRow(other) if not isinstance(other, BitRow) else other
        if len(o) != len(self):
            raise ValueError("bitarrays of equal length expected for bitwise operation")
        r = BitRow.__new__(BitRow)
        r._bits = self._bits & o._bits
        return r

    def __eq__(self, other):
        if isinstance(other, str):
            return False
        try:
            o = other if isinstance(other, BitRow) else BitRow(other)
        except Exception:  # noqa: BLE001
            return NotImplemented
        return len(o) == len(self) and bool(np.array_equal(self._bits, o._bits))

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    __hash__ = None

    def __repr__(self):
        return "BitRow('%s')" % self.to01()


def row_bytes_of(obj):
    """(bytes, nbits) of a BitRow / bitarray / '01' string / bytes-like, in the reference's storage format."""
    if isinstance(obj, (bytes, bytearray, memoryview)):
        b = bytes(obj)
        return b, 8 * len(b)
    if isinstance(obj, str):
        obj = BitRow(obj)
    if hasattr(obj, "tobytes") and hasattr(obj, "__len__"):
        return obj.tobytes(), len(obj)
    raise TypeError("expected a BitRow/bitarray/bytes, got %r" % type(obj))
