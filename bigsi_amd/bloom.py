"""Bloom filter construction (bigsi/bloom/bloomfilter.py:5-39), hashed on the device.

`generate_hashes` and `BloomFilter` keep the reference's names and argument meaning; the MurmurHash3 / floor-mod /
bit-set work runs in libbigsi_hip.so (k_bloom), so a filter built here is bit-identical to what the query kernels
probe.  Unlike the reference's `bitarray(m)` (bloomfilter.py:20, uninitialised memory) filters start all-zero."""
import numpy as np

from . import _lib
from ._lib import check
from .bitrow import BitRow

DEFAULT_DEVICE = 0


def _device_bloom(elements, m, h, raw, device=None):
    """uint8[ceil(m/8)]: the m-bit filter of `elements` (strings of any one length per call group)."""
    from .utils import canonical
    out = np.zeros((int(m) + 7) // 8, dtype=np.uint8)
    dev = DEFAULT_DEVICE if device is None else device
    # The reference hashes the UTF-8 bytes of k CHARACTERS (bloomfilter.py:5-6).  ASCII elements go to the device as they are
    # (it canonicalises byte-wise, which is character-wise for ASCII); anything else is canonicalised here, character by
    # character as utils/fncts.py:38-54 does, and hashed raw on the device as its UTF-8 bytes.
    groups = {}
    for e in elements:
        if isinstance(e, str) and not e.isascii():
            data = (e if raw else canonical(e)).encode("utf-8")
            groups.setdefault((len(data), True), []).append(data)
        else:
            groups.setdefault((len(e), raw), []).append(e)
    for (k, as_is), group in groups.items():
        if k == 0:
            raise ValueError("cannot hash an empty element")
        blob, _ = _lib.pack_seqs(group) if not isinstance(group[0], bytes) else (b"".join(group), None)
        part = np.zeros_like(out)
        check(_lib.lib().bigsi_hip_bloom(dev, blob, len(group), k, int(m), int(h), _lib.BLOOM_RAW if as_is else 0, _lib.ptr(part)))
        out |= part
    return out


def generate_hashes(element, number_hash_functions, bloomfilter_size):
    """{mmh3.hash(element, seed) % m for seed in range(h)} (bloomfilter.py:9-13); the element is hashed as given."""
    filt = _device_bloom([element], bloomfilter_size, number_hash_functions, raw=True)
    out = set()
    for byte in np.flatnonzero(filt):            # at most h non-zero bytes, whatever m is
        for j in range(8):
            if filt[byte] & (0x80 >> j):
                out.add(int(byte) * 8 + j)
    return out


class BloomFilter(object):
    def __init__(self, m, h):
        self.m, self.h = m, h
        self.bitarray = BitRow(int(m))

    def add(self, e):
        return self.update([e])

    def update(self, elements):
        elements = list(elements)
        if elements:
            got = BitRow.frombytes(_device_bloom(elements, self.m, self.h, raw=True).tobytes(), int(self.m))
            self.bitarray._bits |= got._bits
        return self


def load_bitarray(f):
    with open(f, "rb") as inf:
        return BitRow.frombytes(inf.read())
