"""k-mers of a Cortex graph file (`.ctx`, format version 6) -- the input of the reference's `bloom` command
(bigsi/__main__.py:120-131 -> bigsi/utils/cortex.py:23-27).

Only what Bloom-filter construction needs: the header is walked to find the record area, every record's k-mer word is
decoded (2 bits per base, A=0 C=1 G=2 T=3, last base in the lowest bits; k <= 31 as in the reference), replaced by the
lexicographically smaller of itself and its reverse complement, and cut into windows of the index's k.  Coverages, edges
and link files are not read.  Pinned by tests/golden/g10_cortex.json (the reference's reader on its own three files).
"""
import struct

import numpy as np

_MAGIC = b"CORTEX"


class CortexFormatError(ValueError):
    pass


def read_header(f):
    """(kmer_size, words_per_kmer, num_colours, payload_offset) of an open binary file positioned at 0."""
    def u32():
        b = f.read(4)
        if len(b) != 4:
            raise CortexFormatError("truncated header")
        return struct.unpack("<I", b)[0]

    if f.read(len(_MAGIC)) != _MAGIC:
        raise CortexFormatError("not a Cortex graph file")
    version = u32()
    if version != 6:
        raise CortexFormatError("Cortex format version %d is not supported (only 6)" % version)
    kmer_size, words, colours = u32(), u32(), u32()
    f.seek(12 * colours, 1)                     # per colour: mean read length (u32) + total sequence (u64)
    for _ in range(colours):
        f.seek(u32(), 1)                        # sample name
    f.seek(16 * colours, 1)                     # per colour: sequencing error rate (long double)
    for _ in range(colours):
        f.seek(12, 1)                           # cleaning flags / thresholds
        f.seek(u32(), 1)                        # name of the graph the colour was cleaned against
    if f.read(len(_MAGIC)) != _MAGIC:
        raise CortexFormatError("header does not end with the magic word")
    return kmer_size, words, colours, f.tell()


def read_kmers(path):
    """(kmer_size, list of canonical k-mer strings) of every record."""
    with open(path, "rb") as f:
        ksz, words, colours, start = read_header(f)
        if ksz > 31 or words != 1:
            raise CortexFormatError("k-mers longer than 31 bases are not supported")
        payload = f.read()
    rec = 8 * words + 5 * colours
    n = len(payload) // rec
    raw = np.frombuffer(payload[: n * rec], dtype=np.uint8).reshape(n, rec)[:, :8]
    vals = raw.copy().view("<u8").reshape(n)
    shifts = (2 * np.arange(ksz - 1, -1, -1)).astype(np.uint64)               # first base = highest used bits
    codes = ((vals[:, None] >> shifts[None, :]) & np.uint64(3)).astype(np.uint8)       # n x ksz, A0 C1 G2 T3
    fwd = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
    rev = np.frombuffer(b"ACGT", dtype=np.uint8)[(3 - codes)[:, ::-1]]
    out = []
    for a, b in zip(fwd, rev):
        a, b = a.tobytes(), b.tobytes()
        out.append((b if b < a else a).decode("ascii"))
    return ksz, out


def extract_kmers_from_ctx(ctx, k):
    """Same name and meaning as the reference's generator: windows of size k over each record's canonical k-mer."""
    _, kmers = read_kmers(ctx)
    for km in kmers:
        for i in range(len(km) - k + 1):
            yield km[i:i + k]
