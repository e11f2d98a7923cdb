"""Host-side string helpers with the reference's names (bigsi/utils/fncts.py).

These exist for API parity (`from bigsi import seq_to_kmers` style imports, building k-mer lists for
`BIGSI.bloom`).  The query path does NOT use them: k-merising, canonicalisation and hashing of queries run on
the device (bigsi_amd/csrc/bigsi_kernels.hpp: k_kmerize).
"""
import math

_COMPLEMENT = str.maketrans("ACGT", "TGCA")      # fncts.py:12 -- other characters map to themselves


def seq_to_kmers(seq, kmer_size):                # fncts.py:63-65
    for i in range(len(seq) - kmer_size + 1):
        yield seq[i:i + kmer_size]


def reverse_comp(s):                             # fncts.py:38-39
    return s[::-1].translate(_COMPLEMENT)


def canonical(k):                                # fncts.py:51-54
    rc = reverse_comp(k)
    return k if k <= rc else rc


convert_query_kmer = canonical                   # fncts.py:47-48


def convert_query_kmers(kmers):                  # fncts.py:42-44
    for k in kmers:
        yield canonical(k)


def chunks(l, n):                                # fncts.py:32-35
    for i in range(0, len(l), n):
        yield l[i:i + n]


def min_kmers_for(num_unique, threshold):        # graph/bigsi.py:179
    return math.ceil(num_unique * threshold)
