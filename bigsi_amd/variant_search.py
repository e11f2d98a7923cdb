"""Variant genotyping on top of exact search (reference bigsi/cmds/variant_search.py:14-115).

A variant is looked up as two probe sets -- sequences carrying the reference allele and sequences carrying the alternate
allele, each with k-1 flanking bases -- and a sample's genotype follows from which set it matches exactly: both -> "0/1",
ref only -> "0/0", alt only -> "1/1".  The reference runs one `search(probe, 1)` per probe; here all probes of a variant
(or of many variants, `genotype_many`) go to the device as ONE batch.

Probe generation itself is the external `mykrobe variants make-probes` tool in the reference (variant_search.py:46-59,
83-99).  It is called the same way here when installed; `probes=` accepts its output (FASTA text, bytes or a path) so
the search side works on hosts without it.  Record names containing "ref" are reference alleles, all others alternates
(variant_search.py:28-32)."""
import io
import os
import subprocess

from .frontend import read_fasta


def split_probes(fasta):
    """(refs, alts) from make-probes output."""
    if isinstance(fasta, bytes):
        fasta = fasta.decode()
    if isinstance(fasta, str) and not fasta.lstrip().startswith(">") and os.path.exists(fasta):
        recs = read_fasta(fasta)
    else:
        recs = read_fasta(io.StringIO(fasta))
    refs = [s for name, s in recs if "ref" in name]
    alts = [s for name, s in recs if "ref" not in name]
    return refs, alts


def genotypes(ref_samples, alt_samples):
    """[{sample_name, genotype}] from the sample names matching the ref / alt probes (variant_search.py:61-75); samples in
    order of first appearance (the reference iterates a set, i.e. in no defined order)."""
    ref_set, alt_set = set(ref_samples), set(alt_samples)
    out = []
    for name in dict.fromkeys(list(ref_samples) + list(alt_samples)):
        if name in ref_set and name in alt_set:
            g = "0/1"
        elif name in ref_set:
            g = "0/0"
        else:
            g = "1/1"
        out.append({"sample_name": name, "genotype": g})
    return out


class BIGSIVariantSearch(object):
    def __init__(self, bigsi, reference=None):
        self.bigsi = bigsi
        self.reference = reference

    # -- probes
    def _make_probes_cmd(self, var_name):
        return ["mykrobe", "variants", "make-probes", "-k", str(self.bigsi.kmer_size), "-v", var_name, self.reference]

    def create_variant_probe_set(self, var_name):
        return subprocess.check_output(self._make_probes_cmd(var_name))

    # -- search
    def search_for_alleles(self, ref_seqs, alt_seqs):
        ref_seqs, alt_seqs = list(ref_seqs), list(alt_seqs)
        res = self.bigsi.search_batch(ref_seqs + alt_seqs, 1, score=False)
        names = [[r["sample_name"] for r in one] for one in res]
        return {"ref": [n for one in names[:len(ref_seqs)] for n in one],
                "alt": [n for one in names[len(ref_seqs):] for n in one]}

    def genotype_alleles(self, refs, alts):
        found = self.search_for_alleles(refs, alts)
        return genotypes(found["ref"], found["alt"])

    def genotype_many(self, probe_sets):
        """[(refs, alts), ...] -> one result list per variant, all probes in one device batch."""
        flat, spans = [], []
        for refs, alts in probe_sets:
            refs, alts = list(refs), list(alts)
            spans.append((len(flat), len(refs), len(alts)))
            flat += refs + alts
        res = self.bigsi.search_batch(flat, 1, score=False) if flat else []
        names = [[r["sample_name"] for r in one] for one in res]
        out = []
        for start, nr, na in spans:
            out.append(genotypes([n for one in names[start:start + nr] for n in one],
                                 [n for one in names[start + nr:start + nr + na] for n in one]))
        return out

    def _name(self, ref_base, pos, alt_base):
        return "".join([ref_base, str(pos), alt_base])

    def search(self, ref_base, pos, alt_base="X", probes=None):
        var_name = self._name(ref_base, pos, alt_base)
        refs, alts = split_probes(probes if probes is not None else self.create_variant_probe_set(var_name))
        return {"query": var_name, "results": self.genotype_alleles(refs, alts)}


class BIGSIAminoAcidMutationSearch(BIGSIVariantSearch):
    def __init__(self, bigsi, reference=None, genbank=None):
        super(BIGSIAminoAcidMutationSearch, self).__init__(bigsi, reference)
        self.genbank = genbank

    def _make_probes_cmd(self, var_name):
        return ["mykrobe", "variants", "make-probes", "-k", str(self.bigsi.kmer_size), "-v", var_name,
                "-g", self.genbank, self.reference]

    def search(self, gene, ref, pos, alt, probes=None):
        name = "_".join([gene, self._name(ref, pos, alt)])
        refs, alts = split_probes(probes if probes is not None else self.create_variant_probe_set(name))
        return {"query": name, "results": self.genotype_alleles(refs, alts)}
