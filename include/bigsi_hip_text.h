/*
 * bigsi_hip_text.h -- the host-only text helpers of libbigsi_hip.so's batch front-end: no device code, no
 * index handle.  A binder that only searches needs include/bigsi_hip.h alone.
 */
#ifndef BIGSI_HIP_TEXT_H
#define BIGSI_HIP_TEXT_H

#include "bigsi_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ================================================================== FRONT-END TEXT (host only)
 * What `bigsi bulk_search` reads and returns (bigsi/__main__.py:41-72, 261-314: pyfasta records in, json.dumps(records, indent=4)
 * or csv.writer rows out), over the arrays of bigsi_hip_search_stream instead of a Python object per record and per result.
 *
 * bigsi_hip_fasta_pack: the sequences of a FASTA text (header lines start with '>' after stripping; the stripped lines of a record
 * concatenated; '\r', '\n' and "\r\n" all end a line) packed as the search entry points take them: record i =
 * out_seqs[out_offsets[i] .. out_offsets[i+1]).  out_seqs needs n_bytes bytes at most, out_offsets n_records + 1 entries: call
 * once with out_seqs = out_offsets = NULL for the count.  BIGSI_ERR_INVALID for a byte >= 0x80 (the caller's own text route
 * then decides what the file's encoding means).
 *
 * bigsi_hip_format_results: the text of an unscored bulk search.  format 0 = the JSON list (threshold_text / citation_text: the
 * JSON of those two values, written into every record), 1 = the CSV rows of every record joined by '\n' (no header; each
 * record's last '\n' dropped, as the reference does).  exact != 0: threshold == 1.0 (every hit reports all num_unique k-mers,
 * ascending colours); else hits below n_names in a stable sort by count, descending.  names / name_offsets: sample name of
 * colour c = names[name_offsets[c] .. name_offsets[c+1]) (only colours that occur need a non-empty name); name_deleted[c] != 0
 * drops the sample (graph/bigsi.py:186-190).  percent_kmers_found is repr(round(100 * float(found) / num_kmers, 2)).
 * BIGSI_ERR_STATE when the reference would raise instead of answering (a record without k-mers; exact hit on a colour without
 * a name): the caller's per-record route raises its exception in record order.  *out_text == NULL on entry: the text is malloc'ed,
 * NUL-terminated, *out_bytes long: release it with bigsi_hip_free_text.  *out_text != NULL: the caller's own buffer of *out_bytes
 * bytes (the body of a string object of the host language, say: no copy afterwards); BIGSI_ERR_CAPACITY with the size needed in
 * *out_bytes if it is too small -- a call with a zero-byte buffer is the sizing call.  threads = 0: up to 16 host threads. */
int bigsi_hip_fasta_pack(const char *text, uint64_t n_bytes, char *out_seqs, uint64_t *out_offsets, uint64_t max_records, uint64_t *n_records);
int bigsi_hip_format_results(int format, const char *seqs, const uint64_t *offsets, uint64_t n_seqs, const char *threshold_text,
                             const char *citation_text, int exact, const uint32_t *num_unique, const uint64_t *hit_offsets,
                             const uint32_t *colours, const uint32_t *counts, const char *names, const uint64_t *name_offsets,
                             const uint8_t *name_deleted, uint64_t n_names, uint32_t threads, char **out_text, uint64_t *out_bytes);
/* The same for score=True (bulk_search --score): every result carries the 17 fields of Scorer.score and "kmer-presence"
 * (bigsi/scoring/score.py:96-121, graph/bigsi.py:232-239) after the four above, in the reference's key order (CSV: sorted keys).
 * `scored` (per hit, in the order of colours / counts): K6's records and presence bits as bigsi_hip_batch_score_hits /
 * bigsi_hip_search_stream_scored return them, and the four closed-form columns the caller's own math library computes (score.py:
 * 125-151: no two libm agree bit for bit on exp / log10); nident / pident / length are derived here.  Floats are written as
 * Python's repr() writes them (shortest round-trip decimal).  BIGSI_ERR_STATE also for a scored hit of a one-k-mer query (IndexError
 * in the reference).  scored == NULL: bigsi_hip_format_results. */
typedef struct {
    const bigsi_hip_hit_score *scores;
    const uint8_t *bits;
    const uint64_t *bit_offsets;
    const double *evalue, *pvalue, *log_evalue, *log_pvalue;
    uint32_t k; /* length = num_kmers + k - 1 */
    uint32_t reserved;
} bigsi_hip_scored_text;
int bigsi_hip_format_results_scored(int format, const char *seqs, const uint64_t *offsets, uint64_t n_seqs, const char *threshold_text,
                                    const char *citation_text, int exact, const uint32_t *num_unique, const uint64_t *hit_offsets,
                                    const uint32_t *colours, const uint32_t *counts, const char *names, const uint64_t *name_offsets,
                                    const uint8_t *name_deleted, uint64_t n_names, const bigsi_hip_scored_text *scored, uint32_t threads,
                                    char **out_text, uint64_t *out_bytes);
void bigsi_hip_free_text(char *text);

#ifdef __cplusplus
}
#endif
#endif /* BIGSI_HIP_TEXT_H */
