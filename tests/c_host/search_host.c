/* A host of the C ABI that is not Python: plain C99, no torch, no HIP headers -- what a cgo / JNI / N-API binder of
 * include/bigsi_hip.h would compile to.  It builds a small index from the FASTA-like input it is given (sample sequences
 * -> bigsi_hip_insert_kmers, i.e. BIGSI.bloom + build, bigsi/graph/bigsi.py:150-155,92-112), runs every query through the
 * ONE-CALL entry point bigsi_hip_search_batch (BIGSI.search, bigsi/graph/bigsi.py:174-242) exactly and at a threshold, and
 * prints the hit lists as text.  tests/test_gpu_parity.py::test_c_host_of_the_abi runs it and compares the text with the
 * oracle; the CPU suite checks that it compiles as C against the header and links against the library.
 *
 * input (stdin):  m h k n_samples n_queries threshold
 *                 then n_samples lines  "<n_seqs> seq seq ..."   (sample i = column i)
 *                 then n_queries lines  "seq"
 * output:         per pass ("exact" / "threshold"), per query:  q <i> kmers <n> unique <u> min <mk> hits <c>:<count> ...
 *                 then "stream <pass> identical": bigsi_hip_search_stream over the same queries gave the same arrays
 *                 then "scored <pass> identical" and one "s ..." line per hit: bigsi_hip_search_stream_scored's records and bits
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bigsi_hip.h"

#define CHECK(call)                                                                             \
    do {                                                                                        \
        int rc_ = (call);                                                                       \
        if (rc_ != BIGSI_OK) {                                                                  \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, bigsi_hip_last_error());              \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)

enum { MAX_SEQ = 1 << 16 };

static int run_pass(bigsi_hip_index *ix, const char *name, const char *blob, const uint64_t *off, uint32_t nq, uint32_t k, double thr)
{
    uint32_t *nk = malloc(nq * sizeof *nk), *nu = malloc(nq * sizeof *nu), *mk = malloc(nq * sizeof *mk);
    uint64_t *ho = malloc((nq + 1) * sizeof *ho);
    uint64_t cap = 4;                       /* deliberately small: the CAPACITY protocol is part of the boundary */
    uint32_t *col = malloc(cap * sizeof *col), *cnt = malloc(cap * sizeof *cnt);
    int rc;
    if (!nk || !nu || !mk || !ho || !col || !cnt) return 1;
    for (;;) {
        rc = bigsi_hip_search_batch(ix, blob, off, nq, k, thr, 0u, nk, nu, mk, ho, col, cnt, cap);
        if (rc != BIGSI_ERR_CAPACITY) break;
        cap = ho[nq];                       /* hit_offsets are filled even then: the total says how much to bring */
        col = realloc(col, cap * sizeof *col);
        cnt = realloc(cnt, cap * sizeof *cnt);
        if (!col || !cnt) return 1;
    }
    if (rc != BIGSI_OK) {
        fprintf(stderr, "bigsi_hip_search_batch -> %d: %s\n", rc, bigsi_hip_last_error());
        return 1;
    }
    printf("pass %s\n", name);
    for (uint32_t q = 0; q < nq; q++) {
        printf("q %u kmers %u unique %u min %u hits", q, nk[q], nu[q], mk[q]);
        for (uint64_t t = ho[q]; t < ho[q + 1]; t++) printf(" %u:%u", col[t], cnt[t]);
        printf("\n");
    }
    /* the streaming entry point (any number of sequences in one call: bulk_search, bigsi/__main__.py:261-314) must give the same */
    {
        uint32_t *nk2 = malloc(nq * sizeof *nk2), *nu2 = malloc(nq * sizeof *nu2), *mk2 = malloc(nq * sizeof *mk2);
        uint64_t *ho2 = malloc((nq + 1) * sizeof *ho2);
        uint32_t *col2 = malloc((cap ? cap : 1) * sizeof *col2), *cnt2 = malloc((cap ? cap : 1) * sizeof *cnt2);
        if (!nk2 || !nu2 || !mk2 || !ho2 || !col2 || !cnt2) return 1;
        rc = bigsi_hip_search_stream(ix, blob, off, nq, k, thr, 0u, nk2, nu2, mk2, ho2, col2, cnt2, cap);
        if (rc != BIGSI_OK) {
            fprintf(stderr, "bigsi_hip_search_stream -> %d: %s\n", rc, bigsi_hip_last_error());
            return 1;
        }
        int same = memcmp(nk, nk2, nq * sizeof *nk) == 0 && memcmp(nu, nu2, nq * sizeof *nu) == 0 && memcmp(mk, mk2, nq * sizeof *mk) == 0 &&
                   memcmp(ho, ho2, (nq + 1) * sizeof *ho) == 0 && memcmp(col, col2, ho[nq] * sizeof *col) == 0 &&
                   memcmp(cnt, cnt2, ho[nq] * sizeof *cnt) == 0;
        printf("stream %s %s\n", name, same ? "identical" : "DIFFERENT");
        free(nk2); free(nu2); free(mk2); free(ho2); free(col2); free(cnt2);
    }
    /* score=True in one call (bigsi/graph/bigsi.py:80-100 + bigsi/scoring/score.py:96-121): a sizing call, then the real one */
    {
        uint64_t need = 0, one = 0;
        uint64_t *ho3 = malloc((nq + 1) * sizeof *ho3);
        if (!ho3) return 1;
        rc = bigsi_hip_search_stream_scored(ix, blob, off, nq, k, thr, 0u, NULL, NULL, NULL, ho3, NULL, NULL, 0, NULL, 0, &one, NULL, &need);
        if (rc != BIGSI_OK && rc != BIGSI_ERR_CAPACITY) {
            fprintf(stderr, "bigsi_hip_search_stream_scored (sizing) -> %d: %s\n", rc, bigsi_hip_last_error());
            return 1;
        }
        uint64_t n_hits = ho3[nq];
        uint32_t *col3 = malloc((n_hits + 1) * sizeof *col3), *cnt3 = malloc((n_hits + 1) * sizeof *cnt3);
        uint64_t *bo = malloc((n_hits + 1) * sizeof *bo);
        uint8_t *bits = malloc(need + 8);
        bigsi_hip_hit_score *rec = malloc((n_hits + 1) * sizeof *rec);
        if (!col3 || !cnt3 || !bo || !bits || !rec) return 1;
        rc = bigsi_hip_search_stream_scored(ix, blob, off, nq, k, thr, 0u, NULL, NULL, NULL, ho3, col3, cnt3, n_hits, bits, need, bo, rec, &need);
        if (rc != BIGSI_OK) {
            fprintf(stderr, "bigsi_hip_search_stream_scored -> %d: %s\n", rc, bigsi_hip_last_error());
            return 1;
        }
        int same = memcmp(ho, ho3, (nq + 1) * sizeof *ho) == 0 && memcmp(col, col3, n_hits * sizeof *col) == 0 &&
                   memcmp(cnt, cnt3, n_hits * sizeof *cnt) == 0;
        printf("scored %s %s\n", name, same ? "identical" : "DIFFERENT");
        for (uint32_t q = 0; q < nq; q++)
            for (uint64_t t = ho3[q]; t < ho3[q + 1]; t++) {
                /* s <query> <colour> <score> <min_score> <max_score> <mismatches> <min_mismatches> <max_mismatches> <percent> <presence> */
                printf("s %u %u %.17g %.17g %.17g %lld %lld %lld %.17g ", q, col3[t], rec[t].score, rec[t].min_score, rec[t].max_score,
                       (long long)rec[t].mismatches, (long long)rec[t].min_mismatches, (long long)rec[t].max_mismatches, rec[t].percent_kmers_found);
                for (uint32_t i = 0; i < rec[t].num_kmers; i++) putchar((bits[bo[t] + (i >> 3)] >> (7 - (i & 7))) & 1 ? '1' : '0');
                putchar('\n');
            }
        free(ho3); free(col3); free(cnt3); free(bo); free(bits); free(rec);
    }
    free(nk); free(nu); free(mk); free(ho); free(col); free(cnt);
    return 0;
}

int main(void)
{
    unsigned long long m;
    unsigned h, k, n_samples, nq;
    double thr;
    static char buf[MAX_SEQ];
    if (scanf("%llu %u %u %u %u %lf", &m, &h, &k, &n_samples, &nq, &thr) != 6) return 2;
    int n_dev = 0;
    CHECK(bigsi_hip_device_count(&n_dev));
    if (n_dev < 1) { fprintf(stderr, "no device\n"); return 3; }
    bigsi_hip_index *ix = NULL;
    CHECK(bigsi_hip_open(m, 0, n_samples, h, 0, &ix));
    for (unsigned s = 0; s < n_samples; s++) {
        unsigned ns;
        if (scanf("%u", &ns) != 1) return 2;
        char *blob = NULL;
        uint64_t *off = malloc((ns + 1) * sizeof *off), len = 0;
        off[0] = 0;
        for (unsigned i = 0; i < ns; i++) {
            if (scanf("%65535s", buf) != 1) return 2;
            size_t l = strlen(buf);
            blob = realloc(blob, len + l + 1);
            memcpy(blob + len, buf, l);
            len += l;
            off[i + 1] = len;
        }
        CHECK(bigsi_hip_set_num_cols(ix, s + 1));            /* BitMatrix.insert_column appends: bitmatrix.py:67-75 */
        CHECK(bigsi_hip_insert_kmers(ix, s, blob ? blob : "", off, ns, k));
        free(blob); free(off);
    }
    char *qblob = NULL;
    uint64_t *qoff = malloc((nq + 1) * sizeof *qoff), qlen = 0;
    qoff[0] = 0;
    for (unsigned i = 0; i < nq; i++) {
        if (scanf("%65535s", buf) != 1) return 2;
        size_t l = strlen(buf);
        qblob = realloc(qblob, qlen + l + 1);
        memcpy(qblob + qlen, buf, l);
        qlen += l;
        qoff[i + 1] = qlen;
    }
    bigsi_hip_info info;
    CHECK(bigsi_hip_get_info(ix, &info));
    printf("index rows %llu cols %llu hashes %u row_bytes %llu\n", (unsigned long long)info.num_rows, (unsigned long long)info.num_cols,
           info.num_hashes, (unsigned long long)info.row_bytes);
    if (run_pass(ix, "exact", qblob, qoff, nq, k, 1.0)) return 1;
    if (run_pass(ix, "threshold", qblob, qoff, nq, k, thr)) return 1;
    /* error behaviour reaches a C host as a code + message, never as an abort */
    if (bigsi_hip_search_batch(ix, qblob, qoff, 0, k, 1.0, 0u, NULL, NULL, NULL, NULL, NULL, NULL, 0) == BIGSI_OK) return 4;
    printf("error %s\n", bigsi_hip_last_error()[0] ? "reported" : "silent");
    free(qblob); free(qoff);
    CHECK(bigsi_hip_close(ix));
    return 0;
}
