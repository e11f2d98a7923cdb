/* `bigsi bulk_search` for a host that is not Python (C99, links libbigsi_hip.so only): FASTA file in, the reference's JSON / CSV text out
 * (bigsi/__main__.py:41-72, 261-314), in three calls of include/bigsi_hip.h -- bigsi_hip_fasta_pack, bigsi_hip_search_stream,
 * bigsi_hip_format_results.  The index is built from stdin as in search_host.c, with a sample name per line.
 * tests/test_frontend.py::test_c_host_bulk_search_prints_the_references_text compares its output with golden G9 (the reference's own
 * bulk_search output) and with frontend.bulk_search.
 *
 * usage: bulk_host <fasta> <threshold> <json|csv> <threshold_json> <citation_json>  < index description
 * stdin: m h k n_samples, then per sample a line with its name and a line "<n_seqs> seq seq ..."  (sample i = column i; name "D3L3T3D" = deleted) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bigsi_hip.h"
#include "bigsi_hip_text.h"

#define CHECK(call)                                                                             \
    do {                                                                                        \
        int rc_ = (call);                                                                       \
        if (rc_ != BIGSI_OK) {                                                                  \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, bigsi_hip_last_error());              \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)

enum { MAX_SEQ = 1 << 16 };

int main(int argc, char **argv)
{
    if (argc != 6) return 2;
    const double thr = atof(argv[2]);
    const int format = strcmp(argv[3], "csv") == 0 ? 1 : 0;
    unsigned long long m;
    unsigned h, k, n_samples;
    static char buf[MAX_SEQ], name[256];
    if (scanf("%llu %u %u %u", &m, &h, &k, &n_samples) != 4) return 2;
    bigsi_hip_index *ix = NULL;
    CHECK(bigsi_hip_open(m, 0, n_samples, h, 0, &ix));
    char *names = malloc(1);
    uint64_t *name_off = malloc((n_samples + 1) * sizeof *name_off), names_len = 0;
    uint8_t *deleted = calloc(n_samples ? n_samples : 1, 1);
    name_off[0] = 0;
    for (unsigned s = 0; s < n_samples; s++) {
        unsigned ns;
        if (scanf(" %255[^\n]", name) != 1 || scanf("%u", &ns) != 1) return 2;      /* the name: a line of its own (it may hold spaces, quotes, commas) */
        const size_t nl = strlen(name);
        names = realloc(names, names_len + nl + 1);
        memcpy(names + names_len, name, nl);
        names_len += nl;
        name_off[s + 1] = names_len;
        deleted[s] = strcmp(name, "D3L3T3D") == 0;
        char *blob = NULL;
        uint64_t *off = malloc((ns + 1) * sizeof *off), len = 0;
        off[0] = 0;
        for (unsigned i = 0; i < ns; i++) {
            if (scanf("%65535s", buf) != 1) return 2;
            const size_t l = strlen(buf);
            blob = realloc(blob, len + l + 1);
            memcpy(blob + len, buf, l);
            len += l;
            off[i + 1] = len;
        }
        CHECK(bigsi_hip_set_num_cols(ix, s + 1));
        CHECK(bigsi_hip_insert_kmers(ix, s, blob ? blob : "", off, ns, k));
        free(blob); free(off);
    }
    /* 1: the file's sequences, packed */
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    const long fsize = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *text = malloc(fsize > 0 ? (size_t)fsize : 1);
    if (fread(text, 1, (size_t)fsize, f) != (size_t)fsize) return 2;
    fclose(f);
    uint64_t n = 0;
    CHECK(bigsi_hip_fasta_pack(text, (uint64_t)fsize, NULL, NULL, 0, &n));              /* how many records */
    char *seqs = malloc(fsize > 0 ? (size_t)fsize : 1);
    uint64_t *off = malloc((n + 1) * sizeof *off);
    CHECK(bigsi_hip_fasta_pack(text, (uint64_t)fsize, seqs, off, n, &n));
    /* 2: the search (hit buffers grown on BIGSI_ERR_CAPACITY: the offsets are complete even then) */
    uint32_t *nk = malloc((n ? n : 1) * sizeof *nk), *nu = malloc((n ? n : 1) * sizeof *nu);
    uint64_t *ho = calloc(n + 1, sizeof *ho), cap = 4;
    uint32_t *col = malloc(cap * sizeof *col), *cnt = malloc(cap * sizeof *cnt);
    if (n) {
        int rc;
        for (;;) {
            rc = bigsi_hip_search_stream(ix, seqs, off, n, k, thr, 0u, nk, nu, NULL, ho, col, cnt, cap);
            if (rc != BIGSI_ERR_CAPACITY) break;
            cap = ho[n];
            col = realloc(col, cap * sizeof *col);
            cnt = realloc(cnt, cap * sizeof *cnt);
        }
        if (rc != BIGSI_OK) { fprintf(stderr, "bigsi_hip_search_stream -> %d: %s\n", rc, bigsi_hip_last_error()); return 1; }
    }
    /* 3: the text */
    char *out = NULL;
    uint64_t out_bytes = 0;
    CHECK(bigsi_hip_format_results(format, seqs, off, n, argv[4], argv[5], thr == 1.0, nu, ho, col, cnt, names, name_off, deleted, n_samples, 0u,
                                   &out, &out_bytes));
    fwrite(out, 1, (size_t)out_bytes, stdout);
    bigsi_hip_free_text(out);
    free(text); free(seqs); free(off); free(nk); free(nu); free(ho); free(col); free(cnt); free(names); free(name_off); free(deleted);
    CHECK(bigsi_hip_close(ix));
    return 0;
}
