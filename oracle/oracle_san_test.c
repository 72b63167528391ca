/* oracle_san_test.c -- drives the oracle on a small random database and read set with four
 * threads, both lookup variants, batched and plain; built with -fsanitize=address,undefined
 * (oracle/Makefile: sanitize).  Test infrastructure. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "epik_oracle.h"

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd(void)
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

int main(void)
{
    enum { K = 5, SIGMA = 4, N = 97, KEYS = 1024, READS = 3000 };
    uint64_t *offsets = (uint64_t *)calloc(KEYS + 1, sizeof(uint64_t));
    for (int key = 0; key < KEYS; ++key) offsets[key + 1] = offsets[key] + (rnd() % 3 ? rnd() % 40 : 0);
    orc_pkdb_value *values = (orc_pkdb_value *)malloc((offsets[KEYS] + 1) * sizeof(*values));
    for (int key = 0; key < KEYS; ++key) {
        const uint64_t len = offsets[key + 1] - offsets[key];
        const uint32_t start = (uint32_t)(rnd() % (N - len + 1)); /* distinct branches: a run */
        for (uint64_t j = 0; j < len; ++j) {
            values[offsets[key] + j].branch = start + (uint32_t)j;
            values[offsets[key] + j].score = -(float)(rnd() % 4000) / 1000.0f - 0.001f;
        }
    }
    uint32_t char_class[256] = {0};
    char_class['A'] = 1, char_class['C'] = 2, char_class['G'] = 4, char_class['T'] = 8, char_class['N'] = 15, char_class['R'] = 5;
    orc_db db;
    memset(&db, 0, sizeof db);
    db.kmer_size = K, db.alphabet_size = SIGMA, db.num_branches = N, db.keep_at_most = 7, db.keep_factor = 0.01;
    db.threshold = 7.4157715e-3f, db.log_threshold = -2.1298437f; /* (1.5 / 4)^5 */
    db.num_keys = KEYS, db.offsets = offsets, db.values = values, db.char_class = char_class;

    const char letters[] = "ACGTACGTACGTACGTNR-";
    uint64_t *seq_offsets = (uint64_t *)calloc(READS + 1, sizeof(uint64_t));
    for (int i = 0; i < READS; ++i) seq_offsets[i + 1] = seq_offsets[i] + (i % 50 == 0 ? rnd() % 5 : 5 + rnd() % 200);
    char *seqs = (char *)malloc(seq_offsets[READS] + 1);
    for (uint64_t i = 0; i < seq_offsets[READS]; ++i) seqs[i] = letters[rnd() % (sizeof letters - 1)];
    for (int i = 10; i < 40; ++i) /* duplicates inside a batch */
        if (seq_offsets[i + 1] - seq_offsets[i] == seq_offsets[i + 101] - seq_offsets[i + 100])
            memcpy(seqs + seq_offsets[i + 100], seqs + seq_offsets[i], seq_offsets[i + 1] - seq_offsets[i]);

    orc_row *rows[4];
    uint32_t *n_rows[4], *counts[4];
    for (int v = 0; v < 4; ++v) {
        rows[v] = (orc_row *)calloc((size_t)READS * 7, sizeof(orc_row));
        n_rows[v] = (uint32_t *)calloc(READS, sizeof(uint32_t));
        counts[v] = (uint32_t *)calloc((size_t)READS * 7, sizeof(uint32_t));
    }
    struct orc_hash *hash = orc_hash_create(&db);
    int rc = orc_place_batch(&db, seqs, seq_offsets, READS, 4, rows[0], n_rows[0], counts[0]);
    rc |= orc_place_batched(&db, seqs, seq_offsets, READS, 500, 4, rows[1], n_rows[1], counts[1]);
    db.hash = hash;
    rc |= orc_place_batch(&db, seqs, seq_offsets, READS, 4, rows[2], n_rows[2], counts[2]);
    rc |= orc_place_batched(&db, seqs, seq_offsets, READS, 500, 1, rows[3], n_rows[3], counts[3]);
    db.hash = NULL;
    int bad = rc != 0;
    for (int v = 1; v < 4; ++v) {
        bad |= memcmp(n_rows[0], n_rows[v], READS * sizeof(uint32_t)) != 0;
        for (int i = 0; i < READS && !bad; ++i)
            for (uint32_t r = 0; r < n_rows[0][i]; ++r)
                bad |= rows[0][i * 7 + r].branch != rows[v][i * 7 + r].branch || rows[0][i * 7 + r].score != rows[v][i * 7 + r].score ||
                       rows[0][i * 7 + r].lwr != rows[v][i * 7 + r].lwr || counts[0][i * 7 + r] != counts[v][i * 7 + r];
    }
    orc_hash_destroy(hash);
    for (int v = 0; v < 4; ++v) free(rows[v]), free(n_rows[v]), free(counts[v]);
    free(seqs), free(seq_offsets), free(values), free(offsets);
    printf(bad ? "oracle sanitizer run: MISMATCH\n" : "oracle sanitizer run ok\n");
    return bad;
}
