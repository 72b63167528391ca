/*
 * epik_oracle_driver.c -- the CPU restatement run END TO END, the way the reference's driver runs
 * (epik/src/epik/main.cpp:322-381): FASTA read batch by batch (--batch-size, main.cpp:332-340) ->
 * placer::place on the batch (dedup + OpenMP loop, place.cpp:201-275; here orc_place_batched) -> the batch
 * appended to the jplace file (main.cpp:361, jplace.cpp:104-158) -> next batch; read and write on the main
 * thread, as the reference.  Prints "Placement time: <ms> ms" for exactly the span the reference times
 * (after the database is loaded, until the jplace file is closed).
 *
 * TEST INFRASTRUCTURE ONLY (see epik_oracle.h): bench.py's cpu_baseline leg times it beside the GPU driver
 * (timing (b)); tests compare its jplace with the GPU driver's.  It reads this build's own
 * EPIKAMD1 container (the reference's .ipk cannot be read here) in the dense form the oracle takes.
 *
 *   epik_oracle_driver <db.ekdb> <query.fasta> <out.jplace> <tree_with_edge_numbers.txt> <lengths.f64> <threads> [batch]
 * lengths.f64: num_branches x {distal, pendant} doubles (place.cpp:110-123, 435-437), from the caller.
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "epik_oracle.h"

static void die(const char *msg)
{
    fprintf(stderr, "epik_oracle_driver: %s\n", msg);
    exit(1);
}

static void *slurp(const char *path, size_t *size)
{
    FILE *f = fopen(path, "rb");
    if (!f) die("cannot open an input file");
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)n + 1);
    if (!buf || fread(buf, 1, (size_t)n, f) != (size_t)n) die("cannot read an input file");
    buf[n] = 0;
    fclose(f);
    *size = (size_t)n;
    return buf;
}

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec / 1e6;
}

/* the character classes of the k-mer encoder (epik_amd/alphabet.py; the oracle's tests use the same) */
static void char_classes(int amino, uint32_t *table)
{
    memset(table, 0, 256 * sizeof(uint32_t));
    const char *states = amino ? "RHKDESTNQCGPAILMFWYV" : "ACGT";
    for (int i = 0; states[i]; ++i) table[(unsigned char)states[i]] = 1u << i;
    if (!amino) {
        table['U'] = table['T'];
        const char *amb[][2] = {{"R", "AG"}, {"Y", "CT"}, {"S", "CG"}, {"W", "AT"}, {"K", "GT"}, {"M", "AC"},
                                {"B", "CGT"}, {"D", "AGT"}, {"H", "ACT"}, {"V", "ACG"}, {"N", "ACGT"}};
        for (size_t a = 0; a < sizeof amb / sizeof amb[0]; ++a)
            for (const char *m = amb[a][1]; *m; ++m) table[(unsigned char)amb[a][0][0]] |= table[(unsigned char)*m];
    } else {
        table['B'] = table['D'] | table['N'];
        table['Z'] = table['E'] | table['Q'];
        table['J'] = table['I'] | table['L'];
        table['X'] = (1u << 20) - 1u;
    }
    for (int c = 'A'; c <= 'Z'; ++c) table[c - 'A' + 'a'] = table[c];
}

int main(int argc, char **argv)
{
    if (argc < 7) die("usage: epik_oracle_driver db.ekdb query.fasta out.jplace tree.txt lengths.f64 threads [batch]");
    const int threads = atoi(argv[6]);
    const uint64_t batch_size = argc > 7 ? strtoull(argv[7], NULL, 10) : 2000; /* main.cpp:214 */

    /* ---- the database (outside the timed span, as the reference's) ------------------------------ */
    size_t db_size = 0;
    const unsigned char *file = (const unsigned char *)slurp(argv[1], &db_size);
    if (db_size < 48 || memcmp(file, "EPIKAMD1", 8) != 0) die("not an EPIKAMD1 database");
    uint32_t seq_type, k;
    float omega;
    uint64_t num_kmers, newick_len;
    memcpy(&seq_type, file + 12, 4);
    memcpy(&k, file + 16, 4);
    memcpy(&omega, file + 20, 4);
    memcpy(&num_kmers, file + 24, 8);
    memcpy(&newick_len, file + 40, 8);
    const uint32_t sigma = seq_type == 0 ? 4 : 20;
    uint64_t num_keys = 1;
    for (uint32_t i = 0; i < k; ++i) num_keys *= sigma;
    uint64_t *offsets = (uint64_t *)calloc(num_keys + 1, sizeof(uint64_t));
    if (!offsets) die("out of memory");
    const unsigned char *at = file + 48 + newick_len;
    uint64_t total = 0;
    for (uint64_t r = 0; r < num_kmers; ++r) { /* lengths first */
        uint32_t key, n;
        memcpy(&key, at, 4);
        memcpy(&n, at + 4, 4);
        offsets[key + 1] = n;
        total += n;
        at += 8 + (size_t)n * 8;
    }
    for (uint64_t i = 0; i < num_keys; ++i) offsets[i + 1] += offsets[i];
    orc_pkdb_value *values = (orc_pkdb_value *)malloc((total ? total : 1) * sizeof(orc_pkdb_value));
    if (!values) die("out of memory");
    at = file + 48 + newick_len;
    for (uint64_t r = 0; r < num_kmers; ++r) {
        uint32_t key, n;
        memcpy(&key, at, 4);
        memcpy(&n, at + 4, 4);
        memcpy(values + offsets[key], at + 8, (size_t)n * 8);
        at += 8 + (size_t)n * 8;
    }
    size_t tree_size = 0, lengths_size = 0;
    char *tree_text = (char *)slurp(argv[4], &tree_size);
    while (tree_size && (tree_text[tree_size - 1] == '\n' || tree_text[tree_size - 1] == '\r')) tree_text[--tree_size] = 0;
    const double *lengths = (const double *)slurp(argv[5], &lengths_size);
    const uint32_t num_branches = (uint32_t)(lengths_size / 16);
    uint32_t cls[256];
    char_classes(seq_type != 0, cls);
    orc_db db;
    memset(&db, 0, sizeof db);
    db.kmer_size = k;
    db.alphabet_size = sigma;
    db.num_branches = num_branches;
    db.keep_at_most = 7;  /* main.cpp:219 */
    db.keep_factor = 0.01; /* main.cpp:220 */
    db.threshold = (float)pow((double)(omega > 1.5f ? omega : 1.5f) / (double)sigma, (double)k); /* place.cpp:87 */
    db.log_threshold = log10f(db.threshold);
    db.num_keys = num_keys;
    db.offsets = offsets;
    db.values = values;
    db.char_class = cls;
    struct orc_hash *hash = orc_hash_create(&db); /* the reference's lookup structure: a hash map */
    if (!hash) die("out of memory");
    db.hash = hash;

    /* ---- the timed span: main.cpp:322-381 -------------------------------------------------------- */
    const double begin = now_ms();
    FILE *fasta = fopen(argv[2], "rb");
    FILE *out = fopen(argv[3], "wb");
    if (!fasta || !out) die("cannot open the query or the output");
    static char out_buffer[1 << 20];
    setvbuf(out, out_buffer, _IOFBF, sizeof out_buffer);
    fprintf(out, "{\n    \"metadata\": {\"invocation\": \"epik_oracle_driver\"},\n    \"tree\": \"%s\",\n    \"version\": 3,\n"
                 "    \"fields\": [\"edge_num\", \"likelihood\", \"like_weight_ratio\", \"distal_length\", \"pendant_length\"],\n"
                 "    \"placements\": [", tree_text);
    /* a batch: headers and sequences, each NUL-terminated in two growing arenas */
    size_t cap_h = 1 << 20, cap_s = 1 << 22;
    char *headers = (char *)malloc(cap_h), *seqs = (char *)malloc(cap_s);
    uint64_t *seq_off = (uint64_t *)malloc((batch_size + 1) * sizeof(uint64_t));
    uint64_t *head_off = (uint64_t *)malloc((batch_size + 1) * sizeof(uint64_t));
    orc_row *rows = (orc_row *)malloc(batch_size * db.keep_at_most * sizeof(orc_row));
    uint32_t *n_rows = (uint32_t *)malloc(batch_size * sizeof(uint32_t));
    uint32_t *counts = (uint32_t *)malloc(batch_size * db.keep_at_most * sizeof(uint32_t));
    uint64_t *first = (uint64_t *)malloc(batch_size * sizeof(uint64_t)); /* duplicate -> its first occurrence */
    uint64_t *next_same = (uint64_t *)malloc(batch_size * sizeof(uint64_t)); /* -> the next read of the same content */
    uint64_t *last_same = (uint64_t *)malloc(batch_size * sizeof(uint64_t));
    if (!headers || !seqs || !seq_off || !head_off || !rows || !n_rows || !counts || !first || !next_same || !last_same)
        die("out of memory");
    char *line = NULL;
    size_t line_cap = 0;
    ssize_t len = getline(&line, &line_cap, fasta);
    uint64_t placed = 0;
    int first_object = 1;
    while (len >= 0) {
        uint64_t m = 0;
        size_t used_h = 0, used_s = 0;
        seq_off[0] = 0;
        while (len >= 0 && m < batch_size) { /* i2l::io::batch_fasta: header line, sequence lines up to the next '>' */
            if (line[0] != '>') {
                len = getline(&line, &line_cap, fasta);
                continue;
            }
            while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
            if (used_h + (size_t)len + 1 > cap_h) headers = (char *)realloc(headers, cap_h = (cap_h + (size_t)len) * 2);
            if (!headers) die("out of memory");
            head_off[m] = used_h;
            memcpy(headers + used_h, line + 1, (size_t)len); /* with the NUL */
            used_h += (size_t)len;
            while ((len = getline(&line, &line_cap, fasta)) >= 0 && line[0] != '>') {
                while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
                if (used_s + (size_t)len + 1 > cap_s) seqs = (char *)realloc(seqs, cap_s = (cap_s + (size_t)len) * 2);
                if (!seqs) die("out of memory");
                memcpy(seqs + used_s, line, (size_t)len);
                used_s += (size_t)len;
            }
            seq_off[++m] = used_s;
        }
        if (m == 0) break;
        /* placer::place: dedup by content + the OpenMP loop (place.cpp:201-275) */
        if (orc_place_batched(&db, seqs, seq_off, m, m, threads, rows, n_rows, counts)) die("placement failed");
        /* jplace_writer << placed: one object per unique sequence, its names under "nm" (jplace.cpp:104-158).
         * The duplicates of a sequence carry the same rows; they are found again by content, first occurrence first. */
        for (uint64_t i = 0; i < m; ++i) first[i] = last_same[i] = i, next_same[i] = m;
        {
            /* hash by length + prefix to pair equal sequences cheaply */
            uint64_t cap = 16;
            while (cap < 2 * m) cap *= 2;
            uint64_t *table = (uint64_t *)calloc(cap, sizeof(uint64_t));
            if (!table) die("out of memory");
            for (uint64_t i = 0; i < m; ++i) {
                const char *s = seqs + seq_off[i];
                const size_t l = (size_t)(seq_off[i + 1] - seq_off[i]);
                uint64_t h = 1469598103934665603ull;
                for (size_t c = 0; c < l; ++c) h = (h ^ (unsigned char)s[c]) * 1099511628211ull;
                for (uint64_t slot = h & (cap - 1);; slot = (slot + 1) & (cap - 1)) {
                    if (!table[slot]) {
                        table[slot] = i + 1;
                        break;
                    }
                    const uint64_t j = table[slot] - 1;
                    if (seq_off[j + 1] - seq_off[j] == l && memcmp(seqs + seq_off[j], s, l) == 0) {
                        first[i] = j;
                        next_same[last_same[j]] = i; /* the names of a sequence in input order */
                        last_same[j] = i;
                        break;
                    }
                }
            }
            free(table);
        }
        for (uint64_t i = 0; i < m; ++i) {
            if (first[i] != i) continue;
            fputs(first_object ? "\n        {\n            \"p\": [" : ",\n        {\n            \"p\": [", out);
            first_object = 0;
            for (uint32_t r = 0; r < n_rows[i]; ++r) {
                const orc_row *row = rows + i * db.keep_at_most + r;
                const uint32_t c = counts[i * db.keep_at_most + r];
                const double distal = c ? lengths[2 * row->branch] : 0.0, pendant = c ? lengths[2 * row->branch + 1] : 0.0;
                fprintf(out, "%s\n                [%u, %.17g, %.17g, %.17g, %.17g]", r ? "," : "", row->branch, (double)row->score,
                        row->lwr, distal, pendant);
            }
            fputs(n_rows[i] ? "\n            ],\n            \"nm\": [" : "],\n            \"nm\": [", out);
            int first_name = 1;
            for (uint64_t j = i; j < m; j = next_same[j]) {
                fprintf(out, "%s\n                [\"", first_name ? "" : ",");
                for (const char *c = headers + head_off[j]; *c; ++c) {
                    if (*c == '"' || *c == '\\') fputc('\\', out);
                    fputc(*c, out);
                }
                fputs("\", 1]", out);
                first_name = 0;
            }
            fputs("\n            ]\n        }", out);
        }
        placed += m;
    }
    fputs(first_object ? "]\n}\n" : "\n    ]\n}\n", out);
    fclose(out);
    fclose(fasta);
    const double ms = now_ms() - begin;
    printf("Placed %llu sequences.\nPlacement time: %.0f ms\n", (unsigned long long)placed, ms);
    orc_hash_destroy(hash);
    return 0;
}
