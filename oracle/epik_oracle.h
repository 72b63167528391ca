/*
 * epik_oracle.h -- CPU restatement of EPIK's `epik::placer` hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and there only as the checker / the timed CPU baseline.
 * The product path (epik_amd/) never links, imports or calls it.
 *
 * PARITY UNPINNED.  The reference ships no tests, golden vectors or fixtures
 * for this path and cannot be built in this image
 * (i2l, Boost, RapidJSON, cxxopts absent; building against stand-in headers is
 * not allowed).  This restatement follows the reference source line by line
 * (file:line cited at every function) and is cross-checked against a second,
 * independently written numpy restatement (oracle/epik_oracle_np.py) and
 * hand-derived micro cases (tests/golden/), but not against reference output.
 *
 * What is restated (all from /root/reference/epik/src/epik/place.cpp):
 *   place_seq              :320-440   (exact + ambiguous accumulate, correction)
 *   query_kmers            :278-316
 *   sum_scores             :164-184
 *   select_best_placements :134-159
 *   LWR loop of place()    :237-264
 *   filter_by_ratio        :188-199
 * The i2l side (absent submodule `phylo42/i2l`, pinned commit unknown) is
 * restated from its call sites only; every assumption is one function here:
 *   orc_kmer_window()  <-> i2l::to_kmers<one_ambiguity_policy>  (place.cpp:294)
 *   CSR lookup         <-> i2l::phylo_kmer_db::search           (place.cpp:300,311)
 */
#ifndef EPIK_ORACLE_H
#define EPIK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* i2l::pkdb_value {branch, score}: 8 bytes (main.cpp:257, place.cpp:358). */
typedef struct {
    uint32_t branch; /* post-order node id (place.cpp:101-103) */
    float score;     /* log10 probability (place.cpp:181,391) */
} orc_pkdb_value;

/* One output placement row: the fields of epik::impl::placement that the hot
 * path computes (place.h:45-56); distal/pendant lengths are joined by the host. */
typedef struct {
    uint32_t branch;
    float score;
    double lwr;
} orc_row;

typedef struct {
    uint32_t kmer_size;      /* db.kmer_size() */
    uint32_t alphabet_size;  /* sigma: 4 (nucl) or 20 (amino) */
    uint32_t num_branches;   /* tree.get_node_count() (place.cpp:92,166) */
    uint32_t keep_at_most;   /* main.cpp:219, default 7 */
    double keep_factor;      /* main.cpp:220, default 0.01 */
    float threshold;         /* i2l::score_threshold(omega, k)  (place.cpp:87) */
    float log_threshold;     /* std::log10(threshold)           (place.cpp:88) */
    uint64_t num_keys;       /* sigma^k: dense direct-index key space */
    const uint64_t *offsets; /* [num_keys + 1] CSR offsets into values[] */
    const orc_pkdb_value *values;
    /* char_class[c]: bit i set <=> character c may be state i.
     * popcount 1 = unambiguous, >1 = ambiguous, 0 = invalid (gap/unknown). */
    const uint32_t *char_class; /* [256] */
    /* NULL: phylo_kmer_db::search is a direct index into offsets[].  Non-NULL (orc_hash_create): a
     * node-chained hash map key -> separately allocated vector of postings, the data structure shape
     * of the reference's i2l::phylo_kmer_db -- same results, the reference's memory
     * access pattern; used by the timed CPU baseline. */
    const struct orc_hash *hash;
} orc_db;

/* The hash-map form of the database (see orc_db.hash); built from the CSR arrays. */
struct orc_hash *orc_hash_create(const orc_db *db);
/* ... from keys[n_present] (ascending) + offsets[n_present + 1]: no array per possible code; db->offsets unused */
struct orc_hash *orc_hash_create_sparse(const orc_db *db, const uint32_t *keys, const uint64_t *offsets, uint64_t n_present);
void orc_hash_destroy(struct orc_hash *h);

/* Scratch for one thread: the per-thread arrays of placer (place.h:126-137). */
typedef struct orc_scratch orc_scratch;
orc_scratch *orc_scratch_create(const orc_db *db);
void orc_scratch_destroy(orc_scratch *s);

/* Places one read.  rows must hold keep_at_most entries; counts (nullable)
 * receives placement::count for each row.  Returns the number of rows, or -1
 * when len < k (the reference underflows size_t at place.cpp:322 and has no
 * defined behaviour; both oracle and product report "no placement"). */
int orc_place_read(const orc_db *db, orc_scratch *s, const char *seq, size_t len,
                   orc_row *rows, uint32_t *counts);

/* Places n reads given as a concatenated byte buffer + offsets[n+1].
 * rows: n * keep_at_most, n_rows: n (0 for too-short reads), counts nullable.
 * num_threads mirrors `#pragma omp parallel for schedule(dynamic)
 * num_threads(j)` of place.cpp:218-230 (1 = plain loop).  Returns 0. */
int orc_place_batch(const orc_db *db, const char *seqs, const uint64_t *seq_offsets,
                    uint64_t n, int num_threads, orc_row *rows, uint32_t *n_rows,
                    uint32_t *counts);

/* placer::place (place.cpp:201-275) as the driver calls it (main.cpp:332-344): the reads in batches
 * of batch_size (--batch-size, 2000), every batch de-duplicated by sequence content (:73-81, 207-212),
 * its unique sequences placed in an OpenMP dynamic loop (:218-230), duplicates sharing the result.
 * Same outputs as orc_place_batch. */
int orc_place_batched(const orc_db *db, const char *seqs, const uint64_t *seq_offsets, uint64_t n,
                      uint64_t batch_size, int num_threads, orc_row *rows, uint32_t *n_rows,
                      uint32_t *counts);

/* Algorithmic bytes of SURVEY.md 8(d): L + 8*n_kmers + 8*sum|list| + 16*rows. */
uint64_t orc_algorithmic_bytes(const orc_db *db, const char *seq, size_t len,
                               uint32_t rows_out);

/* Number of OpenMP threads the library would use for num_threads<=0 (all cores). */
int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
