/*
 * epik_oracle.c -- CPU restatement of EPIK's per-read placement loop.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see epik_oracle.h).
 *
 * Every function cites the reference lines it follows
 * (/root/reference/epik/src/epik/place.cpp unless another file is named).
 * Numerics follow the GCC build of the reference (the canonical one: bioconda,
 * CI ubuntu): epik::impl::pow is double std::pow (place.cpp:46), scores are
 * float32, x86-64 baseline has no FMA -> build with -ffp-contract=off.
 */
#include "epik_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* Per-thread arrays of epik::placer (place.h:126-137, place.cpp:92-96). */
struct orc_scratch {
    uint32_t n;           /* num_branches */
    float *scores;        /* _scores[thread]      */
    float *scores_amb;    /* _scores_amb[thread]  */
    uint64_t *counts;     /* _counts[thread]   (size_t in the reference) */
    uint64_t *counts_amb; /* _counts_amb[thread] */
    uint32_t *edges;      /* _edges[thread], insertion order */
    size_t n_edges;
    uint32_t *l_amb;      /* std::unordered_set l_amb of place.cpp:378, kept in insertion order */
    size_t n_lamb;
    /* query_kmers results (place.cpp:284-289): CSR ranges instead of optionals */
    uint64_t *ex_start; /* the lists' postings (pointers) */
    uint64_t *ex_len;
    size_t ex_cap, n_ex;
    uint64_t *am_start;
    uint64_t *am_len; /* 0 = key not found (search returned empty optional) */
    size_t am_cap, n_am;
};

orc_scratch *orc_scratch_create(const orc_db *db)
{
    orc_scratch *s = (orc_scratch *)calloc(1, sizeof(*s));
    if (!s) return NULL;
    const size_t n = db->num_branches ? db->num_branches : 1;
    s->n = db->num_branches;
    s->scores = (float *)calloc(n, sizeof(float));
    s->scores_amb = (float *)calloc(n, sizeof(float));
    s->counts = (uint64_t *)calloc(n, sizeof(uint64_t));
    s->counts_amb = (uint64_t *)calloc(n, sizeof(uint64_t));
    s->edges = (uint32_t *)calloc(n, sizeof(uint32_t));
    s->l_amb = (uint32_t *)calloc(n, sizeof(uint32_t));
    if (!s->scores || !s->scores_amb || !s->counts || !s->counts_amb || !s->edges || !s->l_amb) {
        orc_scratch_destroy(s);
        return NULL;
    }
    return s;
}

void orc_scratch_destroy(orc_scratch *s)
{
    if (!s) return;
    free(s->scores);
    free(s->scores_amb);
    free(s->counts);
    free(s->counts_amb);
    free(s->edges);
    free(s->l_amb);
    free(s->ex_start);
    free(s->ex_len);
    free(s->am_start);
    free(s->am_len);
    free(s);
}

/* ---- phylo_kmer_db::search (place.cpp:300,311): direct index, or the hash map of orc_db.hash ---- */
typedef struct orc_node {
    struct orc_node *next;
    uint64_t key;
    orc_pkdb_value *begin; /* its own allocation, like std::vector<pkdb_value> */
    uint64_t len;
} orc_node;
struct orc_hash {
    orc_node **buckets;
    uint64_t n_buckets;
    int borrowed; /* the nodes' lists lie in the caller's values[] (orc_hash_create_sparse) */
};

static int is_prime(uint64_t x)
{
    if (x < 2) return 0;
    for (uint64_t d = 2; d * d <= x; ++d)
        if (x % d == 0) return 0;
    return 1;
}

struct orc_hash *orc_hash_create(const orc_db *db)
{
    uint64_t n_present = 0;
    for (uint64_t k = 0; k < db->num_keys; ++k) n_present += db->offsets[k + 1] > db->offsets[k];
    struct orc_hash *h = (struct orc_hash *)calloc(1, sizeof(*h));
    if (!h) return NULL;
    h->n_buckets = n_present + 1; /* load factor <= 1, a prime bucket count, key % buckets: libstdc++'s policy */
    while (!is_prime(h->n_buckets)) ++h->n_buckets;
    h->buckets = (orc_node **)calloc(h->n_buckets, sizeof(orc_node *));
    if (!h->buckets) {
        free(h);
        return NULL;
    }
    for (uint64_t k = 0; k < db->num_keys; ++k) {
        const uint64_t b = db->offsets[k], e = db->offsets[k + 1];
        if (e == b) continue;
        orc_node *node = (orc_node *)malloc(sizeof(*node));
        orc_pkdb_value *v = (orc_pkdb_value *)malloc((size_t)(e - b) * sizeof(*v));
        if (!node || !v) {
            free(node);
            free(v);
            orc_hash_destroy(h);
            return NULL;
        }
        memcpy(v, db->values + b, (size_t)(e - b) * sizeof(*v));
        node->key = k;
        node->begin = v;
        node->len = e - b;
        node->next = h->buckets[k % h->n_buckets];
        h->buckets[k % h->n_buckets] = node;
    }
    return h;
}

/* The same map from the sparse form of a database -- keys[n_present] ascending, offsets[n_present + 1] into
 * db->values -- for key spaces where an offset per POSSIBLE code is out of reach (amino k = 7: 10 GB).  The lists
 * stay where they are (db->values must outlive the map); db->offsets is not read and may be NULL. */
struct orc_hash *orc_hash_create_sparse(const orc_db *db, const uint32_t *keys, const uint64_t *offsets, uint64_t n_present)
{
    struct orc_hash *h = (struct orc_hash *)calloc(1, sizeof(*h));
    if (!h) return NULL;
    h->borrowed = 1;
    h->n_buckets = n_present + 1;
    while (!is_prime(h->n_buckets)) ++h->n_buckets;
    h->buckets = (orc_node **)calloc(h->n_buckets, sizeof(orc_node *));
    if (!h->buckets) {
        free(h);
        return NULL;
    }
    for (uint64_t i = 0; i < n_present; ++i) {
        if (offsets[i + 1] == offsets[i]) continue;
        orc_node *node = (orc_node *)malloc(sizeof(*node));
        if (!node) {
            orc_hash_destroy(h);
            return NULL;
        }
        node->key = keys[i];
        node->begin = (orc_pkdb_value *)(uintptr_t)(db->values + offsets[i]);
        node->len = offsets[i + 1] - offsets[i];
        node->next = h->buckets[node->key % h->n_buckets];
        h->buckets[node->key % h->n_buckets] = node;
    }
    return h;
}

void orc_hash_destroy(struct orc_hash *h)
{
    if (!h) return;
    for (uint64_t i = 0; h->buckets && i < h->n_buckets; ++i) {
        orc_node *node = h->buckets[i];
        while (node) {
            orc_node *next = node->next;
            if (!h->borrowed) free(node->begin);
            free(node);
            node = next;
        }
    }
    free(h->buckets);
    free(h);
}

/* -> the list's postings and length (0: the key has none) */
static inline uint64_t orc_search(const orc_db *db, uint64_t key, const orc_pkdb_value **list)
{
    if (db->hash) {
        for (const orc_node *node = db->hash->buckets[key % db->hash->n_buckets]; node; node = node->next)
            if (node->key == key) {
                *list = node->begin;
                return node->len;
            }
        *list = NULL;
        return 0;
    }
    *list = db->values + db->offsets[key];
    return db->offsets[key + 1] - db->offsets[key];
}

static int grow(uint64_t **a, uint64_t **b, size_t *cap, size_t need)
{
    if (need <= *cap) return 0;
    size_t nc = *cap ? *cap : 256;
    while (nc < need) nc *= 2;
    uint64_t *na = (uint64_t *)realloc(*a, nc * sizeof(uint64_t));
    if (!na) return -1;
    *a = na;
    uint64_t *nb = (uint64_t *)realloc(*b, nc * sizeof(uint64_t));
    if (!nb) return -1;
    *b = nb;
    *cap = nc;
    return 0;
}

/*
 * i2l::to_kmers<i2l::one_ambiguity_policy>(seq, k), one window (place.cpp:293-294).
 * ASSUMPTION (i2l absent): the key is the base-sigma number of the k state codes,
 * first character most significant; a window is yielded iff it has no invalid
 * character and at most one ambiguous one (comment at place.cpp:293).
 * Returns 0 = not yielded, 1 = one key (*key), 2 = ambiguous: *key holds the key
 * with state 0 at the ambiguous position, *amb_mask the admissible states and
 * *amb_weight = sigma^(k-1-pos) so that key(state) = *key + state * *amb_weight.
 */
static int orc_kmer_window(const orc_db *db, const char *w, uint64_t *key, uint32_t *amb_mask,
                           uint64_t *amb_weight)
{
    const uint32_t k = db->kmer_size;
    const uint64_t sigma = db->alphabet_size;
    uint64_t code = 0;
    uint32_t n_amb = 0;
    uint32_t amb_pos = 0;
    for (uint32_t j = 0; j < k; ++j) {
        const uint32_t cls = db->char_class[(unsigned char)w[j]];
        if (cls == 0) return 0;
        uint32_t state = 0;
        if (cls & (cls - 1)) {
            if (++n_amb > 1) return 0;
            amb_pos = j;
            *amb_mask = cls;
        } else {
            while (!((cls >> state) & 1u)) ++state;
        }
        code = code * sigma + state;
    }
    *key = code;
    if (n_amb == 0) return 1;
    uint64_t wgt = 1;
    for (uint32_t j = amb_pos + 1; j < k; ++j) wgt *= sigma;
    *amb_weight = wgt;
    return 2;
}

/* query_kmers (place.cpp:278-316): exact hits keep only found keys (:301-304);
 * every resolved key of an ambiguous k-mer gets its own entry, found or not (:308-312). */
static int orc_query_kmers(const orc_db *db, orc_scratch *s, const char *seq, size_t num_kmers)
{
    s->n_ex = 0;
    s->n_am = 0;
    for (size_t p = 0; p < num_kmers; ++p) {
        uint64_t key = 0, wgt = 0;
        uint32_t mask = 0;
        const int kind = orc_kmer_window(db, seq + p, &key, &mask, &wgt);
        const orc_pkdb_value *list;
        if (kind == 1) {
            const uint64_t len = orc_search(db, key, &list);
            if (len) {
                if (grow(&s->ex_start, &s->ex_len, &s->ex_cap, s->n_ex + 1)) return -1;
                s->ex_start[s->n_ex] = (uint64_t)(uintptr_t)list;
                s->ex_len[s->n_ex] = len;
                ++s->n_ex;
            }
        } else if (kind == 2) {
            for (uint32_t st = 0; st < db->alphabet_size; ++st) {
                if (!((mask >> st) & 1u)) continue;
                const uint64_t len = orc_search(db, key + (uint64_t)st * wgt, &list);
                if (grow(&s->am_start, &s->am_len, &s->am_cap, s->n_am + 1)) return -1;
                s->am_start[s->n_am] = (uint64_t)(uintptr_t)list;
                s->am_len[s->n_am] = len;
                ++s->n_am;
            }
        }
    }
    return 0;
}

/* placer::place_seq (place.cpp:320-440) up to and including the score correction
 * (:418-422).  On return s->edges[0..n_edges) are the touched branches in
 * insertion order, s->scores / s->counts hold their corrected scores / counts. */
static int orc_place_seq(const orc_db *db, orc_scratch *s, const char *seq, size_t len)
{
    const size_t k = db->kmer_size;
    const size_t num_kmers = len - k + 1; /* :322 */

    /* sparse reset (:335-342) */
    for (size_t i = 0; i < s->n_edges; ++i) {
        const uint32_t e = s->edges[i];
        s->counts[e] = 0;
        s->scores[e] = 0.0f;
        s->counts_amb[e] = 0;
        s->scores_amb[e] = 0.0f;
    }
    s->n_edges = 0;

    if (orc_query_kmers(db, s, seq, num_kmers)) return -1; /* :345 */

    /* exact k-mers (:349-371): k-mer order, then posting order, float32 adds */
    for (size_t i = 0; i < s->n_ex; ++i) {
        const orc_pkdb_value *v = (const orc_pkdb_value *)(uintptr_t)s->ex_start[i];
        const uint64_t n = s->ex_len[i];
        for (uint64_t j = 0; j < n; ++j) {
            const uint32_t b = v[j].branch;
            if (s->counts[b] == 0) s->edges[s->n_edges++] = b; /* :360-363 */
            ++s->counts[b];                                    /* :365 */
            s->scores[b] += v[j].score;                        /* :366 */
        }
    }

    /* ambiguous k-mers (:373-415), one resolved key at a time */
    for (size_t i = 0; i < s->n_am; ++i) {
        s->n_lamb = 0; /* l_amb is local to each ambiguous_result (:378) */
        if (s->am_len[i] == 0) continue; /* if (exact_result) (:381) */
        const orc_pkdb_value *v = (const orc_pkdb_value *)(uintptr_t)s->am_start[i];
        const uint64_t n = s->am_len[i];
        for (uint64_t j = 0; j < n; ++j) {
            const uint32_t b = v[j].branch;
            if (s->counts_amb[b] == 0) s->l_amb[s->n_lamb++] = b; /* :385-388 */
            s->counts_amb[b] += 1;                                /* :390 */
            /* std::pow(10, score): int,float -> double pow, then cast (:391) */
            s->scores_amb[b] += (float)pow(10.0, (double)v[j].score);
        }
        const size_t w_size = k; /* :395 (k, not the number of resolved keys) */
        for (size_t j = 0; j < s->n_lamb; ++j) {
            const uint32_t b = s->l_amb[j];
            /* :400-402, all float32: (amb + float(w - c) * threshold) / float(w) */
            const float average_prob =
                (s->scores_amb[b] + (float)(w_size - s->counts_amb[b]) * db->threshold) /
                (float)w_size;
            if (s->counts[b] == 0) s->edges[s->n_edges++] = b; /* :404-407 */
            s->counts[b] += 1;                                 /* :409 */
            s->scores[b] += average_prob;                      /* :410 */
        }
    }

    /* score correction (:418-422) */
    for (size_t i = 0; i < s->n_edges; ++i) {
        const uint32_t e = s->edges[i];
        s->scores[e] += (float)(num_kmers - s->counts[e]) * db->log_threshold;
        s->scores[e] /= (float)k;
    }
    return 0;
}

/* placer::sum_scores (place.cpp:164-184), GCC flavour: double pow, double sum. */
static double orc_sum_scores(const orc_db *db, const orc_scratch *s, size_t len)
{
    const float num_branches = (float)db->num_branches;            /* :166 */
    const float num_placements = (float)s->n_edges;                /* :167 */
    const float num_kmers = (float)(len - db->kmer_size + 1);      /* :168 */
    const float kmer_size = (float)db->kmer_size;                  /* :169 */
    /* :174-175: (float - float) * pow(10.0, double(float expr)) */
    const double sum_not_placed = (double)(num_branches - num_placements) *
                                  pow(10.0, (double)(num_kmers * db->log_threshold / kmer_size));
    double sum_placed = 0.0; /* :178 */
    for (size_t i = 0; i < s->n_edges; ++i) {
        sum_placed += pow(10.0, (double)s->scores[s->edges[i]]); /* :181 */
    }
    return sum_not_placed + sum_placed; /* :183 */
}

int orc_place_read(const orc_db *db, orc_scratch *s, const char *seq, size_t len, orc_row *rows,
                   uint32_t *counts)
{
    if (len < db->kmer_size) return -1;
    if (orc_place_seq(db, s, seq, len)) return -2;

    const size_t num_kmers = len - db->kmer_size + 1;     /* :239 */
    const double score_sum = orc_sum_scores(db, s, len);  /* :238, before top-k */
    double keep_factor = db->keep_factor;                 /* :232 */

    /* select_best_placements (:134-159).  std::partial_sort leaves the order of
     * equal scores unspecified; the restatement (and the product) fix it as
     * (score descending, branch ascending). */
    size_t return_size = db->keep_at_most < s->n_edges ? db->keep_at_most : s->n_edges; /* :137 */
    size_t n_rows = 0;
    if (return_size == 0) {
        /* :141-152: no k-mer found -> first keep_at_most branches at the threshold score */
        const float threshold_score =
            db->log_threshold * (float)num_kmers / (float)db->kmer_size; /* :146-147 */
        for (size_t i = 0; i < db->keep_at_most; ++i) {
            rows[i].branch = (uint32_t)i;
            rows[i].score = threshold_score;
            rows[i].lwr = 0.0;
            if (counts) counts[i] = 0;
        }
        n_rows = db->keep_at_most;
    } else {
        /* top-return_size by repeated selection over the touched branches */
        float prev_score = INFINITY;
        uint32_t prev_branch = 0;
        for (size_t r = 0; r < return_size; ++r) {
            int have = 0;
            float best_score = 0.0f;
            uint32_t best_branch = 0;
            for (size_t i = 0; i < s->n_edges; ++i) {
                const uint32_t e = s->edges[i];
                const float sc = s->scores[e];
                /* strictly after (prev_score, prev_branch) in (score desc, branch asc) */
                if (r > 0 && !(sc < prev_score || (sc == prev_score && e > prev_branch))) continue;
                if (!have || sc > best_score || (sc == best_score && e < best_branch)) {
                    have = 1;
                    best_score = sc;
                    best_branch = e;
                }
            }
            rows[r].branch = best_branch;
            rows[r].score = best_score;
            rows[r].lwr = 0.0;
            if (counts) counts[r] = (uint32_t)s->counts[best_branch];
            prev_score = best_score;
            prev_branch = best_branch;
        }
        n_rows = return_size;
    }

    /* LWR (:241-264) */
    for (size_t r = 0; r < n_rows; ++r) {
        if (score_sum == 0) { /* :247-251 */
            rows[r].lwr = 0.0;
            keep_factor = 0.0;
        } else {
            const double power = pow(10.0, (double)rows[r].score); /* :254 */
            rows[r].lwr = (power == 0.0) ? 0.0 : power / score_sum; /* :255-262 */
        }
    }

    /* filter_by_ratio (:188-199); rows are sorted, rows[0] is the best */
    const double best_ratio = n_rows ? rows[0].lwr : 0.0;
    const double ratio_threshold = best_ratio * keep_factor;
    size_t kept = 0;
    for (size_t r = 0; r < n_rows; ++r) {
        if (rows[r].lwr >= ratio_threshold) {
            rows[kept] = rows[r];
            if (counts) counts[kept] = counts[r];
            ++kept;
        }
    }
    return (int)kept;
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* placer::place (place.cpp:201-275) without the dedup (:207-212), which the
 * host does before the boundary: OpenMP dynamic loop over reads, per-thread scratch. */
int orc_place_batch(const orc_db *db, const char *seqs, const uint64_t *seq_offsets, uint64_t n,
                    int num_threads, orc_row *rows, uint32_t *n_rows, uint32_t *counts)
{
    if (num_threads <= 0) num_threads = orc_max_threads();
    int failed = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(num_threads)
#endif
    {
        orc_scratch *s = orc_scratch_create(db);
        if (!s) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
            failed = 1;
        }
#ifdef _OPENMP
#pragma omp for schedule(dynamic)
#endif
        for (int64_t i = 0; i < (int64_t)n; ++i) {
            if (!s) continue;
            const uint64_t b = seq_offsets[i], e = seq_offsets[i + 1];
            const int r = orc_place_read(db, s, seqs + b, (size_t)(e - b), rows + i * db->keep_at_most,
                                         counts ? counts + i * db->keep_at_most : NULL);
            n_rows[i] = r > 0 ? (uint32_t)r : 0;
            if (r == -2) { /* the scratch could not grow: the call fails, it does not report "no placement" */
#ifdef _OPENMP
#pragma omp atomic write
#endif
                failed = 1;
            }
        }
        orc_scratch_destroy(s);
    }
    return failed ? -1 : 0;
}

/* group_by_sequence_content (place.cpp:73-81) of one batch: first[i] = index of the first read of the
 * batch with read i's content.  Open addressing over FNV-1a of the bytes. */
static int dedup_batch(const char *seqs, const uint64_t *offs, uint64_t n, uint64_t *first)
{
    uint64_t cap = 16;
    while (cap < 2 * n) cap *= 2;
    uint64_t *table = (uint64_t *)malloc(cap * sizeof(uint64_t));
    if (!table) return -1;
    for (uint64_t i = 0; i < cap; ++i) table[i] = UINT64_MAX;
    for (uint64_t i = 0; i < n; ++i) {
        const char *s = seqs + offs[i];
        const uint64_t len = offs[i + 1] - offs[i];
        uint64_t h = 1469598103934665603ull;
        for (uint64_t j = 0; j < len; ++j) h = (h ^ (unsigned char)s[j]) * 1099511628211ull;
        uint64_t slot = h & (cap - 1);
        for (;;) {
            const uint64_t other = table[slot];
            if (other == UINT64_MAX) {
                table[slot] = i;
                first[i] = i;
                break;
            }
            if (offs[other + 1] - offs[other] == len && memcmp(seqs + offs[other], s, len) == 0) {
                first[i] = other;
                break;
            }
            slot = (slot + 1) & (cap - 1);
        }
    }
    free(table);
    return 0;
}

int orc_place_batched(const orc_db *db, const char *seqs, const uint64_t *seq_offsets, uint64_t n,
                      uint64_t batch_size, int num_threads, orc_row *rows, uint32_t *n_rows, uint32_t *counts)
{
    if (num_threads <= 0) num_threads = orc_max_threads();
    if (batch_size == 0) batch_size = 2000; /* main.cpp:214 */
    uint64_t *first = (uint64_t *)malloc((batch_size + 1) * sizeof(uint64_t));
    uint64_t *unique = (uint64_t *)malloc((batch_size + 1) * sizeof(uint64_t));
    orc_scratch **scratch = (orc_scratch **)calloc((size_t)num_threads, sizeof(orc_scratch *));
    int failed = !first || !unique || !scratch;
    for (int t = 0; !failed && t < num_threads; ++t) failed = (scratch[t] = orc_scratch_create(db)) == NULL;
    const uint32_t keep = db->keep_at_most;
    for (uint64_t b0 = 0; !failed && b0 < n; b0 += batch_size) {
        const uint64_t m = n - b0 < batch_size ? n - b0 : batch_size;
        if (dedup_batch(seqs, seq_offsets + b0, m, first)) {
            failed = 1;
            break;
        }
        uint64_t n_unique = 0;
        for (uint64_t i = 0; i < m; ++i)
            if (first[i] == i) unique[n_unique++] = i;
        int batch_failed = 0; /* a read whose scratch could not grow: the whole call fails, as orc_place_batch */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic) num_threads(num_threads) reduction(| : batch_failed)
#endif
        for (int64_t u = 0; u < (int64_t)n_unique; ++u) {
#ifdef _OPENMP
            orc_scratch *s = scratch[omp_get_thread_num()];
#else
            orc_scratch *s = scratch[0];
#endif
            const uint64_t i = b0 + unique[u];
            const int r = orc_place_read(db, s, seqs + seq_offsets[i], (size_t)(seq_offsets[i + 1] - seq_offsets[i]),
                                         rows + i * keep, counts ? counts + i * keep : NULL);
            n_rows[i] = r > 0 ? (uint32_t)r : 0;
            if (r == -2) batch_failed |= 1; /* (-1: a read shorter than k, no placement) */
        }
        if (batch_failed) {
            failed = 1;
            break;
        }
        for (uint64_t i = 0; i < m; ++i) { /* the headers of a duplicate share the placement (jplace "nm") */
            if (first[i] == i) continue;
            const uint64_t src = b0 + first[i], dst = b0 + i;
            memcpy(rows + dst * keep, rows + src * keep, keep * sizeof(orc_row));
            if (counts) memcpy(counts + dst * keep, counts + src * keep, keep * sizeof(uint32_t));
            n_rows[dst] = n_rows[src];
        }
    }
    for (int t = 0; scratch && t < num_threads; ++t) orc_scratch_destroy(scratch[t]);
    free(scratch);
    free(first);
    free(unique);
    return failed ? -1 : 0;
}

uint64_t orc_algorithmic_bytes(const orc_db *db, const char *seq, size_t len, uint32_t rows_out)
{
    if (len < db->kmer_size) return len;
    const size_t num_kmers = len - db->kmer_size + 1;
    uint64_t entries = 0;
    for (size_t p = 0; p < num_kmers; ++p) {
        uint64_t key = 0, wgt = 0;
        uint32_t mask = 0;
        const int kind = orc_kmer_window(db, seq + p, &key, &mask, &wgt);
        if (kind == 1) {
            entries += db->offsets[key + 1] - db->offsets[key];
        } else if (kind == 2) {
            for (uint32_t st = 0; st < db->alphabet_size; ++st) {
                if ((mask >> st) & 1u) {
                    const uint64_t kk = key + (uint64_t)st * wgt;
                    entries += db->offsets[kk + 1] - db->offsets[kk];
                }
            }
        }
    }
    return (uint64_t)len + 8ull * num_kmers + 8ull * entries + 16ull * rows_out;
}
