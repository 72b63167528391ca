/*
 * epik_amd_stub.c -- TEST-ONLY stand-in for libepik_amd.so, for running the host driver under
 * ThreadSanitizer / AddressSanitizer on a machine without a GPU (tests/test_sanitizers_cpu.py).
 * It computes nothing: every read gets canned rows derived from its length.  It is never built
 * into, linked by or loaded from the product (epik_amd/); the product has no CPU path.
 */
#include <stdlib.h>
#include <string.h>

#include "epik_amd.h"

struct epik_amd_placer {
    uint32_t kmer_size, num_branches, keep_at_most;
};

int epik_amd_device_count(void) { return 2; }
const char *epik_amd_last_error(void) { return "stub"; }

int epik_amd_placer_create(const epik_amd_placer_desc *desc, epik_amd_placer **out)
{
    epik_amd_placer *p = (epik_amd_placer *)malloc(sizeof(*p));
    if (!p) return EPIK_AMD_ERR_INVALID;
    p->kmer_size = desc->kmer_size;
    p->num_branches = desc->num_branches;
    p->keep_at_most = desc->keep_at_most;
    *out = p;
    return EPIK_AMD_OK;
}

void epik_amd_placer_destroy(epik_amd_placer *p) { free(p); }

int epik_amd_placer_place(epik_amd_placer *p, const char *seqs, const uint64_t *seq_offsets, uint64_t n,
                          epik_amd_placement *rows, uint32_t *n_rows, uint32_t *kmer_counts)
{
    (void)seqs;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t len = seq_offsets[i + 1] - seq_offsets[i];
        const uint32_t count = len < p->kmer_size ? 0u : (p->keep_at_most < 3 ? p->keep_at_most : 3u);
        n_rows[i] = count;
        for (uint32_t r = 0; r < count; ++r) {
            rows[i * p->keep_at_most + r].branch = (uint32_t)((len + r) % p->num_branches);
            rows[i * p->keep_at_most + r].score = -(float)len / 100.0f - (float)r;
            rows[i * p->keep_at_most + r].lwr = 1.0 / (double)(r + 2);
            if (kmer_counts) kmer_counts[i * p->keep_at_most + r] = 1;
        }
    }
    return EPIK_AMD_OK;
}

/* the sharded forms the driver's --db-shard path calls: the same canned rows (what matters under the
 * sanitizers is the host side around the call) */
int epik_amd_placer_create_sharded(const epik_amd_placer_desc *desc, uint32_t shard_index, uint32_t shard_count,
                                   epik_amd_placer **out)
{
    if (shard_count == 0 || shard_index >= shard_count) return EPIK_AMD_ERR_INVALID;
    return epik_amd_placer_create(desc, out);
}

/* (the driver asks, right after creating shard 0, whether the tree's kernels leave partial lists) */
int epik_amd_placer_partial_info(const epik_amd_placer *p, epik_amd_partial_info *out)
{
    (void)p;
    memset(out, 0, sizeof *out);
    out->lists = 1;
    out->slices = 4;
    out->entry_bytes = 8;
    return EPIK_AMD_OK;
}

int epik_amd_placer_place_sharded(epik_amd_placer *const *shards, uint32_t n_shards, const char *seqs,
                                  const uint64_t *seq_offsets, uint64_t n, epik_amd_placement *rows, uint32_t *n_rows,
                                  uint32_t *kmer_counts)
{
    if (n_shards == 0) return EPIK_AMD_ERR_INVALID;
    return epik_amd_placer_place(shards[0], seqs, seq_offsets, n, rows, n_rows, kmer_counts);
}
