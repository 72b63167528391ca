// place_kernel.h -- launch interface between the C-ABI layer (capi.hip) and the
// device kernels (place_kernel.hip).  Internal; the public boundary is include/epik_amd.h.
#ifndef EPIK_AMD_PLACE_KERNEL_H
#define EPIK_AMD_PLACE_KERNEL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "epik_amd.h"

#define EPIK_AMD_TILES_PER_PASS 3
#ifndef EPIK_AMD_RING
#define EPIK_AMD_RING 8  // posting-chunk loads kept in flight per wave (power of two)
#endif

namespace epik_amd {

// How the phylo-k-mer database is laid out in HBM (place_kernel.hip documents both).
enum class DbLayout : int {
    kCompact32 = 0,  // CSR: 32-bit offsets[num_keys + 1], 8-byte {f32 score, u32 cell} postings back to back
    kCompact64 = 1,  // same with 64-bit offsets
    kPacked = 2,     // 8-byte {len, first 128-byte line} entry per k-mer code, lists on whole lines,
                     // 6 bytes per posting: f32 score[cnt] then u16 cell[cnt] per chunk of <= 64
    kPaired = 3,     // the same lists; the table keyed by the (k-1)-mer two consecutive k-mers share
                     // (4-letter alphabets): one table line per two lookups, 16 bytes per code
    kFiltered = 4,   // kPacked behind a presence filter keyed the same way (other alphabets, sparse
                     // databases): one filter word per two lookups, the table only for present codes
};

// Kernel arguments: the database in HBM, the placer constants of place.cpp:83-96, and
// one batch of reads.
struct PlaceParams {
    const void *table;           // compact: OffT offsets[num_keys + 1]; packed: uint2 {len, line}[num_keys];
                                 // paired: uint2 {len, line}[num_keys / 4][8]
    const uint64_t *filter;      // filtered: [alphabet_size^(kmer_size-1)] presence words
    uint32_t sigma_pow_km1;      // alphabet_size^(kmer_size-1)
    const uint8_t *postings;     // scores + cells, cell = n_pad - 1 - branch
    const uint32_t *char_class;  // [256]
    const uint8_t *seqs;
    const uint64_t *seq_offsets; // [n_reads + 1]
    uint64_t n_reads;
    epik_amd_placement *rows;    // [n_reads * keep_at_most]
    uint32_t *n_rows;            // [n_reads]
    uint32_t *kmer_counts;       // [n_reads * keep_at_most] or null
    // k-mer-space shard: raw per-branch sums, [n_reads][num_branches].  Non-null in
    // place_reads_kernel = accumulate only (no epilogue); the input of finish_reads_kernel.
    float *partial_scores;
    uint32_t *partial_counts;
    uint32_t kmer_size;
    uint32_t alphabet_size;
    uint32_t num_branches;
    uint32_t keep_at_most;
    double keep_factor;
    float threshold;
    float log_threshold;
    uint32_t n_pad;                  // LDS rows per wave: num_branches + the dummy row, rounded up to 64
    uint32_t lds_wave_bytes;         // LDS bytes per wave (scores + counts + chunk descriptors)
    uint32_t ablate;                 // timing experiments only (-DEPIK_AMD_ABLATION builds)
    unsigned long long *dbg;         // phase cycle sums (-DEPIK_AMD_ABLATION builds, EPIK_AMD_STAMPS=1)
};

// Width of the per-branch k-mer counts in LDS: 16 bits by default (reads of up to 32767 k-mers),
// 32 for longer reads, 8 (reads of up to 255 k-mers) when that lets more waves share a CU.
enum CountBits : int { kCounts8 = 0, kCounts16 = 1, kCounts32 = 2 };

hipError_t launch_place_reads(const PlaceParams &p, DbLayout layout, int counts, dim3 grid, dim3 block,
                              size_t lds_bytes, hipStream_t stream);
hipError_t set_place_reads_lds_limit(DbLayout layout, int counts, size_t lds_bytes);
hipError_t place_reads_occupancy(DbLayout layout, int counts, int block_threads, size_t lds_bytes,
                                 int *blocks_per_cu);
hipError_t launch_finish_reads(const PlaceParams &p, int counts, dim3 grid, dim3 block, size_t lds_bytes,
                               hipStream_t stream);
hipError_t set_finish_reads_lds_limit(int counts, size_t lds_bytes);
hipError_t launch_algorithmic_bytes(const PlaceParams &p, DbLayout layout, unsigned long long *d_total,
                                    hipStream_t stream);

}  // namespace epik_amd
#endif
