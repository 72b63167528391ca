// place_kernel.h -- launch interface between the C-ABI layer (capi.hip) and the
// device kernels (place_kernel.hip).  Internal; the public boundary is include/epik_amd.h.
#ifndef EPIK_AMD_PLACE_KERNEL_H
#define EPIK_AMD_PLACE_KERNEL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "epik_amd.h"

#include "db_layout.h"

namespace epik_amd {

// Kernel arguments: the database in HBM, the placer constants of place.cpp:83-96, and
// one batch of reads.
struct PlaceParams {
    const void *table;           // compact: OffT offsets[num_keys + 1]; packed: uint2 {len, line}[num_keys];
                                 // paired: uint2 {len, line}[num_keys / 4][8]
    const uint64_t *filter;      // filtered: [alphabet_size^(kmer_size-1)] presence records of filter_rec_bytes each
    uint32_t sigma_pow_km1;      // alphabet_size^(kmer_size-1)
    uint32_t filter_rec_bytes;   // 8: 64-bit words; 5: 40 bits packed (any byte: read as two dwords)
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
    uint16_t *partial_counts;        // (a read of a k-mer-space shard has at most 65535 k-mers)
    // ... and how its ambiguous k-mers cross the shards (place.cpp:385-388 holds over the whole
    // database): amb_slot[read] >= 0 names the read's row in amb_order / amb_avg,
    // [slots][num_branches]: accumulate records there, per branch, the order of the first ambiguous
    // key of ITS lists that reached the branch and that key's average probability; the caller keeps,
    // per branch, the record of smallest order over the shards; finish adds it.  Null: a shard
    // scores its ambiguous k-mers by itself (right with one shard only).
    const int32_t *amb_slot;
    uint32_t *amb_order;
    float *amb_avg;
    uint32_t kmer_size;
    uint32_t alphabet_size;
    uint32_t num_branches;
    uint32_t keep_at_most;
    double keep_factor;
    float threshold;
    float log_threshold;
    uint32_t max_kmers_cap;          // != 0: a read with more k-mers than this is not placed whatever the LDS counts hold
                                     // (the uint16 counts of the dense partial vectors: 65535)
    uint32_t n_pad;                  // LDS rows per wave: num_branches + the dummy row, rounded up to 64
    uint32_t lds_wave_bytes;         // LDS bytes per wave (scores + counts + chunk descriptors)
    uint32_t ablate;                 // timing experiments only (-DEPIK_AMD_ABLATION builds)
    unsigned long long *dbg;         // phase cycle sums (-DEPIK_AMD_ABLATION builds, EPIK_AMD_STAMPS=1)
};

// n_rows of a read with more k-mers than the launch's count width holds (EPIK_AMD_ROWS_COUNTS_TOO_NARROW)
constexpr uint32_t kCountsTooNarrow = 0xffffffffu;

// ---- team kernel (team_kernel.hip): one workgroup of W waves per read, the branch range in slices ----
struct TeamParams {
    PlaceParams base;            // first member: the out-of-line device functions get &base
                                 // (base.postings = the sliced posting region; base.table / filter unused)
    const uint8_t *team_table;   // [passes][num_keys] entries of team_entry_bytes(W); paired: [passes][num_keys / 4][8]
    uint64_t num_keys;           // entries per pass: the codes of the key space, or -- a k-mer-space shard -- those of the shard
    uint32_t shard_index, shard_count;  // shard_count > 1: the table holds the codes with code % shard_count == shard_index
                                 // only, code / shard_count being the entry's place; the other codes have no list HERE and
                                 // are not looked up
    uint32_t team_paired;        // the table is keyed by the (k-1)-mer X two consecutive k-mers share (4 letters): block X =
                                 // the entries of a.X (slots 0-3) and X.b (slots 4-7); a k-mer at an even position of the
                                 // read is looked up as a.X, the next one as X.b -- the same 128-byte line
    uint32_t passes;             // P: the tree is placed in P passes of W slices each
    uint32_t slice_rows;         // branches per slice; slice s = pass * W + wave starts at branch s * slice_rows
    uint32_t rows_pad;           // LDS rows per slice: slice_rows + the dummy row, rounded up to 64
    uint32_t desc_cap;           // chunk descriptors per slice and round (a multiple of the ring, >= 64)
    uint32_t slice_bytes;        // LDS bytes of one slice's rows (scores then counts), a multiple of 16
    uint32_t desc_bytes;         // LDS bytes of one slice's descriptor list, a multiple of 16
    // team_place_kernel over a list of reads (null: reads 0 .. base.n_reads - 1): the reads the front
    // kernel below could not prepare
    const uint64_t *read_list;
    const unsigned long long *read_list_count;
    // The front end as a kernel of its own (team_stream.hip).  team_front_kernel leaves, per read, a
    // header {u32 off8, u32 flags, u32 length, u32 cnt[S]} (S = W * passes slices; front_hdr_stride bytes
    // apart) and the chunk descriptors of every slice, in read order, in the pool: slice s of the read owns
    // the descriptors from off8 * 8 + sum of the earlier slices' counts (each rounded up to the ring, the
    // padding being null chunks) on.  front_cursor[0] = descriptors handed out so far (it runs past
    // front_pool_cap when the pool is too small: those reads go on slow_list, front_cursor[1] counts them),
    // front_cursor[2] = reads of the launch.
    uint8_t *front_hdr;
    uint32_t front_hdr_stride;
    uint64_t *front_pool;
    uint64_t front_pool_cap;
    unsigned long long *front_cursor;
    uint64_t *slow_list;
    // what the slices of a read hand to team_merge_kernel: [n_reads][S][keep_at_most] ranked rows
    // {ord(score), branch, k-mer count, -} (empty slots 0) and [n_reads][S] partial sums (TeamPartial)
    void *slice_rows_out;
    void *slice_sums_out;
    // The partial LISTS of a k-mer-space shard (include/epik_amd.h, "sparse partials"): instead of a dense
    // [num_branches] vector per read, the rows a read touched in each slice, compacted.  The front kernel leaves
    // sparse_cap[read][S] = how many entries the list of (read, slice) may take (the postings of the slice's
    // sublists, at most the slice's rows); team_sparse_scan_kernel turns them into sparse_index[read][S].x =
    // first entry of the list inside the read's PART (parts = the n_parts equal runs of sparse_part_reads reads
    // that go to one finisher each) and sparse_part_total[part]; the streaming kernel (the other kernel for the
    // reads left to it) writes the entries and sparse_index[..].y = how many.
    uint32_t *sparse_cap;
    uint2 *sparse_index;
    uint8_t *sparse_entries;
    uint64_t sparse_entries_cap;          // entries
    unsigned long long *sparse_part_total;  // [n_parts]
    uint32_t sparse_parts, sparse_part_reads;
    // The slice epilogue over the touched quads (team_epilogue.hpp): an item's touched quads are counted, and it takes
    // that epilogue when there are at most sparse_quads of them (0: never).  sparse_chunks: an item that streamed more
    // chunks than that is not even asked (0xffffffff: every item is).
    uint32_t sparse_chunks, sparse_quads;
};
// sparse_quads for a slice of rows_pad rows: up to four trips of 64 quads, and at most half the slice's trips (a
// list of more saves less than it costs)
constexpr uint32_t team_sparse_quads(uint32_t rows_pad)
{
    const uint32_t dense_trips = (rows_pad / 4u + 63u) / 64u;
    return 64u * (dense_trips / 2u < 4u ? dense_trips / 2u : 4u);
}
// An entry of a partial list: {f32 sum, u32 row | count << 16} with 8- and 16-bit counts, {f32 sum, u32 row,
// u32 count, 0} with 32-bit counts (reads of 32768 k-mers or more); row = branch - first branch of the slice.
constexpr uint32_t sparse_entry_bytes(int counts) { return counts == kCounts32 ? 16u : 8u; }
constexpr uint32_t kSparseOverflow = 0xffffffffu;  // sparse_index[..].y of a list that did not fit sparse_entries
constexpr int kMaxShards = 16;                     // EPIK_AMD_MAX_SHARDS
// what finish reads: the lists of its reads as every shard sent them (entries = start of the part's segment)
struct SparseSources {
    const uint8_t *entries[kMaxShards];
    const uint2 *index[kMaxShards];
    uint32_t n_shards;
};
// header flags
constexpr uint32_t kFrontAmbiguous = 1u;  // the read has an ambiguous k-mer (place.cpp:306-313)
constexpr uint32_t kFrontSlow = 2u;       // its descriptors are not in the pool: team_place_kernel places it
constexpr uint32_t kFrontNoRows = 4u;     // shorter than k: no placement
constexpr uint32_t kFrontTooNarrow = 8u;  // more k-mers than the launch's counts hold
// the header's words sit one per lane in the consumer
constexpr uint32_t kFrontHdrWords = 3;
constexpr uint32_t kFrontMaxSlices = 64 - kFrontHdrWords;
constexpr uint32_t front_hdr_stride(uint32_t slices) { return ((kFrontHdrWords + slices) * 4u + 15u) & ~15u; }
// a wave of the front kernel takes the pool this many descriptors at a time (one atomic add each) and hands
// them to its reads itself; a read that needs more takes exactly what it needs
constexpr uint32_t kFrontPoolChunk = 4096;
enum : int { kTeamModePlace = 0, kTeamModeAccumulate = 1, kTeamModeFinish = 2, kTeamModeAccumulateLists = 3,
             kTeamModeFinishLists = 4 };
hipError_t launch_team(const TeamParams &tp, int waves, int counts, int mode, dim3 grid, size_t lds_bytes,
                       hipStream_t stream);
hipError_t set_team_lds_limit(int waves, int counts, size_t lds_bytes);
hipError_t team_occupancy(int waves, int counts, size_t lds_bytes, int *blocks_per_cu);
// team_stream.hip: the front kernel (any grid of 256-thread workgroups, one read per wave) and the
// streaming kernel (one workgroup per read, grid = resident workgroups, LDS as the team kernel's)
hipError_t launch_team_front(const TeamParams &tp, int waves, int counts, bool lists, dim3 grid, hipStream_t stream);
// (bw: waves of a workgroup of the streaming kernel, 4, or 2 for the one-pass placement where that puts more waves on a CU)
hipError_t launch_team_stream(const TeamParams &tp, int waves, int counts, int mode, int bw, dim3 grid, size_t lds_bytes,
                              hipStream_t stream, const SparseSources *sources = nullptr);
// (tile_sums: scratch of sparse_scan_tiles(reads per part, slices) * parts 64-bit words)
uint64_t sparse_scan_tiles(uint64_t part_reads, uint32_t slices);
hipError_t launch_team_sparse_scan(const TeamParams &tp, int waves, unsigned long long *tile_sums, hipStream_t stream);
hipError_t launch_team_headers(const TeamParams &tp, int waves, int counts, hipStream_t stream);
hipError_t launch_team_merge(const TeamParams &tp, int waves, dim3 grid, hipStream_t stream);
constexpr size_t kTeamPartialBytes = 24;  // sizeof(TeamPartial) (place_device.hpp)
hipError_t set_team_stream_lds_limit(int waves, int counts, int mode, int bw, size_t lds_bytes);
bool team_stream_is_wide(int waves, size_t lds_bytes, int bw);  // the build launch_team_stream picks (EPIK_AMD_STREAM_WIDE overrides the rule)
hipError_t team_stream_occupancy(int waves, int counts, int mode, int bw, size_t lds_bytes, int *blocks_per_cu);
hipError_t launch_team_algorithmic_bytes(const TeamParams &tp, int waves, unsigned long long *d_total, hipStream_t stream);

hipError_t launch_place_reads(const PlaceParams &p, DbLayout layout, bool runs, int counts, dim3 grid, dim3 block,
                              size_t lds_bytes, hipStream_t stream);
hipError_t set_place_reads_lds_limit(DbLayout layout, bool runs, int counts, size_t lds_bytes);
hipError_t place_reads_occupancy(DbLayout layout, bool runs, int counts, int block_threads, size_t lds_bytes,
                                 int *blocks_per_cu);
hipError_t launch_finish_reads(const PlaceParams &p, int counts, dim3 grid, dim3 block, size_t lds_bytes,
                               hipStream_t stream);
hipError_t set_finish_reads_lds_limit(int counts, size_t lds_bytes);
hipError_t launch_algorithmic_bytes(const PlaceParams &p, DbLayout layout, bool runs, unsigned long long *d_total,
                                    hipStream_t stream);

}  // namespace epik_amd
#endif
