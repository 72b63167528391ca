// placer_impl.hpp -- what a handle of the C ABI holds.  Internal to libepik_amd (capi.hip creates and launches,
// shard_place.hip drives several handles as the shards of one database); the public boundary is include/epik_amd.h.
#ifndef EPIK_AMD_PLACER_IMPL_HPP
#define EPIK_AMD_PLACER_IMPL_HPP
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "db_image.hpp"
#include "epik_amd.h"
#include "place_kernel.h"

struct epik_amd_placer {
    int device = 0;
    epik_amd::DbLayout layout = epik_amd::DbLayout::kCompact32;
    int counts = epik_amd::kCounts16;  // width of the per-branch counts the next device launch uses
    bool counts_forced = false;        // set by set_wide_counts / the environment: place() chooses only to WIDEN
    bool timing = false;
    void *d_table = nullptr;       // offsets (compact) or {len, line} entries (packed)
    uint64_t *d_filter = nullptr;  // presence words of the filtered layout
    uint8_t *d_postings = nullptr; // 6-byte postings
    uint64_t db_bytes = 0;
    uint32_t *d_char_class = nullptr;
    // buffers of epik_amd_placer_place_sharded, kept from call to call on the first handle of the set (shard_place.hip)
    // which create() made this handle, counted over the process: a set of handles kept from call to call
    // (shard_place.hip) knows a handle by it, not by an address that a later create() may be given again
    uint64_t generation = 0;
    void *shard_state = nullptr;
    void (*shard_state_free)(void *) = nullptr;
    std::vector<uint32_t> h_char_class;  // the same on the host (shard_place.hip: which reads may hold an ambiguous k-mer)
    epik_amd::PlaceParams params{};  // batch fields are filled per call
    epik_amd::image::Plan plan{};    // kernel, layout and sizes chosen at create()
    bool team = false;               // the team kernel (one workgroup per read) places; else one wavefront per read
    int team_waves = 0;
    const uint8_t *team_table = nullptr;
    uint32_t team_passes = 0, team_slice_rows = 0, team_rows_pad = 0;
    // the team placement as front kernel + streaming kernel (team_stream.hip); the scratch of a launch,
    // grown on demand: a header per read, the reads left to team_place_kernel, the descriptor pool
    bool team_front = false;
    uint8_t *d_front_hdr = nullptr;
    size_t front_hdr_bytes = 0;
    uint8_t *d_finish_hdr = nullptr;  // the headers of a FINISH launch (its own: the finish of one batch may run beside
    size_t finish_hdr_bytes = 0;      // the accumulate of the next on another stream, which writes d_front_hdr)
    uint64_t *d_slow_list = nullptr;
    size_t slow_list_reads = 0;
    void *d_slice_rows = nullptr, *d_slice_sums = nullptr;  // the slices' results on their way to the merge kernel
    size_t slice_out_reads = 0;
    uint64_t *d_front_pool = nullptr;
    uint64_t front_pool_cap = 0;                  // descriptors
    // the slice epilogue over the touched quads (team_epilogue.hpp): see TeamParams; EPIK_AMD_TEAM_SPARSE=0 (never) |
    // always (whatever was streamed, up to the list's capacity: tests on small trees) | <chunks> (that threshold)
    uint32_t sparse_chunks = 0, sparse_quads = 0;
    uint64_t front_pool_forced = 0;               // EPIK_AMD_TEAM_POOL: that many, whatever the batch (tests)
    unsigned long long *d_front_cursor = nullptr; // [0] descriptors asked for, [1] reads on the slow list, [2] reads of the launch
    unsigned long long *h_front_cursor = nullptr; // pinned: the same of the last launch that has finished
    uint64_t longest_read_hint = 0;               // epik_amd_placer_choose_counts
    uint32_t max_blocks_cap = 0;                  // tests: EPIK_AMD_MAX_BLOCKS
    uint32_t front_blocks = 0;                    // grid of the front kernel: the workgroups a device holds
    uint32_t merge_blocks = 0;                    // ... of the merge kernel (four waves each)
    uint64_t grid_percent = 0;                    // diagnostic builds: EPIK_AMD_GRID_PERCENT (of the resident workgroups; 0: as the product)
    uint64_t num_keys = 0;
    uint64_t num_entries = 0;
    // launch geometry per count width (epik_amd::CountBits)
    struct geometry {
        uint32_t waves_per_block = 4;
        uint32_t lds_wave_bytes = 0;
        uint32_t lds_block_bytes = 0;
        uint32_t max_blocks = 0;
        uint32_t resident_waves = 0;  // per CU
        // team_stream_kernel (team placement as front + streaming + merge kernels): workgroups of 4 waves -- [0], the
        // halves of a k-mer-space-sharded placement -- or of stream_bw[1] = 4 or 2 -- [1], the one-pass placement
        // (db_layout.h: stream_block_waves: two where that puts more waves on a CU)
        uint32_t stream_lds_bytes[2] = {0, 0}, stream_blocks[2] = {0, 0};
        int stream_bw[2] = {4, 4};
        uint32_t last_stream = 0;  // which of the two the last launch used
    } geo[3];
    uint32_t *d_sparse_cap = nullptr;             // partial lists: room per (read, slice), front kernel -> scan kernel
    unsigned long long *d_scan_tiles = nullptr;   // ... and the scan's tile sums
    size_t sparse_cap_items = 0;
    uint64_t front_failed_reads = 0;              // != 0: the scratch of the three-kernel placement could not be had for a
                                                  // launch of that many reads (not tried again for as many or more)
    uint32_t last_blocks = 0;
    bool last_streamed = false;  // the last launch went through team_stream_kernel
    uint32_t last_geo = 0;
    // staging buffers for the host-pointer entry point (grown on demand)
    uint8_t *d_seqs = nullptr;
    size_t d_seqs_cap = 0;
    uint64_t *d_seq_offsets = nullptr;
    epik_amd_placement *d_rows = nullptr;
    uint32_t *d_n_rows = nullptr;
    uint32_t *d_counts = nullptr;
    size_t d_reads_cap = 0;
    unsigned long long *d_total = nullptr;
    hipStream_t stream = nullptr;  // owned, for the synchronous entry point: kernels
    hipStream_t stream_in = nullptr, stream_out = nullptr;  // ... its copies in and out
    std::vector<hipEvent_t> ev_in, ev_kernel;               // per chunk of that entry point
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool ev_recorded = false;
};

namespace epik_amd {
// the message of the failing call, for epik_amd_last_error() (capi.hip)
int fail_with(int code, const std::string &msg);
}  // namespace epik_amd
#endif
