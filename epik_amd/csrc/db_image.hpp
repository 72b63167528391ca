// db_image.hpp -- from the caller's CSR database (epik_amd_placer_desc) to the device image,
// on the host, streaming.
//
// Replaces what i2l::load + the hash map behind phylo_kmer_db::search are to the reference
// (main.cpp:277, place.cpp:300): the posting lists re-laid for the kernels (db_layout.h).  The
// image is never materialised on the host: every part (table, presence filter, postings) is
// produced front to back, in k-mer-code order, into a Sink that hands out the next few bytes --
// pinned staging buffers on the way to HBM in create() (capi.hip), a checksum in the CPU tests.
// Host memory beyond the caller's own arrays: one uint32 per branch (validation) and the sinks'
// buffers.  Plain C++, no HIP.  Internal; the public boundary is include/epik_amd.h.
#ifndef EPIK_AMD_DB_IMAGE_HPP
#define EPIK_AMD_DB_IMAGE_HPP

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "db_layout.h"
#include "epik_amd.h"

namespace epik_amd::image {

// The caller's database, and which part of it this placer keeps (k-mer-space shard: the lists of the
// codes with code % shard_count == shard_index; 0 of 1 = everything).
struct Source {
    const epik_amd_placer_desc *d = nullptr;
    uint32_t shard_index = 0, shard_count = 1;
    bool sparse() const { return d->keys != nullptr; }
    // entry i of offsets[] (dense form: i = a k-mer code; sparse form: i = a position in keys[])
    uint64_t offset_at(uint64_t i) const
    {
        return d->offset_bits == 64 ? static_cast<const uint64_t *>(d->offsets)[i]
                                    : static_cast<const uint32_t *>(d->offsets)[i];
    }
    uint64_t offsets_len() const { return (sparse() ? d->num_present : d->num_keys) + 1; }
    bool kept(uint64_t key) const { return shard_count == 1 || key % shard_count == shard_index; }
};

// The lists of the database by k-mer code, for walks that ask for rising codes (every loop of the image
// builder does: the key space front to back, or a few such walks side by side for the tables keyed by a
// (k-1)-mer).  Dense form: an array lookup.  Sparse form: a position in keys[] that moves forward with the
// codes asked for -- amortised O(1); a code below the last one asked for is found by bisection.
class Cursor {
public:
    explicit Cursor(const Source &src) : _s(src) {}
    // the length of the list this placer keeps for `key` (0: none, or another shard's); *first = where it
    // begins in values[]
    uint64_t list(uint64_t key, uint64_t *first = nullptr)
    {
        if (!_s.kept(key)) return 0;
        if (!_s.sparse()) {
            const uint64_t b = _s.offset_at(key);
            if (first) *first = b;
            return _s.offset_at(key + 1) - b;
        }
        const uint32_t *keys = _s.d->keys;
        const uint64_t n = _s.d->num_present;
        if (_pos > 0 && _pos <= n && keys[_pos - 1] >= key) {  // backwards: bisect
            uint64_t lo = 0, hi = _pos;
            while (lo < hi) {
                const uint64_t mid = (lo + hi) / 2;
                if (keys[mid] < key) lo = mid + 1; else hi = mid;
            }
            _pos = lo;
        }
        // forward: gallop, then bisect inside the last step
        if (_pos < n && keys[_pos] < key) {
            uint64_t step = 1, lo = _pos;
            while (lo + step < n && keys[lo + step] < key) lo += step, step *= 2;
            uint64_t hi = lo + step < n ? lo + step : n;  // keys[hi] >= key or hi == n; keys[lo] < key
            ++lo;
            while (lo < hi) {
                const uint64_t mid = (lo + hi) / 2;
                if (keys[mid] < key) lo = mid + 1; else hi = mid;
            }
            _pos = lo;
        }
        if (_pos >= n || keys[_pos] != key) return 0;
        const uint64_t b = _s.offset_at(_pos);
        if (first) *first = b;
        return _s.offset_at(_pos + 1) - b;
    }

private:
    const Source &_s;
    uint64_t _pos = 0;  // sparse form: the first position of keys[] not below the last code asked for
};

// What create() decided and how large the parts are.
struct Plan {
    DbLayout layout = DbLayout::kCompact32;
    uint32_t n_pad = 0;  // one-wavefront-per-read kernels: LDS rows per wave (branches + the dummy row, to 64)
    // team kernel (layout == kTeam)
    int team_waves = 0;
    uint32_t team_passes = 0, team_slice_rows = 0, team_rows_pad = 0;
    uint64_t team_chunks = 0;  // chunks of <= 64 postings over all sublists (what a read's descriptors take: capi.hip)
    bool team_paired = false;  // the team table keyed by the (k-1)-mer two consecutive k-mers share (4 letters, 16-byte entries)
    std::vector<uint64_t> team_quarter_lines;  // paired: [pass][4] posting lines of the pass in front of each quarter of the key space
    // device image
    uint64_t table_bytes = 0, filter_bytes = 0, posting_bytes = 0;
    uint32_t filter_rec_bytes = 8;  // bytes of a presence record of the filtered layout: 8, or 5 (the 2 x 20 bits of a protein database, packed)
    uint64_t kept_entries = 0, present_codes = 0;
    bool runs = false;  // packed layouts: run-coded lists (place_device.hpp, kRuns)
    uint64_t quarter_lines[4] = {0, 0, 0, 0};  // paired table: posting lines in front of each quarter of the key space
    uint32_t wave_resident[3] = {0, 0, 0};     // resident waves per CU by count width (what chose the kernel)
    // k-mer-space shard (include/epik_amd.h): the placer keeps the lists of the codes with code % shard_count ==
    // shard_index.  The sliced table of such a placer holds an entry per code OF THE SHARD, at code / shard_count
    // (table_keys of them per pass): 1 / shard_count of the bytes, and the kernels test code % shard_count before
    // they fetch -- the other shards' k-mers of a read cost no table line.
    uint32_t shard_index = 0, shard_count = 1;
    uint64_t table_keys = 0;  // sliced layout: entries per pass
};

// codes of [0, num_keys) with code % count == index
inline uint64_t shard_keys(uint64_t num_keys, uint32_t index, uint32_t count)
{
    return count <= 1 ? num_keys : (num_keys > index ? (num_keys - index + count - 1) / count : 0);
}

// Sequential writer of one part of the image.
struct Sink {
    virtual ~Sink() = default;
    // `n` writable bytes that follow everything reserved before, zero-filled; valid until the next call.
    virtual uint8_t *reserve(size_t n) = 0;
};

// What part of a database the placer is to keep: the caller's (shard_index, shard_count), or -- with (0, 1) -- the shard
// the descriptor says it holds already (epik_amd_placer_desc.shard); the two must not disagree.
int resolve_shard(const epik_amd_placer_desc *d, uint32_t &shard_index, uint32_t &shard_count, std::string &err);

// Argument and consistency checks of create() that need no device: sizes, monotone offsets,
// branches in range and distinct inside a list, finite scores.  Returns an epik_amd_status.
int validate(const epik_amd_placer_desc *d, uint32_t shard_index, uint32_t shard_count, std::string &err);

// Chooses kernel and layout (forced_layout: EPIK_AMD_LAYOUT; forced_kernel: EPIK_AMD_KERNEL =
// wave | team | team4 | team8; either may be null) for a device with `free_mem` bytes free and sizes
// the parts.  Reads every offset, and for the team layout every posting, once.
int make_plan(const Source &src, size_t free_mem, const char *forced_layout, const char *forced_kernel, Plan &plan,
              std::string &err);

// The same plan from the SIZES of a database alone (epik_amd_placer_plan_sizes: capacity planning before a database
// exists or is at hand -- main.cpp:252-266's --max-ram is the reference's only capacity control): the tree, the key
// space, and how many lists of which length the placer (its shard) will keep.  Kernel, layout, geometry, table and
// filter come out as make_plan() would give them on such a database; the posting region exactly for the layouts of
// the one-wavefront kernels, as an upper bound (`posting_bytes_is_bound`) for the sliced layout.
struct SizeDesc {
    uint32_t kmer_size = 0, alphabet_size = 0, num_branches = 0, keep_at_most = 7;
    const epik_amd_list_bin *bins = nullptr;
    uint64_t n_bins = 0;
    uint32_t shard_index = 0, shard_count = 1;
};
int plan_sizes(const SizeDesc &sizes, size_t free_mem, const char *forced_layout, const char *forced_kernel, Plan &plan,
               bool &posting_bytes_is_bound, std::string &err);

// Produces the image: exactly plan.table_bytes into `table`, plan.filter_bytes into `filter` (may be
// null when 0), plan.posting_bytes into `postings`.
int build(const Source &src, const Plan &plan, Sink &table, Sink *filter, Sink &postings, std::string &err);

}  // namespace epik_amd::image
#endif
