// db_image.cpp -- see db_image.hpp.
#include "db_image.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

namespace epik_amd::image {

namespace {

// Fixed-size records into a Sink, a batch at a time (one virtual call per 64 KiB, not per record).
class RecordWriter {
public:
    RecordWriter(Sink &sink, size_t record_bytes, uint64_t total_records)
        : _sink(sink), _record(record_bytes), _left(total_records) {}
    uint8_t *next()
    {
        if (_at == _end) {
            const uint64_t n = std::min<uint64_t>(_left, std::max<size_t>(1, (64u << 10) / _record));
            _at = _sink.reserve((size_t)n * _record);
            _end = _at + (size_t)n * _record;
            _left -= n;
        }
        uint8_t *r = _at;
        _at += _record;
        return r;
    }

private:
    Sink &_sink;
    size_t _record;
    uint64_t _left;
    uint8_t *_at = nullptr, *_end = nullptr;
};

inline void put_u32(uint8_t *dst, uint32_t v) { std::memcpy(dst, &v, 4); }

// One list of the packed layouts: chunks of <= 64 postings, each f32 score[cnt] then u16 cell[cnt]
// with cell = top - branch (top = the dummy row's cell 0 seen from the other end).
void write_chunks(uint8_t *dst, const epik_amd_pkdb_value *src, uint64_t len, uint32_t top)
{
    for (uint64_t c0 = 0; c0 < len; c0 += 64) {
        const uint32_t cnt = (uint32_t)((len - c0 < 64) ? len - c0 : 64);
        for (uint32_t j = 0; j < cnt; ++j) {
            const uint16_t cell = (uint16_t)(top - src[c0 + j].branch);
            std::memcpy(dst + 4u * j, &src[c0 + j].score, 4);
            std::memcpy(dst + 4u * cnt + 2u * j, &cell, 2);
        }
        dst += (size_t)cnt * 6u;
    }
}


// The run-coded form of the packed layouts (place_device.hpp, kRuns): a list whose branches are one ascending
// run b, b + 1, ... is stored without its cells -- 4 bytes per posting -- and its table entry carries the first
// cell.  Lists of 65536 postings or more (no tree of the one-wavefront kernels has that many branches) and lists
// that are not a run stay explicit.  EPIK_AMD_RUNS=0: every list explicit (measurements).
// Run-coding pays where the kernel is bound by what HBM delivers: measured (r03, N = 999, 1 M x 150 bp reads), the
// k = 11 database (1.13 GB explicit) gains 5-8 % from a third fewer lines per list, while the headline database
// (285 MB, served mostly by the 256 MiB Infinity Cache) loses 3 % to the three more instructions per chunk.  So:
// when the explicit image would be beyond twice the Infinity Cache.  EPIK_AMD_RUNS=0 / 1 forces it.
// Round 5: ... or when it brings the image from beyond the Infinity Cache to inside it, on a tree whose kernel keeps its
// full twenty waves on a CU -- there the kernel is bound by the memory system (0.885 of the roofline on the headline
// database, 285 MB explicit, 213 MB run-coded) and the lines not fetched win: 171.0 -> 177.6 M reads/s, 0.887 -> 0.921
// (round 3 had measured +1.2 % there, before the ring's stages lost an instruction).  With fewer waves (N = 1 303: 18,
// 1 499 ... 1 949: 16, 14) the kernel is bound by the instructions it issues and the run path's three more per chunk
// lose 4-10 % even though the image crosses under the cache (measured on the round's kernels: 165.7 -> 158.3, 159.1 ->
// 149.1, 143.1 -> 129.3, 140.5 -> 127.1 M reads/s): those keep their cells.
constexpr uint64_t kInfinityCacheBytes = 256ull << 20;
inline bool choose_runs(uint64_t explicit_image_bytes, uint64_t run_coded_image_bytes, uint64_t postings, uint64_t postings_in_runs,
                        bool full_occupancy)
{
    if (const char *e = std::getenv("EPIK_AMD_RUNS")) return e[0] != '0';
    // (... and nearly all postings in runs: the kernels with the run path wait longer than needed behind chunks
    // with explicit cells, place_device.hpp)
    if (postings_in_runs * 10u < postings * 9u) return false;
    if (explicit_image_bytes > (512ull << 20)) return true;
    return full_occupancy && explicit_image_bytes > kInfinityCacheBytes && run_coded_image_bytes <= kInfinityCacheBytes;
}
inline bool is_run(const epik_amd_pkdb_value *v, uint64_t len)
{
    if (len == 0 || len >= 65536u) return false;
    for (uint64_t i = 1; i < len; ++i)
        if (v[i].branch != v[i - 1].branch + 1u) return false;
    return true;
}
// bytes / lines of a list in that form, and the first word of its table entry
inline uint64_t run_lines(const epik_amd_pkdb_value *v, uint64_t len, bool runs)
{
    return ((runs && is_run(v, len) ? len * 4u : len * 6u) + 127u) / 128u;
}
inline uint32_t run_entry_word(const epik_amd_pkdb_value *v, uint64_t len, bool runs, uint32_t top)
{
    // len | first cell << 16; first cell 0 = explicit cells (cell 0 is the dummy row, never a posting's)
    return (uint32_t)len | ((runs && is_run(v, len)) ? (top - v[0].branch) << 16 : 0u);
}
void write_run_scores(uint8_t *dst, const epik_amd_pkdb_value *src, uint64_t len)
{
    for (uint64_t j = 0; j < len; ++j) std::memcpy(dst + 4u * j, &src[j].score, 4);
}
inline uint32_t pad4(uint32_t bytes) { return (bytes + 3u) & ~3u; }

struct TeamChoice {
    int waves = 0;
    uint32_t passes = 0, slice_rows = 0, rows_pad = 0, resident = 0;
};

// (W, P) of the team kernel for a tree of `n` branches.  Measured (r02, N = 10 000 .. 30 000, 150 bp
// reads): a read takes about (0.15 + P) x 20 us in a workgroup whatever the slice size, so the
// geometry that places most reads per second is the one with most resident workgroups per unit
// of that time -- e.g. two passes of half-size slices at N = 20 000 (3 workgroups per CU) beat one
// pass (1 workgroup) 1.8 x.  4 waves always beat 8 (fewer slices to merge, fewer idle waves in the
// front end): 8 only when forced; 2 -- half the items per read, each with twice the rows -- for trees of up to
// 4 200 branches (3 500 until round 5: with two-wave workgroups where they fit more waves, a merge that packs four reads
// to a wave and 8-byte table entries, two slices hold out longer: N = 3 999: 107.8 M reads/s against 102.5 with four).
// The 32-bit-count kernel must fit a CU.
constexpr bool kNarrowFilterByDefault = true;  // (protein k = 7, 998 M postings: 245.9 M reads/s against 237.9 M with 64-bit words)

TeamChoice choose_team(uint32_t n, uint32_t keep, int forced_waves, uint32_t forced_passes)
{
    TeamChoice best;
    uint32_t best_blocks = 0;
    // (measured, round 4, 150 bp reads, million-read batches, M reads/s with one wavefront per read / 2 / 4 slices:
    // N = 1 999: 125 / 130 / 107; 2 499: 103 / 127 / 103; 2 999: 93 / 114 / 103; 3 999: 71 / 92 / 100; 4 999: 58 / 88 / 97;
    // 9 999: 25 / 63 / 67 -- an item's fixed cost against the slice epilogue's sweeps and the waves LDS leaves room for)
    const int waves = forced_waves ? forced_waves : (n <= 4200u ? 2 : 4);
    // Passes (measured, round 4, `profiles/r04_sweep_tree_sizes_passes.txt` and DESIGN.md 3.2: M reads/s by passes --
    // N = 12 499: 42.6 / 44.7; 14 999: 41.8 / 43.4; 15 999: 23.1 / 42.3; 19 999: 19.2 / 41.9 / 26.1; 29 999: - / 29.2 / 24.9 /
    // 17.7; 39 999: - / - / 19.6 / 19.5 / 14.7; 49 999: - / - / - / 15.7 / 14.6 / 12.2): the fewest passes whose slices
    // leave room for two workgroups of the streaming kernel on a CU -- three where one pass would do: every pass
    // looks every k-mer up again and adds an item per slice, which costs more than the third workgroup brings.
    // (Round 2's rule -- most resident workgroups per (0.15 + passes) -- took 6 passes at N = 49 999: 12.2 M.)
    for (uint32_t passes = forced_passes ? forced_passes : 1; passes <= 4096; ++passes) {
        const uint32_t slices = (uint32_t)waves * passes;
        const uint32_t rows = (n + slices - 1) / slices;
        const uint32_t rows_pad = team_rows_pad(rows);
        const uint32_t desc = team_desc_bytes(keep);
        if (team_lds_bytes(waves, passes, team_slice_bytes(rows_pad, kCounts32), desc, keep) > kLdsPerCu) continue;
        // judged with the counts short reads get: 8 bits where that kernel exists (its "seen" bits of the
        // ambiguous path must fit the slice's descriptor list), else 16
        const int usual = (rows_pad + 7u) / 8u <= desc ? kCounts8 : kCounts16;
        const size_t normal = team_lds_bytes(waves, passes, team_slice_bytes(rows_pad, usual), desc, keep);
        const uint32_t blocks = team_resident_blocks(waves, normal);
        // (in waves of the streaming kernel that LDS leaves room for, with the better of its two workgroup sizes)
        const uint32_t slice_bytes = team_slice_bytes(rows_pad, usual);
        const uint32_t stream_waves = std::max(stream_blocks_by_lds(stream_lds_bytes(slice_bytes, desc, 2)) * 2u,
                                               stream_blocks_by_lds(stream_lds_bytes(slice_bytes, desc, 4)) * 4u);
        const TeamChoice choice{waves, passes, rows, rows_pad, blocks * (uint32_t)waves};
        if (best.waves == 0 || stream_waves > best_blocks) best = choice, best_blocks = stream_waves;  // (should none reach eight)
        if (forced_passes || stream_waves >= (passes == 1 ? 12u : 8u)) return choice;
    }
    return best;
}

}  // namespace

int resolve_shard(const epik_amd_placer_desc *d, uint32_t &shard_index, uint32_t &shard_count, std::string &err)
{
    if (!d || d->shard == 0) return EPIK_AMD_OK;
    const uint32_t g = d->shard & 0xffffu, count = d->shard >> 16;
    if (count < 2 || g >= count) {
        err = "desc.shard must be 0 or g | G << 16 with G >= 2 and g < G";
        return EPIK_AMD_ERR_INVALID;
    }
    if (shard_count > 1 && (shard_count != count || shard_index != g)) {
        err = "desc.shard says the descriptor holds another shard than the one asked for";
        return EPIK_AMD_ERR_INVALID;
    }
    shard_index = g, shard_count = count;
    return EPIK_AMD_OK;
}

int validate(const epik_amd_placer_desc *d, uint32_t shard_index, uint32_t shard_count, std::string &err)
{
    auto fail = [&](int code, const char *msg) {
        err = msg;
        return code;
    };
    if (!d) return fail(EPIK_AMD_ERR_INVALID, "null argument");
    if (shard_count == 0 || shard_index >= shard_count)
        return fail(EPIK_AMD_ERR_INVALID, "shard_index must be below shard_count");
    if (d->abi_version != EPIK_AMD_ABI_VERSION) return fail(EPIK_AMD_ERR_INVALID, "abi_version mismatch");
    if (d->kmer_size < 1 || d->kmer_size > 32) return fail(EPIK_AMD_ERR_UNSUPPORTED, "kmer_size must be in [1, 32]");
    if (d->alphabet_size < 2 || d->alphabet_size > 32)
        return fail(EPIK_AMD_ERR_INVALID, "alphabet_size must be in [2, 32]");
    if (d->num_branches == 0 || d->num_branches >= (1u << 24))
        return fail(EPIK_AMD_ERR_INVALID, "num_branches out of range");
    if (d->num_entries >= (1ull << 40)) return fail(EPIK_AMD_ERR_UNSUPPORTED, "more than 2^40 postings");
    if (d->keep_at_most == 0 || d->keep_at_most > 64)
        return fail(EPIK_AMD_ERR_UNSUPPORTED, "keep_at_most must be in [1, 64]");
    if (d->offset_bits != 32 && d->offset_bits != 64) return fail(EPIK_AMD_ERR_INVALID, "offset_bits must be 32 or 64");
    if (!d->offsets || !d->char_class || (!d->values && d->num_entries))
        return fail(EPIK_AMD_ERR_INVALID, "null database pointer");
    // dense key space: num_keys == sigma^k, and codes are 32-bit on the device (nucl k <= 15, amino k <= 7)
    {
        uint64_t nk = 1;
        for (uint32_t i = 0; i < d->kmer_size; ++i) {
            nk *= d->alphabet_size;
            if (nk > 0xffffffffull) return fail(EPIK_AMD_ERR_UNSUPPORTED, "alphabet_size^kmer_size exceeds 2^32 keys");
        }
        if (nk != d->num_keys) return fail(EPIK_AMD_ERR_INVALID, "num_keys != alphabet_size^kmer_size");
    }
    const Source src{d, 0, 1};
    if (d->offset_bits == 32 && d->num_entries > 0xffffffffull)
        return fail(EPIK_AMD_ERR_INVALID, "num_entries needs 64-bit offsets");
    if (src.sparse() && d->num_present > d->num_keys) return fail(EPIK_AMD_ERR_INVALID, "num_present > num_keys");
    if (!src.sparse() && d->num_present != 0) return fail(EPIK_AMD_ERR_INVALID, "num_present without keys");
    const uint64_t n_lists = src.offsets_len() - 1;  // dense: one (possibly empty) list per code; sparse: per present code
    if (src.offset_at(0) != 0 || src.offset_at(n_lists) != d->num_entries)
        return fail(EPIK_AMD_ERR_INVALID, "offsets[0] != 0 or the last offset != num_entries");
    // The lists themselves: monotone offsets, lists shorter than 2^24, every branch below num_branches,
    // finite scores, and -- what the kernels' lane-parallel read-add-write of a list relies on -- no
    // branch twice in one list (the reference's lists are built per branch, one score each: main.cpp:257).
    // Sparse form: the codes strictly ascending and inside the key space.
    std::vector<uint32_t> seen_in;  // seen_in[b] = 1 + the last list that held branch b
    try {
        seen_in.assign(d->num_branches, 0);
    } catch (const std::bad_alloc &) {
        return fail(EPIK_AMD_ERR_INVALID, "out of host memory");
    }
    uint32_t list_id = 0;
    for (uint64_t i = 0; i < n_lists; ++i) {
        if (src.sparse()) {
            if (d->keys[i] >= d->num_keys) return fail(EPIK_AMD_ERR_INVALID, "keys[] holds a code outside the key space");
            if (i && d->keys[i] <= d->keys[i - 1]) return fail(EPIK_AMD_ERR_INVALID, "keys[] not strictly ascending");
        }
        const uint64_t b = src.offset_at(i), e = src.offset_at(i + 1);
        if (e < b || e > d->num_entries) return fail(EPIK_AMD_ERR_INVALID, "offsets not monotone");
        if (e - b >= (1ull << 24)) return fail(EPIK_AMD_ERR_INVALID, "posting list of 2^24 entries or more");
        if (e == b) continue;
        if (d->shard != 0 && (src.sparse() ? d->keys[i] : i) % (d->shard >> 16) != (d->shard & 0xffffu))
            return fail(EPIK_AMD_ERR_INVALID, "desc.shard names a shard, and the descriptor holds a list of another shard's code");
        if (++list_id == 0) {  // the list counter wrapped (> 4 G non-empty lists): start a new epoch
            std::fill(seen_in.begin(), seen_in.end(), 0u);
            list_id = 1;
        }
        for (uint64_t j = b; j < e; ++j) {
            const uint32_t branch = d->values[j].branch;
            if (branch >= d->num_branches) return fail(EPIK_AMD_ERR_INVALID, "posting with branch >= num_branches");
            if (!std::isfinite(d->values[j].score))  // the kernels mark "no edge" with -inf
                return fail(EPIK_AMD_ERR_INVALID, "posting with a non-finite score");
            if (seen_in[branch] == list_id)
                return fail(EPIK_AMD_ERR_INVALID, "a posting list names the same branch twice");
            seen_in[branch] = list_id;
        }
    }
    return EPIK_AMD_OK;
}

namespace {

// Which kernel places a read, from the tree alone (+ EPIK_AMD_KERNEL): one wavefront per read while enough of them fit
// a CU, else the branch range in slices.  Fills plan.n_pad / wave_resident.
struct KernelDecision {
    bool team = false;
    int forced_waves = 0;
    uint32_t forced_passes = 0;
};
int decide_kernel(uint32_t num_branches, const char *forced_kernel, const char *forced_layout, Plan &plan, KernelDecision &k,
                  std::string &err)
{
    auto fail = [&](int code, const char *msg) {
        err = msg;
        return code;
    };
    plan.n_pad = (num_branches + 1u + 63u) & ~63u;
    const bool wave_fits = wave_lds_bytes(plan.n_pad, kCounts32) <= kLdsPerCu;
    for (int c = 0; c < 3; ++c) plan.wave_resident[c] = wave_fits ? wave_kernel_resident_waves(plan.n_pad, c) : 0;
    // (measured, round 4 -- choose_team: with two slices per pass the front + streaming + merge kernels win where fewer
    // than 12 waves of the other fit a CU with 16-bit counts, i.e. from 2 048 LDS rows per wave on, N >= 1 984)
    k.team = !wave_fits || plan.wave_resident[kCounts16] < 12;
    if (forced_kernel && forced_kernel[0]) {
        if (std::strcmp(forced_kernel, "wave") == 0) {
            if (!wave_fits) return fail(EPIK_AMD_ERR_UNSUPPORTED, "num_branches too large for the one-wavefront-per-read kernel");
            k.team = false;
        } else if (std::strcmp(forced_kernel, "team") == 0) {
            k.team = true;
        } else if (std::strncmp(forced_kernel, "team2", 5) == 0 || std::strncmp(forced_kernel, "team4", 5) == 0 ||
                   std::strncmp(forced_kernel, "team8", 5) == 0) {
            k.team = true;  // teamW or teamWxP: W waves, (tests) at least P passes
            k.forced_waves = forced_kernel[4] - '0';
            if (forced_kernel[5] == 'x') k.forced_passes = (uint32_t)std::strtoul(forced_kernel + 6, nullptr, 10);
            if (forced_kernel[5] != 0 && (forced_kernel[5] != 'x' || k.forced_passes == 0 || k.forced_passes > 64))
                return fail(EPIK_AMD_ERR_INVALID, "EPIK_AMD_KERNEL must be wave, team, team2[xP], team4[xP] or team8[xP]");
        } else {
            return fail(EPIK_AMD_ERR_INVALID, "EPIK_AMD_KERNEL must be wave, team, team2[xP], team4[xP] or team8[xP]");
        }
    }
    if (forced_layout && forced_layout[0] && std::strcmp(forced_layout, "compact") != 0 &&
        std::strcmp(forced_layout, "packed") != 0 && std::strcmp(forced_layout, "paired") != 0 &&
        std::strcmp(forced_layout, "filtered") != 0)
        return fail(EPIK_AMD_ERR_INVALID, "EPIK_AMD_LAYOUT must be compact, packed, paired or filtered");
    return EPIK_AMD_OK;
}

// The sliced layout's geometry and its table (the posting region depends on the lists: the callers size it).
int plan_team_geometry(uint32_t num_branches, uint32_t keep_at_most, uint32_t alphabet_size, uint32_t kmer_size, uint64_t num_keys,
                       const KernelDecision &k, Plan &plan, std::string &err)
{
    const TeamChoice c = choose_team(num_branches, keep_at_most, k.forced_waves, k.forced_passes);
    if (c.waves == 0) {
        err = "no team geometry fits this tree";
        return EPIK_AMD_ERR_UNSUPPORTED;
    }
    plan.layout = DbLayout::kTeam;
    plan.team_waves = c.waves;
    plan.team_passes = c.passes;
    plan.team_slice_rows = c.slice_rows;
    plan.team_rows_pad = c.rows_pad;
    // A table entry is 16 bytes, a lookup fetches a 128-byte line: with 4 letters the table is keyed, as the
    // paired layout of the one-wavefront kernels, by the (k-1)-mer X two consecutive k-mers of a read share
    // -- block X = the entries of a.X (slots 0-3) and X.b (slots 4-7), one line -- so that two lookups
    // cost one fetch (the front kernel is bound by exactly these fetches).  Every entry is stored twice.
    // (a shard's table is not paired: of two consecutive k-mers of a read at most one is the shard's as a rule,
    // and the kernels do not fetch for the other)
    plan.team_paired = alphabet_size == 4 && team_entry_bytes(c.waves) <= 16 && kmer_size >= 2 && plan.shard_count == 1 &&
                       !(std::getenv("EPIK_AMD_TEAM_TABLE") && std::strcmp(std::getenv("EPIK_AMD_TEAM_TABLE"), "plain") == 0);
    plan.table_keys = shard_keys(num_keys, plan.shard_index, plan.shard_count);
    plan.table_bytes = (uint64_t)c.passes * plan.table_keys * (uint64_t)team_entry_bytes(c.waves) * (plan.team_paired ? 2u : 1u);
    return EPIK_AMD_OK;
}

// The layout of the one-wavefront kernels (place_kernel.hip), from the key space, how many codes have a list and the
// device's free memory (+ EPIK_AMD_LAYOUT):
//  packed  : an 8-byte {len, line} entry per k-mer code + every list on 128-byte lines of its own;
//  paired  : the same lists behind a table keyed by the (k-1)-mer that two consecutive k-mers of a read
//            share: one table line per two lookups, 16 bytes per code; 4-letter alphabets -- their default;
//  filtered: packed behind a presence filter keyed like the paired table: the other alphabets when at
//            most a quarter of the codes have a list;
//  compact : the CSR (4- or 8-byte offsets), 8-byte postings back to back: when the table would take
//            more than a quarter of the device's free memory.
struct WaveLayout {
    bool paired = false, filtered = false, packed = false;
};
WaveLayout choose_wave_layout(uint32_t alphabet_size, uint64_t num_keys, uint64_t present_codes, size_t free_mem,
                              const char *forced_layout)
{
    const bool sparse = present_codes * 4u <= num_keys;
    const bool can_pair = alphabet_size == 4, can_filter = alphabet_size <= 32;
    const bool table_fits = num_keys * 16u <= free_mem / 4;
    WaveLayout w;
    if (forced_layout && forced_layout[0]) {
        w.filtered = can_filter && (std::strcmp(forced_layout, "filtered") == 0 ||
                                    (!can_pair && std::strcmp(forced_layout, "paired") == 0));
        w.paired = can_pair && std::strcmp(forced_layout, "paired") == 0;
        w.packed = w.paired || w.filtered || std::strcmp(forced_layout, "compact") != 0;
    } else if (table_fits) {
        w.paired = can_pair;
        w.filtered = !w.paired && can_filter && sparse;
        w.packed = true;
    }
    return w;
}
// table and presence filter of the packed layouts
void size_packed_tables(uint32_t alphabet_size, uint64_t num_keys, const WaveLayout &w, Plan &plan)
{
    plan.table_bytes = num_keys * (w.paired ? 16u : 8u) + 8u;
    // A presence record holds 2 x sigma bits.  In 64-bit words the filter of a protein database with k = 7 takes 512 MB,
    // twice the Infinity Cache; its 40 bits packed into 5 bytes (read as two dwords at any byte) 320 MB, and a record
    // straddles a 128-byte line once in 32.  EPIK_AMD_FILTER=wide|narrow forces one.
    plan.filter_rec_bytes = 8;
    if (w.filtered && 2u * alphabet_size <= 40u) {
        const char *forced_filter = std::getenv("EPIK_AMD_FILTER");
        const bool narrow = forced_filter ? std::strcmp(forced_filter, "narrow") == 0 : kNarrowFilterByDefault;
        if (narrow) plan.filter_rec_bytes = 5;
    }
    // (+ 8: a record's second dword may lie behind the last record)
    plan.filter_bytes = w.filtered ? (num_keys / alphabet_size) * plan.filter_rec_bytes + (plan.filter_rec_bytes == 8 ? 0u : 8u) : 0;
}

}  // namespace

int make_plan(const Source &src, size_t free_mem, const char *forced_layout, const char *forced_kernel, Plan &plan,
              std::string &err)
{
    const epik_amd_placer_desc *d = src.d;
    auto fail = [&](int code, const char *msg) {
        err = msg;
        return code;
    };
    plan = Plan{};
    plan.shard_index = src.shard_index, plan.shard_count = src.shard_count;
    KernelDecision kernel;
    if (const int rc = decide_kernel(d->num_branches, forced_kernel, forced_layout, plan, kernel, err); rc != EPIK_AMD_OK) return rc;
    const bool team = kernel.team;

    {
        Cursor all(src);
        for (uint64_t key = 0; key < d->num_keys; ++key) {
            const uint64_t len = all.list(key);
            plan.kept_entries += len;
            plan.present_codes += len != 0;
        }
    }

    if (team) {
        if (const int rc = plan_team_geometry(d->num_branches, d->keep_at_most, d->alphabet_size, d->kmer_size, d->num_keys, kernel, plan, err);
            rc != EPIK_AMD_OK)
            return rc;
        const TeamChoice c{plan.team_waves, plan.team_passes, plan.team_slice_rows, plan.team_rows_pad, 0};
        // size of the sliced posting region: every (code, pass) on 128-byte lines of its own
        const uint32_t slices = (uint32_t)c.waves * c.passes;
        std::vector<uint32_t> cnt(slices);
        uint64_t lines = 0;
        if (plan.team_paired) plan.team_quarter_lines.assign((size_t)c.passes * 4, 0);
        std::vector<uint64_t> pass_lines(c.passes, 0);  // lines of each pass so far (a pass's lines are numbered from the region's start: see build)
        const uint64_t quarter = d->num_keys / 4;
        Cursor walk(src);
        for (uint64_t key = 0; key < d->num_keys; ++key) {
            if (plan.team_paired && quarter && key % quarter == 0 && key / quarter < 4)
                for (uint32_t p = 0; p < c.passes; ++p) plan.team_quarter_lines[(size_t)p * 4 + key / quarter] = pass_lines[p];
            uint64_t first = 0;
            const uint64_t len = walk.list(key, &first);
            if (len == 0) continue;
            std::fill(cnt.begin(), cnt.end(), 0u);
            const epik_amd_pkdb_value *v = d->values + first;
            for (uint64_t i = 0; i < len; ++i) ++cnt[v[i].branch / c.slice_rows];
            for (uint32_t p = 0; p < c.passes; ++p) {
                uint32_t bytes = 0;
                for (int w = 0; w < c.waves; ++w) {
                    bytes += pad4(cnt[p * c.waves + w] * 6u);
                    plan.team_chunks += (cnt[p * c.waves + w] + 63u) / 64u;
                }
                lines += (bytes + 127u) / 128u;
                pass_lines[p] += (bytes + 127u) / 128u;
            }
        }
        if (lines >= (1ull << 32)) return fail(EPIK_AMD_ERR_UNSUPPORTED, "posting region of 512 GiB or more");
        plan.posting_bytes = lines * 128u + 512u;  // +512: room behind the last list (descriptors are exact)
        return EPIK_AMD_OK;
    }

    // ---- layouts of the one-wavefront-per-read kernels (place_kernel.hip): choose_wave_layout -----------------
    const WaveLayout w = choose_wave_layout(d->alphabet_size, d->num_keys, plan.present_codes, free_mem, forced_layout);
    const bool paired = w.paired, filtered = w.filtered, packed = w.packed;
    if (!packed) {
        plan.layout = d->offset_bits == 64 ? DbLayout::kCompact64 : DbLayout::kCompact32;
        plan.table_bytes = (d->num_keys + 1) * (d->offset_bits / 8u);  // (the device table is dense whatever the form handed over)
        plan.posting_bytes = plan.kept_entries * 8u + 512u;
        return EPIK_AMD_OK;
    }
    plan.layout = paired ? DbLayout::kPaired : filtered ? DbLayout::kFiltered : DbLayout::kPacked;
    uint64_t lines = 0;
    const uint64_t quarter = d->num_keys / 4;
    if (plan.n_pad > 65536u) return fail(EPIK_AMD_ERR_UNSUPPORTED, "num_branches too large for the packed layouts");
    const uint64_t table_bytes = d->num_keys * (paired ? 16u : 8u) + 8u;
    {   // how large the image is with every list explicit, and run-coded, decides whether the lists are run-coded
        uint64_t explicit_lines = 0, coded_lines = 0, in_runs = 0;
        Cursor walk(src);
        for (uint64_t key = 0; key < d->num_keys; ++key) {
            uint64_t first = 0;
            const uint64_t len = walk.list(key, &first);
            explicit_lines += (len * 6u + 127u) / 128u;
            const bool run = is_run(d->values + first, len);
            coded_lines += ((run ? len * 4u : len * 6u) + 127u) / 128u;
            if (run) in_runs += len;
        }
        plan.runs = choose_runs(explicit_lines * 128u + table_bytes, coded_lines * 128u + table_bytes, plan.kept_entries, in_runs,
                                plan.wave_resident[kCounts8] >= kWaveKernelWavesPerCu);
    }
    const bool runs = plan.runs;
    Cursor walk(src);
    for (uint64_t key = 0; key < d->num_keys; ++key) {
        if (paired && quarter && key % quarter == 0 && key / quarter < 4) plan.quarter_lines[key / quarter] = lines;
        uint64_t first = 0;
        const uint64_t len = walk.list(key, &first);
        if (len >= 65536u) return fail(EPIK_AMD_ERR_INVALID, "a posting list of 65536 entries or more in a tree of fewer branches");
        lines += run_lines(d->values + first, len, runs);
    }
    if (lines >= (1ull << 32)) return fail(EPIK_AMD_ERR_UNSUPPORTED, "posting region of 512 GiB or more");
    plan.posting_bytes = lines * 128u + 512u;
    size_packed_tables(d->alphabet_size, d->num_keys, w, plan);
    return EPIK_AMD_OK;
}

int plan_sizes(const SizeDesc &z, size_t free_mem, const char *forced_layout, const char *forced_kernel, Plan &plan,
               bool &posting_bytes_is_bound, std::string &err)
{
    auto fail = [&](int code, const char *msg) {
        err = msg;
        return code;
    };
    plan = Plan{};
    posting_bytes_is_bound = false;
    if (z.shard_count == 0 || z.shard_index >= z.shard_count) return fail(EPIK_AMD_ERR_INVALID, "shard_index must be below shard_count");
    if (z.kmer_size < 1 || z.kmer_size > 32) return fail(EPIK_AMD_ERR_UNSUPPORTED, "kmer_size must be in [1, 32]");
    if (z.alphabet_size < 2 || z.alphabet_size > 32) return fail(EPIK_AMD_ERR_INVALID, "alphabet_size must be in [2, 32]");
    if (z.num_branches == 0 || z.num_branches >= (1u << 24)) return fail(EPIK_AMD_ERR_INVALID, "num_branches out of range");
    if (z.keep_at_most == 0 || z.keep_at_most > 64) return fail(EPIK_AMD_ERR_UNSUPPORTED, "keep_at_most must be in [1, 64]");
    if (z.n_bins && !z.bins) return fail(EPIK_AMD_ERR_INVALID, "null argument");
    uint64_t num_keys = 1;
    for (uint32_t i = 0; i < z.kmer_size; ++i) {
        num_keys *= z.alphabet_size;
        if (num_keys > 0xffffffffull) return fail(EPIK_AMD_ERR_UNSUPPORTED, "alphabet_size^kmer_size exceeds 2^32 keys");
    }
    plan.shard_index = z.shard_index, plan.shard_count = z.shard_count;
    uint64_t in_runs = 0;
    for (uint64_t i = 0; i < z.n_bins; ++i) {
        const epik_amd_list_bin &b = z.bins[i];
        if (b.lists_in_runs > b.lists) return fail(EPIK_AMD_ERR_INVALID, "a bin with more lists in runs than lists");
        if (b.length == 0 || b.lists == 0) continue;
        if (b.length > z.num_branches) return fail(EPIK_AMD_ERR_INVALID, "a list longer than the tree has branches (branches are distinct inside a list)");
        plan.kept_entries += b.length * b.lists;
        plan.present_codes += b.lists;
        if (b.length < 65536u) in_runs += b.length * b.lists_in_runs;
    }
    if (plan.present_codes > shard_keys(num_keys, z.shard_index, z.shard_count))
        return fail(EPIK_AMD_ERR_INVALID, "more lists than the shard has codes");
    if (plan.kept_entries >= (1ull << 40)) return fail(EPIK_AMD_ERR_UNSUPPORTED, "more than 2^40 postings");
    KernelDecision kernel;
    if (const int rc = decide_kernel(z.num_branches, forced_kernel, forced_layout, plan, kernel, err); rc != EPIK_AMD_OK) return rc;
    if (kernel.team) {
        if (const int rc = plan_team_geometry(z.num_branches, z.keep_at_most, z.alphabet_size, z.kmer_size, num_keys, kernel, plan, err);
            rc != EPIK_AMD_OK)
            return rc;
        // The posting region: every (code, pass) on 128-byte lines of its own, the list cut into its sublists, each
        // padded to 4 bytes.  How a list falls over the slices is not in a histogram of lengths; what is certain: its
        // sublists take at most 6 len + 3 min(len, slices) bytes, and each pass it reaches beyond the first at most one
        // more line -- an upper bound, a few percent above what the lists will take.
        const uint64_t slices = (uint64_t)plan.team_waves * plan.team_passes;
        uint64_t lines = 0;
        for (uint64_t i = 0; i < z.n_bins; ++i) {
            const epik_amd_list_bin &b = z.bins[i];
            if (b.length == 0 || b.lists == 0) continue;
            const uint64_t bytes = 6u * b.length + 3u * std::min<uint64_t>(b.length, slices);
            lines += b.lists * ((bytes + 127u) / 128u + (std::min<uint64_t>(b.length, plan.team_passes) - 1u));
            plan.team_chunks += b.lists * ((b.length + 63u) / 64u + std::min<uint64_t>(b.length, slices) - 1u);
        }
        if (lines >= (1ull << 32)) return fail(EPIK_AMD_ERR_UNSUPPORTED, "posting region of 512 GiB or more");
        plan.posting_bytes = lines * 128u + 512u;
        posting_bytes_is_bound = true;
        return EPIK_AMD_OK;
    }
    const WaveLayout w = choose_wave_layout(z.alphabet_size, num_keys, plan.present_codes, free_mem, forced_layout);
    if (!w.packed) {
        const uint32_t offset_bits = plan.kept_entries > 0xffffffffull ? 64u : 32u;  // (what a caller's offsets[] would need)
        plan.layout = offset_bits == 64 ? DbLayout::kCompact64 : DbLayout::kCompact32;
        plan.table_bytes = (num_keys + 1) * (offset_bits / 8u);
        plan.posting_bytes = plan.kept_entries * 8u + 512u;
        return EPIK_AMD_OK;
    }
    plan.layout = w.paired ? DbLayout::kPaired : w.filtered ? DbLayout::kFiltered : DbLayout::kPacked;
    if (plan.n_pad > 65536u) return fail(EPIK_AMD_ERR_UNSUPPORTED, "num_branches too large for the packed layouts");
    uint64_t explicit_lines = 0, coded_lines = 0;
    for (uint64_t i = 0; i < z.n_bins; ++i) {
        const epik_amd_list_bin &b = z.bins[i];
        if (b.length == 0 || b.lists == 0) continue;
        explicit_lines += b.lists * ((b.length * 6u + 127u) / 128u);
        const uint64_t as_runs = b.length < 65536u ? b.lists_in_runs : 0;
        coded_lines += (b.lists - as_runs) * ((b.length * 6u + 127u) / 128u) + as_runs * ((b.length * 4u + 127u) / 128u);
    }
    const uint64_t packed_table = num_keys * (w.paired ? 16u : 8u) + 8u;
    plan.runs = choose_runs(explicit_lines * 128u + packed_table, coded_lines * 128u + packed_table, plan.kept_entries, in_runs,
                            plan.wave_resident[kCounts8] >= kWaveKernelWavesPerCu);
    uint64_t lines = 0;
    for (uint64_t i = 0; i < z.n_bins; ++i) {
        const epik_amd_list_bin &b = z.bins[i];
        if (b.length == 0 || b.lists == 0) continue;
        if (b.length >= 65536u) return fail(EPIK_AMD_ERR_INVALID, "a posting list of 65536 entries or more in a tree of fewer branches");
        const uint64_t as_runs = plan.runs ? b.lists_in_runs : 0;
        lines += (b.lists - as_runs) * ((b.length * 6u + 127u) / 128u) + as_runs * ((b.length * 4u + 127u) / 128u);
    }
    if (lines >= (1ull << 32)) return fail(EPIK_AMD_ERR_UNSUPPORTED, "posting region of 512 GiB or more");
    plan.posting_bytes = lines * 128u + 512u;
    size_packed_tables(z.alphabet_size, num_keys, w, plan);
    return EPIK_AMD_OK;
}

int build(const Source &src, const Plan &plan, Sink &table, Sink *filter, Sink &postings, std::string &err)
{
    const epik_amd_placer_desc *d = src.d;
    const uint64_t num_keys = d->num_keys;
    try {
        if (plan.layout == DbLayout::kTeam) {
            const int W = plan.team_waves;
            const uint32_t rows = plan.team_slice_rows, top = plan.team_rows_pad - 1u;
            const size_t entry_bytes = (size_t)team_entry_bytes(W);
            std::vector<std::vector<epik_amd_pkdb_value>> sub((size_t)W);
            // sublist lengths of a code in a pass, and the 128-byte lines its sublists take
            auto slice_counts = [&](Cursor &at, uint64_t key, uint32_t first_slice, uint32_t *cnt) {
                for (int w = 0; w < W; ++w) cnt[w] = 0;
                uint64_t first = 0;
                const uint64_t len = at.list(key, &first);
                const epik_amd_pkdb_value *v = d->values + first;
                for (uint64_t i = 0; i < len; ++i) {
                    const uint32_t slice = v[i].branch / rows;
                    if (slice >= first_slice && slice < first_slice + (uint32_t)W) ++cnt[slice - first_slice];
                }
                uint32_t bytes = 0;
                for (int w = 0; w < W; ++w) bytes += pad4(cnt[w] * 6u);
                return (uint64_t)((bytes + 127u) / 128u);
            };
            auto put_entry = [&](uint8_t *entry, uint64_t line, const uint32_t *cnt, uint64_t n_lines) {
                if (n_lines == 0) return;  // zero-filled: an absent code has len[] = 0
                put_u32(entry, (uint32_t)line);
                for (int w = 0; w < W; ++w) {
                    const uint16_t n = (uint16_t)cnt[w];
                    std::memcpy(entry + 4 + 2 * w, &n, 2);
                }
            };
            uint64_t line = 0;
            std::vector<uint32_t> cnt((size_t)W);
            for (uint32_t pass = 0; pass < plan.team_passes; ++pass) {
                const uint32_t first_slice = pass * (uint32_t)W;
                const uint64_t pass_first_line = line;
                // ---- the postings of the pass, in code order (and, unpaired, the table with them) ----------
                RecordWriter entries(table, entry_bytes, plan.team_paired ? 0 : plan.table_keys);
                Cursor walk(src);
                for (uint64_t key = 0; key < num_keys; ++key) {
                    // (a shard's table: an entry per code of the shard, code / shard_count being its place)
                    uint8_t *entry = plan.team_paired || !src.kept(key) ? nullptr : entries.next();
                    uint64_t first = 0;
                    const uint64_t len = walk.list(key, &first);
                    if (len == 0) continue;
                    for (auto &s : sub) s.clear();
                    const epik_amd_pkdb_value *v = d->values + first;
                    for (uint64_t i = 0; i < len; ++i) {  // stable: a sublist keeps the list's order
                        const uint32_t slice = v[i].branch / rows;
                        if (slice >= first_slice && slice < first_slice + (uint32_t)W)
                            sub[slice - first_slice].push_back({v[i].branch - slice * rows, v[i].score});
                    }
                    uint32_t bytes = 0;
                    for (int w = 0; w < W; ++w) {
                        cnt[w] = (uint32_t)sub[w].size();
                        bytes += pad4(cnt[w] * 6u);
                    }
                    if (bytes == 0) continue;
                    const uint64_t n_lines = (bytes + 127u) / 128u;
                    uint8_t *dst = postings.reserve((size_t)n_lines * 128u);
                    if (entry) put_entry(entry, line, cnt.data(), n_lines);
                    for (int w = 0; w < W; ++w) {
                        write_chunks(dst, sub[w].data(), sub[w].size(), top);
                        dst += pad4((uint32_t)sub[w].size() * 6u);
                    }
                    line += n_lines;
                }
                if (!plan.team_paired) continue;
                // ---- the paired table of the pass: block X = entries of a.X (slots 0-3) and X.b (slots 4-7).
                // The line of a code is the number of lines of all codes of the pass in front of it: five
                // cursors walk the key space in step -- one per quarter (a.X, a fixed, X rising) and one over
                // all codes (X.b) -- so no per-code array is needed; the lists are counted again for it.
                const uint64_t blocks = num_keys / 4;  // 4^(k-1)
                RecordWriter out(table, 8 * entry_bytes, blocks);
                uint64_t quarter_line[4], seq_line = pass_first_line;
                for (int a = 0; a < 4; ++a) quarter_line[a] = pass_first_line + plan.team_quarter_lines[(size_t)pass * 4 + a];
                Cursor quarter_at[4] = {Cursor(src), Cursor(src), Cursor(src), Cursor(src)}, seq_at(src);
                for (uint64_t x = 0; x < blocks; ++x) {
                    uint8_t *block = out.next();
                    for (uint32_t a = 0; a < 4; ++a) {
                        const uint64_t n_lines = slice_counts(quarter_at[a], a * blocks + x, first_slice, cnt.data());
                        put_entry(block + entry_bytes * a, quarter_line[a], cnt.data(), n_lines);
                        quarter_line[a] += n_lines;
                    }
                    for (uint32_t bb = 0; bb < 4; ++bb) {
                        const uint64_t n_lines = slice_counts(seq_at, x * 4 + bb, first_slice, cnt.data());
                        put_entry(block + entry_bytes * (4 + bb), seq_line, cnt.data(), n_lines);
                        seq_line += n_lines;
                    }
                }
            }
            postings.reserve(512);
            return EPIK_AMD_OK;
        }
        if (plan.layout == DbLayout::kCompact32 || plan.layout == DbLayout::kCompact64) {
            // the kept lists back to back as {f32 score, u32 cell}; offsets over the kept lists
            const size_t off_bytes = d->offset_bits / 8u;
            const uint32_t top = plan.n_pad - 1u;
            RecordWriter offsets(table, off_bytes, num_keys + 1);
            uint64_t at = 0;
            Cursor walk(src);
            for (uint64_t key = 0; key <= num_keys; ++key) {
                uint8_t *o = offsets.next();
                if (off_bytes == 8) {
                    std::memcpy(o, &at, 8);
                } else {
                    const uint32_t at32 = (uint32_t)at;
                    std::memcpy(o, &at32, 4);
                }
                if (key == num_keys) break;
                uint64_t first = 0;
                const uint64_t len = walk.list(key, &first);
                if (len == 0) continue;
                const epik_amd_pkdb_value *v = d->values + first;
                for (uint64_t c0 = 0; c0 < len; c0 += 4096) {
                    const uint64_t n = std::min<uint64_t>(4096, len - c0);
                    uint8_t *dst = postings.reserve((size_t)n * 8u);
                    for (uint64_t j = 0; j < n; ++j) {
                        const uint32_t cell = top - v[c0 + j].branch;
                        std::memcpy(dst + 8u * j, &v[c0 + j].score, 4);
                        std::memcpy(dst + 8u * j + 4u, &cell, 4);
                    }
                }
                at += len;
            }
            postings.reserve(512);
            return EPIK_AMD_OK;
        }
        // ---- packed lists; plain, paired or filtered table -----------------------------------------------
        const uint32_t top = plan.n_pad - 1u;
        const bool runs = plan.runs;
        {
            Cursor walk(src);
            for (uint64_t key = 0; key < num_keys; ++key) {
                uint64_t first = 0;
                const uint64_t len = walk.list(key, &first);
                if (len == 0) continue;
                const epik_amd_pkdb_value *v = d->values + first;
                uint8_t *dst = postings.reserve((size_t)run_lines(v, len, runs) * 128u);
                if (runs && is_run(v, len))
                    write_run_scores(dst, v, len);
                else
                    write_chunks(dst, v, len, top);
            }
        }
        postings.reserve(512);
        if (plan.layout == DbLayout::kPaired) {
            // Block X (a (k-1)-mer) holds the entries of the four codes a.X (slots 0-3) and the four X.b
            // (slots 4-7).  The line of a code is the number of lines of all codes in front of it: five
            // cursors walk the key space in step -- one per quarter (a.X, a fixed, X rising) and one over
            // all codes (X.b) -- so no per-code array is needed.
            const uint64_t blocks = num_keys / 4;  // 4^(k-1)
            RecordWriter out(table, 64, blocks);
            uint64_t quarter_line[4] = {plan.quarter_lines[0], plan.quarter_lines[1], plan.quarter_lines[2],
                                        plan.quarter_lines[3]};
            uint64_t seq_line = 0;
            Cursor quarter_at[4] = {Cursor(src), Cursor(src), Cursor(src), Cursor(src)}, seq_at(src);
            for (uint64_t x = 0; x < blocks; ++x) {
                uint8_t *block = out.next();
                for (uint32_t a = 0; a < 4; ++a) {
                    uint64_t first = 0;
                    const uint64_t len = quarter_at[a].list(a * blocks + x, &first);
                    put_u32(block + 8 * a, run_entry_word(d->values + first, len, runs, top));
                    put_u32(block + 8 * a + 4, (uint32_t)quarter_line[a]);
                    quarter_line[a] += run_lines(d->values + first, len, runs);
                }
                for (uint32_t b = 0; b < 4; ++b) {
                    uint64_t first = 0;
                    const uint64_t len = seq_at.list(x * 4 + b, &first);
                    put_u32(block + 32 + 8 * b, run_entry_word(d->values + first, len, runs, top));
                    put_u32(block + 32 + 8 * b + 4, (uint32_t)seq_line);
                    seq_line += run_lines(d->values + first, len, runs);
                }
            }
            table.reserve(8);
        } else {
            RecordWriter out(table, 8, num_keys);
            uint64_t line = 0;
            Cursor walk(src);
            for (uint64_t key = 0; key < num_keys; ++key) {
                uint8_t *e = out.next();
                uint64_t first = 0;
                const uint64_t len = walk.list(key, &first);
                put_u32(e, run_entry_word(d->values + first, len, runs, top));
                put_u32(e + 4, (uint32_t)line);
                line += run_lines(d->values + first, len, runs);
            }
            table.reserve(8);
        }
        if (plan.layout == DbLayout::kFiltered) {
            if (!filter) {
                err = "no sink for the presence filter";
                return EPIK_AMD_ERR_INVALID;
            }
            // filter[X], X a (k-1)-mer: bit a <=> code a.X has a list, bit sigma + b <=> code X.b has one
            const uint64_t sigma = d->alphabet_size, blocks = num_keys / sigma;  // sigma^(k-1)
            const uint32_t rec = plan.filter_rec_bytes;
            RecordWriter out(*filter, rec, blocks);
            std::vector<Cursor> first_at(sigma, Cursor(src));  // a.X, a fixed, X rising: one walk per first letter
            Cursor seq_at(src);                                 // X.b: the key space front to back
            for (uint64_t x = 0; x < blocks; ++x) {
                uint64_t word = 0;
                for (uint64_t a = 0; a < sigma; ++a)
                    if (first_at[a].list(a * blocks + x) != 0) word |= 1ull << a;
                for (uint64_t b = 0; b < sigma; ++b)
                    if (seq_at.list(x * sigma + b) != 0) word |= 1ull << (sigma + b);
                std::memcpy(out.next(), &word, rec);  // (little endian: the low 2 sigma bits)
            }
            if (rec != 8) {  // the padding behind the last record
                RecordWriter pad(*filter, 8, 1);
                const uint64_t zero = 0;
                std::memcpy(pad.next(), &zero, 8);
            }
        }
        return EPIK_AMD_OK;
    } catch (const std::bad_alloc &) {
        err = "out of host memory building the device database";
        return EPIK_AMD_ERR_INVALID;
    }
}

}  // namespace epik_amd::image
