// capi.hip -- the C-ABI layer of libepik_amd.so (declared in include/epik_amd.h).
//
// Host side of the drop-in boundary: replaces the constructor and the OpenMP loop
// of epik::placer (reference epik/src/epik/place.cpp:83-126, :201-275).  HIP
// runtime only -- no torch, no C++ types in any signature, no exceptions out.
// There is deliberately no CPU path: without a HIP device every compute entry
// point fails with EPIK_AMD_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "epik_amd.h"
#include "place_kernel.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

int fail_hip(hipError_t e, const char *what)
{
    return fail(EPIK_AMD_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(expr)                                        \
    do {                                                     \
        const hipError_t e_ = (expr);                        \
        if (e_ != hipSuccess) return fail_hip(e_, #expr);    \
    } while (0)

constexpr uint32_t kMaxLdsPerBlock = 160u * 1024u;  // gfx950: 160 KiB per CU

}  // namespace

struct epik_amd_placer {
    int device = 0;
    bool offsets64 = false;
    epik_amd::DbLayout layout = epik_amd::DbLayout::kCompact32;
    int counts = epik_amd::kCounts16;  // width of the per-branch counts the next device launch uses
    bool counts_forced = false;        // set by the caller / the environment: place() does not choose
    bool timing = false;
    void *d_table = nullptr;       // offsets (compact) or {len, line} entries (packed)
    uint64_t *d_filter = nullptr;  // presence words of the filtered layout
    uint8_t *d_postings = nullptr; // 6-byte postings
    uint64_t db_bytes = 0;
    uint32_t *d_char_class = nullptr;
    epik_amd::PlaceParams params{};  // batch fields are filled per call
    uint64_t num_keys = 0;
    uint64_t num_entries = 0;
    // launch geometry per count width (epik_amd::CountBits)
    struct geometry {
        uint32_t waves_per_block = 4;
        uint32_t lds_wave_bytes = 0;
        uint32_t lds_block_bytes = 0;
        uint32_t max_blocks = 0;
        uint32_t resident_waves = 0;  // per CU
    } geo[3];
    uint32_t last_blocks = 0;
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

extern "C" {

int epik_amd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *epik_amd_last_error(void) { return g_last_error.c_str(); }

void epik_amd_placer_destroy(epik_amd_placer *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
#ifdef EPIK_AMD_ABLATION
    if (p->params.dbg) {  // diagnostic build: where the waves spent their cycles, by phase
        unsigned long long t[8] = {0};
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(t, p->params.dbg, sizeof t, hipMemcpyDeviceToHost);
        double sum = 0;
        for (double x : t) sum += x;
        const char *names[8] = {"front-issue", "lookup-wait+scan", "expand+stream", "correction", "rows-out+reset",
                                "tau-rounds", "candidate-sweep", "score_sum+rank"};
        std::fprintf(stderr, "phase shares of wave cycles:");
        for (int i = 0; i < 8; ++i) std::fprintf(stderr, "  %s %.1f%%", names[i], 100 * t[i] / sum);
        std::fprintf(stderr, "\n");
        (void)hipFree(p->params.dbg);
    }
#endif
    (void)hipFree(p->d_table);
    (void)hipFree(p->d_filter);
    (void)hipFree(p->d_postings);
    (void)hipFree(p->d_char_class);
    (void)hipFree(p->d_seqs);
    (void)hipFree(p->d_seq_offsets);
    (void)hipFree(p->d_rows);
    (void)hipFree(p->d_n_rows);
    (void)hipFree(p->d_counts);
    (void)hipFree(p->d_total);
    if (p->ev_start) (void)hipEventDestroy(p->ev_start);
    if (p->ev_stop) (void)hipEventDestroy(p->ev_stop);
    for (hipEvent_t e : p->ev_in) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->ev_kernel) (void)hipEventDestroy(e);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    if (p->stream_in) (void)hipStreamDestroy(p->stream_in);
    if (p->stream_out) (void)hipStreamDestroy(p->stream_out);
    delete p;
}

int epik_amd_placer_create(const epik_amd_placer_desc *d, epik_amd_placer **out)
{
    return epik_amd_placer_create_sharded(d, 0, 1, out);
}

int epik_amd_placer_create_sharded(const epik_amd_placer_desc *d, uint32_t shard_index, uint32_t shard_count,
                                   epik_amd_placer **out)
{
    if (!d || !out) return fail(EPIK_AMD_ERR_INVALID, "null argument");
    *out = nullptr;
    if (shard_count == 0 || shard_index >= shard_count)
        return fail(EPIK_AMD_ERR_INVALID, "shard_index must be below shard_count");
    if (d->abi_version != EPIK_AMD_ABI_VERSION)
        return fail(EPIK_AMD_ERR_INVALID, "abi_version mismatch");
    if (d->kmer_size < 1 || d->kmer_size > 32)
        return fail(EPIK_AMD_ERR_UNSUPPORTED, "kmer_size must be in [1, 32]");
    if (d->alphabet_size < 2 || d->alphabet_size > 32)
        return fail(EPIK_AMD_ERR_INVALID, "alphabet_size must be in [2, 32]");
    if (d->num_branches == 0 || d->num_branches >= (1u << 24))
        return fail(EPIK_AMD_ERR_INVALID, "num_branches out of range");
    if (d->num_entries >= (1ull << 40))
        return fail(EPIK_AMD_ERR_UNSUPPORTED, "more than 2^40 postings");
    if (d->keep_at_most == 0 || d->keep_at_most > 64)
        return fail(EPIK_AMD_ERR_UNSUPPORTED, "keep_at_most must be in [1, 64]");
    if (d->offset_bits != 32 && d->offset_bits != 64)
        return fail(EPIK_AMD_ERR_INVALID, "offset_bits must be 32 or 64");
    if (!d->offsets || !d->char_class || (!d->values && d->num_entries))
        return fail(EPIK_AMD_ERR_INVALID, "null database pointer");
    // dense key space: num_keys == sigma^k, and codes are 32-bit on the device
    {
        uint64_t nk = 1;
        for (uint32_t i = 0; i < d->kmer_size; ++i) {
            nk *= d->alphabet_size;
            if (nk > 0xffffffffull) return fail(EPIK_AMD_ERR_UNSUPPORTED, "alphabet_size^kmer_size exceeds 2^32 keys");
        }
        if (nk != d->num_keys) return fail(EPIK_AMD_ERR_INVALID, "num_keys != alphabet_size^kmer_size");
    }
    // offsets must be monotone and end at num_entries (cheap checks of both ends)
    {
        uint64_t first, last;
        if (d->offset_bits == 32) {
            const uint32_t *o = static_cast<const uint32_t *>(d->offsets);
            first = o[0];
            last = o[d->num_keys];
            if (d->num_entries > 0xffffffffull) return fail(EPIK_AMD_ERR_INVALID, "num_entries needs 64-bit offsets");
        } else {
            const uint64_t *o = static_cast<const uint64_t *>(d->offsets);
            first = o[0];
            last = o[d->num_keys];
        }
        if (first != 0 || last != d->num_entries)
            return fail(EPIK_AMD_ERR_INVALID, "offsets[0] != 0 or offsets[num_keys] != num_entries");
    }

    // The lists themselves (host-only checks, before any device is touched): monotone offsets,
    // lists shorter than 2^24, every branch below num_branches, finite scores, and -- what the kernel's
    // lane-parallel read-add-write of a list relies on -- no branch twice in one list
    // (the reference's lists are built per branch, one score each: main.cpp:257).
    {
        const bool o64 = d->offset_bits == 64;
        auto off = [&](uint64_t key) -> uint64_t {
            return o64 ? static_cast<const uint64_t *>(d->offsets)[key] : static_cast<const uint32_t *>(d->offsets)[key];
        };
        std::vector<uint32_t> seen_in;  // seen_in[b] = 1 + the last list that held branch b
        try {
            seen_in.assign(d->num_branches, 0);
        } catch (const std::bad_alloc &) {
            return fail(EPIK_AMD_ERR_INVALID, "out of host memory");
        }
        uint32_t list_id = 0;
        for (uint64_t key = 0; key < d->num_keys; ++key) {
            const uint64_t b = off(key), e = off(key + 1);
            if (e < b || e > d->num_entries) return fail(EPIK_AMD_ERR_INVALID, "offsets not monotone");
            if (e - b >= (1ull << 24)) return fail(EPIK_AMD_ERR_INVALID, "posting list of 2^24 entries or more");
            if (e == b) continue;
            if (++list_id == 0) {  // the list counter wrapped (> 4 G non-empty lists): start a new epoch
                std::fill(seen_in.begin(), seen_in.end(), 0u);
                list_id = 1;
            }
            for (uint64_t i = b; i < e; ++i) {
                const uint32_t branch = d->values[i].branch;
                if (branch >= d->num_branches)
                    return fail(EPIK_AMD_ERR_INVALID, "posting with branch >= num_branches");
                if (!std::isfinite(d->values[i].score))  // the kernel marks "no edge" with -inf
                    return fail(EPIK_AMD_ERR_INVALID, "posting with a non-finite score");
                if (seen_in[branch] == list_id)
                    return fail(EPIK_AMD_ERR_INVALID, "a posting list names the same branch twice");
                seen_in[branch] = list_id;
            }
        }
    }

    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(EPIK_AMD_ERR_NO_DEVICE, "no HIP device available (libepik_amd has no CPU fallback)");
    if (d->device < 0 || d->device >= n_dev) return fail(EPIK_AMD_ERR_INVALID, "device ordinal out of range");
    HIP_TRY(hipSetDevice(d->device));

    epik_amd_placer *p = new (std::nothrow) epik_amd_placer();
    if (!p) return fail(EPIK_AMD_ERR_INVALID, "out of host memory");
#define CREATE_TRY(expr)                                  \
    do {                                                  \
        const hipError_t e_ = (expr);                     \
        if (e_ != hipSuccess) {                           \
            epik_amd_placer_destroy(p);                   \
            return fail_hip(e_, #expr);                   \
        }                                                 \
    } while (0)
    p->device = d->device;
    p->offsets64 = d->offset_bits == 64;
    p->num_keys = d->num_keys;
    p->num_entries = d->num_entries;


    // LDS rows per wave: the branches + the dummy row that out-of-range lanes of the posting
    // loads fall on; a multiple of 64 (the epilogue's sweeps).  Postings carry a 16-bit cell =
    // n_pad - 1 - branch.
    epik_amd::PlaceParams &pp = p->params;
    pp.n_pad = (d->num_branches + 1u + 63u) & ~63u;
    if (pp.n_pad * 8u + (EPIK_AMD_TILES_PER_PASS * 64u + EPIK_AMD_RING) * 8u > kMaxLdsPerBlock) {  // the 32-bit-count kernels
        epik_amd_placer_destroy(p);
        return fail(EPIK_AMD_ERR_UNSUPPORTED, "num_branches too large for the LDS-resident score vector");
    }

    // ---- choose the HBM layout ---------------------------------------------------------
    //  packed : an 8-byte {len, line} entry per k-mer code + every list on 128-byte lines of its own;
    //  paired : the same lists behind a table keyed by the (k-1)-mer that two consecutive k-mers of
    //           a read share: one table line per two lookups, 16 bytes per code; 4-letter alphabets
    //           only (place_kernel.hip: PackedLayout<true>) -- the default for them;
    //  compact: the CSR (4- or 8-byte offsets), 8-byte postings back to back.
    // paired / packed are chosen when the table is at most a quarter of the device's free memory;
    //  filtered: packed behind a presence filter keyed like the paired table: chosen for the other
    //           alphabets when at most a quarter of the codes have a list (for 4 letters it measures
    //           the same as paired, sparse or not);
    // EPIK_AMD_LAYOUT=compact|packed|paired|filtered overrides (paired means filtered for other alphabets).
    size_t free_mem = 0, total_mem = 0;
    CREATE_TRY(hipMemGetInfo(&free_mem, &total_mem));
    const char *lay = std::getenv("EPIK_AMD_LAYOUT");
    if (lay && std::strcmp(lay, "compact") != 0 && std::strcmp(lay, "packed") != 0 && std::strcmp(lay, "paired") != 0 &&
        std::strcmp(lay, "filtered") != 0) {
        epik_amd_placer_destroy(p);
        return fail(EPIK_AMD_ERR_INVALID, "EPIK_AMD_LAYOUT must be compact, packed, paired or filtered");
    }
    // filtered: one presence word per (k-1)-mer (2 * sigma bits) in front of the packed table, when
    // few codes have a list: then most lookups end at the filter, two per fetched line
    uint64_t present_codes = 0;
    for (uint64_t key = 0; key < d->num_keys; ++key)
        present_codes += (p->offsets64 ? static_cast<const uint64_t *>(d->offsets)[key + 1] != static_cast<const uint64_t *>(d->offsets)[key]
                                       : static_cast<const uint32_t *>(d->offsets)[key + 1] != static_cast<const uint32_t *>(d->offsets)[key]);
    const bool sparse = present_codes * 4u <= d->num_keys;
    const bool can_pair = d->alphabet_size == 4, can_filter = d->alphabet_size <= 32;
    const bool table_fits = d->num_keys * 16u <= free_mem / 4;
    bool paired = false, filtered = false, packed = false;
    if (lay) {
        filtered = can_filter && (std::strcmp(lay, "filtered") == 0 || (!can_pair && std::strcmp(lay, "paired") == 0));
        paired = can_pair && std::strcmp(lay, "paired") == 0;
        packed = paired || filtered || std::strcmp(lay, "compact") != 0;
    } else if (table_fits) {
        paired = can_pair;  // for 4 letters the paired table already reads one line per two lookups
        filtered = !paired && can_filter && sparse;
        packed = true;
    }
    auto offset_at = [&](uint64_t key) -> uint64_t {
        return p->offsets64 ? static_cast<const uint64_t *>(d->offsets)[key]
                            : static_cast<const uint32_t *>(d->offsets)[key];
    };
    // k-mer-space shard: this placer keeps the lists of the codes with code % count == index
    auto kept_len = [&](uint64_t key) -> uint64_t {
        return (shard_count == 1 || key % shard_count == shard_index) ? offset_at(key + 1) - offset_at(key) : 0;
    };
    uint64_t lines = 0;  // packed: 128-byte lines of the posting region
    uint64_t kept_entries = 0;
    for (uint64_t key = 0; key < d->num_keys; ++key) {
        const uint64_t len = kept_len(key);
        kept_entries += len;
        lines += (len * 6u + 127u) / 128u;
    }
    if (packed && lines >= (1ull << 32)) {
        epik_amd_placer_destroy(p);
        return fail(EPIK_AMD_ERR_UNSUPPORTED, "posting region of 512 GiB or more");
    }
    {
        // +512: room behind the last list (descriptors are exact, nothing reads it)
        p->db_bytes = (packed ? lines * 128u : kept_entries * 8u) + 512u;
        std::vector<uint8_t> host;
        std::vector<uint32_t> table;  // packed: {len, line} per code
        try {
            host.assign(p->db_bytes, 0);
            if (packed) table.assign(d->num_keys * (paired ? 4 : 2), 0);
        } catch (const std::bad_alloc &) {
            epik_amd_placer_destroy(p);
            return fail(EPIK_AMD_ERR_INVALID, "out of host memory building the device database");
        }
        const uint32_t top = pp.n_pad - 1u;
        // packed: one list = chunks of <= 64 postings, each chunk f32 score[cnt] then u16 cell[cnt]
        auto write_list = [&](uint8_t *dst, const epik_amd_pkdb_value *src, uint64_t len) {
            for (uint64_t c0 = 0; c0 < len; c0 += 64) {
                const uint32_t cnt = (uint32_t)((len - c0 < 64) ? len - c0 : 64);
                for (uint32_t j = 0; j < cnt; ++j) {
                    const uint16_t cell = (uint16_t)(top - src[c0 + j].branch);
                    std::memcpy(dst + 4u * j, &src[c0 + j].score, 4);
                    std::memcpy(dst + 4u * cnt + 2u * j, &cell, 2);
                }
                dst += (size_t)cnt * 6u;
            }
        };
        if (packed) {
            p->layout = paired     ? epik_amd::DbLayout::kPaired
                        : filtered ? epik_amd::DbLayout::kFiltered
                                   : epik_amd::DbLayout::kPacked;
            // paired: the entry of code c = a.X = Y.b (X its last k-1 letters, Y its first k-1) is stored
            // in block X at slot a and in block Y at slot 4 + b; a block is 8 entries
            const uint32_t shift = 2u * d->kmer_size - 2u;
            uint64_t line = 0;
            for (uint64_t key = 0; key < d->num_keys; ++key) {
                const uint64_t b = offset_at(key), len = kept_len(key);
                if (paired) {
                    const uint64_t as_suffix = ((key & ((1ull << shift) - 1ull)) * 8u + (key >> shift)) * 2u;
                    const uint64_t as_prefix = ((key >> 2) * 8u + 4u + (key & 3u)) * 2u;
                    table[as_suffix] = table[as_prefix] = (uint32_t)len;
                    table[as_suffix + 1] = table[as_prefix + 1] = (uint32_t)line;
                } else {
                    table[2 * key] = (uint32_t)len;
                    table[2 * key + 1] = (uint32_t)line;
                }
                write_list(host.data() + line * 128u, d->values + b, len);
                line += (len * 6u + 127u) / 128u;
            }
            CREATE_TRY(hipMalloc(&p->d_table, table.size() * 4u + 8u));
            CREATE_TRY(hipMemcpy(p->d_table, table.data(), table.size() * 4u, hipMemcpyHostToDevice));
            if (filtered) {
                // filter[X], X a (k-1)-mer: bit a <=> code a.X has a list, bit sigma + b <=> code X.b has one
                const uint64_t sigma = d->alphabet_size, blocks = d->num_keys / sigma;  // sigma^(k-1)
                std::vector<uint64_t> filter;
                try {
                    filter.assign(blocks, 0);
                } catch (const std::bad_alloc &) {
                    epik_amd_placer_destroy(p);
                    return fail(EPIK_AMD_ERR_INVALID, "out of host memory building the presence filter");
                }
                for (uint64_t key = 0; key < d->num_keys; ++key) {
                    if (kept_len(key) == 0) continue;
                    filter[key % blocks] |= 1ull << (key / blocks);          // as a.X: X = its last k-1 letters
                    filter[key / sigma] |= 1ull << (sigma + key % sigma);    // as X.b: X = its first k-1 letters
                }
                CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_filter), blocks * 8u));
                CREATE_TRY(hipMemcpy(p->d_filter, filter.data(), blocks * 8u, hipMemcpyHostToDevice));
                pp.filter = p->d_filter;
            }
        } else {
            p->layout = p->offsets64 ? epik_amd::DbLayout::kCompact64 : epik_amd::DbLayout::kCompact32;
            // the kept lists back to back as {f32 score, u32 cell}; with one shard the offsets
            // are the caller's, otherwise they are rebuilt over the kept lists
            std::vector<uint8_t> own_offsets;
            const size_t off_bytes = (size_t)(d->num_keys + 1) * (p->offsets64 ? 8 : 4);
            if (shard_count > 1) own_offsets.assign(off_bytes, 0);
            uint64_t at = 0;
            for (uint64_t key = 0; key < d->num_keys; ++key) {
                const uint64_t b = offset_at(key), len = kept_len(key);
                for (uint64_t j = 0; j < len; ++j, ++at) {
                    const uint32_t cell = top - d->values[b + j].branch;
                    std::memcpy(host.data() + 8u * at, &d->values[b + j].score, 4);
                    std::memcpy(host.data() + 8u * at + 4u, &cell, 4);
                }
                if (shard_count > 1) {
                    if (p->offsets64)
                        reinterpret_cast<uint64_t *>(own_offsets.data())[key + 1] = at;
                    else
                        reinterpret_cast<uint32_t *>(own_offsets.data())[key + 1] = (uint32_t)at;
                }
            }
            CREATE_TRY(hipMalloc(&p->d_table, off_bytes));
            CREATE_TRY(hipMemcpy(p->d_table, shard_count > 1 ? static_cast<const void *>(own_offsets.data()) : d->offsets,
                                 off_bytes, hipMemcpyHostToDevice));
        }
        CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_postings), p->db_bytes));
        CREATE_TRY(hipMemcpy(p->d_postings, host.data(), p->db_bytes, hipMemcpyHostToDevice));
        pp.table = p->d_table;
        pp.postings = p->d_postings;
    }
    CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_char_class), 256 * sizeof(uint32_t)));
    CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_total), sizeof(unsigned long long)));
    CREATE_TRY(hipMemcpy(p->d_char_class, d->char_class, 256 * sizeof(uint32_t), hipMemcpyHostToDevice));
    CREATE_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    CREATE_TRY(hipStreamCreateWithFlags(&p->stream_in, hipStreamNonBlocking));
    CREATE_TRY(hipStreamCreateWithFlags(&p->stream_out, hipStreamNonBlocking));
    CREATE_TRY(hipEventCreate(&p->ev_start));
    CREATE_TRY(hipEventCreate(&p->ev_stop));

    pp.char_class = p->d_char_class;
    pp.sigma_pow_km1 = (uint32_t)(d->num_keys / d->alphabet_size);
    pp.kmer_size = d->kmer_size;
    pp.alphabet_size = d->alphabet_size;
    pp.num_branches = d->num_branches;
    pp.keep_at_most = d->keep_at_most;
    pp.keep_factor = d->keep_factor;
    pp.threshold = d->threshold;
    pp.log_threshold = d->log_threshold;
#ifdef EPIK_AMD_ABLATION
    if (const char *ab = std::getenv("EPIK_AMD_ABLATE")) pp.ablate = (uint32_t)std::atoi(ab);
    if (const char *st = std::getenv("EPIK_AMD_STAMPS"); st && st[0] == '1') {
        CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&pp.dbg), 8 * sizeof(unsigned long long)));
        CREATE_TRY(hipMemset(pp.dbg, 0, 8 * sizeof(unsigned long long)));
    }
#endif
    hipDeviceProp_t prop;
    CREATE_TRY(hipGetDeviceProperties(&prop, d->device));
    for (int counts = 0; counts < 3; ++counts) {
        auto &g = p->geo[counts];
        // per wave: float32 scores + 8/16/32-bit counts + the chunk descriptors of one round
        // + one trip of spare entries (the kernel prefetches a trip ahead)
        const uint32_t desc_bytes = (EPIK_AMD_TILES_PER_PASS * 64u + EPIK_AMD_RING) * 8u;
        g.lds_wave_bytes = (pp.n_pad * (4u + (1u << counts)) + desc_bytes + 15u) & ~15u;
        if (counts == epik_amd::kCounts8 && (pp.n_pad + 7u) / 8u > desc_bytes) {
            g.max_blocks = 0;  // the 8-bit kernel keeps one flag bit per row in the descriptor area: no room
            continue;
        }
        if (g.lds_wave_bytes > kMaxLdsPerBlock) {
            epik_amd_placer_destroy(p);
            return fail(EPIK_AMD_ERR_UNSUPPORTED, "num_branches too large for the LDS-resident score vector");
        }
        // Workgroup of 4, 2 or 1 independent waves: whichever keeps the most waves resident on a
        // CU (ties: the larger workgroup).  LDS is handed out in units of 1280 bytes (160 KiB / 128;
        // measured: 5 x 32512 B did not fit a CU, 5 x 30976 B do), which the occupancy query does not
        // know -- and a workgroup that is not resident with the others runs behind them: with this
        // kernel's fixed stride over the reads that doubles the launch time.  The grid is exactly the
        // resident workgroups (registers, LDS and the waves-per-CU cap decide), each striding over the reads.
        uint32_t best_waves = 0;
        for (uint32_t wpb = 4; wpb >= 1; wpb >>= 1) {
            const uint32_t block_bytes = wpb * g.lds_wave_bytes;
            if (block_bytes > kMaxLdsPerBlock) continue;
            CREATE_TRY(epik_amd::set_place_reads_lds_limit(p->layout, counts, block_bytes));
            int per_cu = 0;
            CREATE_TRY(epik_amd::place_reads_occupancy(p->layout, counts, (int)(wpb * 64u), block_bytes, &per_cu));
            const uint32_t lds_units = (block_bytes + 1279u) / 1280u;
            per_cu = std::min<int>(per_cu, (int)(128u / std::max(lds_units, 1u)));
            if (per_cu < 1) per_cu = 1;
            if ((uint32_t)per_cu * wpb > best_waves) {
                best_waves = (uint32_t)per_cu * wpb;
                g.waves_per_block = wpb;
                g.lds_block_bytes = block_bytes;
                g.max_blocks = (uint32_t)prop.multiProcessorCount * (uint32_t)per_cu;
            }
        }
        g.resident_waves = best_waves;
        CREATE_TRY(epik_amd::set_place_reads_lds_limit(p->layout, counts, g.lds_block_bytes));
        CREATE_TRY(epik_amd::set_finish_reads_lds_limit(counts, g.lds_block_bytes));
    }
    // EPIK_AMD_WIDE_COUNTS=0|1|2: 16-, 32-, 8-bit counts whatever the reads (tests, experiments)
    if (const char *w = std::getenv("EPIK_AMD_WIDE_COUNTS")) {
        p->counts = w[0] == '1' ? epik_amd::kCounts32 : w[0] == '2' ? epik_amd::kCounts8 : epik_amd::kCounts16;
        if (p->counts == epik_amd::kCounts8 && p->geo[epik_amd::kCounts8].max_blocks == 0) p->counts = epik_amd::kCounts16;
        p->counts_forced = true;
    }
#undef CREATE_TRY

    *out = p;
    return EPIK_AMD_OK;
}

// The narrowest counts that hold the k-mers of a read of `longest` characters, 8 bits only when
// that keeps more waves on a CU than 16.
static int counts_for(const epik_amd_placer *p, uint64_t longest)
{
    const uint64_t kmers = longest >= p->params.kmer_size ? longest - p->params.kmer_size + 1 : 0;
    if (kmers >= 32768u) return epik_amd::kCounts32;
    if (kmers <= 255u && p->geo[epik_amd::kCounts8].resident_waves > p->geo[epik_amd::kCounts16].resident_waves)
        return epik_amd::kCounts8;
    return epik_amd::kCounts16;
}

static int launch(epik_amd_placer *p, const void *d_seqs, const void *d_seq_offsets, uint64_t n,
                  void *d_rows, void *d_n_rows, void *d_counts, hipStream_t stream,
                  float *partial_scores = nullptr, uint32_t *partial_counts = nullptr)
{
    if (n == 0) return EPIK_AMD_OK;
    epik_amd::PlaceParams pp = p->params;
    pp.partial_scores = partial_scores;  // non-null: accumulate only
    pp.partial_counts = partial_counts;
    pp.seqs = static_cast<const uint8_t *>(d_seqs);
    pp.seq_offsets = static_cast<const uint64_t *>(d_seq_offsets);
    pp.n_reads = n;
    pp.rows = static_cast<epik_amd_placement *>(d_rows);
    pp.n_rows = static_cast<uint32_t *>(d_n_rows);
    pp.kmer_counts = static_cast<uint32_t *>(d_counts);
    const auto &g = p->geo[p->counts];
    pp.lds_wave_bytes = g.lds_wave_bytes;
    uint64_t blocks = (n + g.waves_per_block - 1) / g.waves_per_block;
    if (blocks > g.max_blocks) blocks = g.max_blocks;
    p->last_blocks = (uint32_t)blocks;
    p->last_geo = (uint32_t)p->counts;
    if (p->timing) HIP_TRY(hipEventRecord(p->ev_start, stream));
    HIP_TRY(epik_amd::launch_place_reads(pp, p->layout, p->counts, dim3((unsigned)blocks),
                                         dim3(g.waves_per_block * 64u), g.lds_block_bytes, stream));
    if (p->timing) {
        HIP_TRY(hipEventRecord(p->ev_stop, stream));
        p->ev_recorded = true;
    }
    return EPIK_AMD_OK;
}

int epik_amd_placer_accumulate_device(epik_amd_placer *p, const void *d_seqs, const void *d_seq_offsets,
                                      uint64_t n, void *d_scores, void *d_counts, void *stream)
{
    if (!p) return fail(EPIK_AMD_ERR_INVALID, "null placer");
    if (n && (!d_seqs || !d_seq_offsets || !d_scores || !d_counts))
        return fail(EPIK_AMD_ERR_INVALID, "null device buffer");
    HIP_TRY(hipSetDevice(p->device));
    return launch(p, d_seqs, d_seq_offsets, n, nullptr, nullptr, nullptr, static_cast<hipStream_t>(stream),
                  static_cast<float *>(d_scores), static_cast<uint32_t *>(d_counts));
}

int epik_amd_placer_finish_device(epik_amd_placer *p, const void *d_seq_offsets, uint64_t n,
                                  const void *d_scores, const void *d_counts, void *d_rows, void *d_n_rows,
                                  void *d_kmer_counts, void *stream)
{
    if (!p) return fail(EPIK_AMD_ERR_INVALID, "null placer");
    if (n == 0) return EPIK_AMD_OK;
    if (!d_seq_offsets || !d_scores || !d_counts || !d_rows || !d_n_rows)
        return fail(EPIK_AMD_ERR_INVALID, "null device buffer");
    HIP_TRY(hipSetDevice(p->device));
    epik_amd::PlaceParams pp = p->params;
    pp.seq_offsets = static_cast<const uint64_t *>(d_seq_offsets);
    pp.n_reads = n;
    pp.rows = static_cast<epik_amd_placement *>(d_rows);
    pp.n_rows = static_cast<uint32_t *>(d_n_rows);
    pp.kmer_counts = static_cast<uint32_t *>(d_kmer_counts);
    pp.partial_scores = const_cast<float *>(static_cast<const float *>(d_scores));
    pp.partial_counts = const_cast<uint32_t *>(static_cast<const uint32_t *>(d_counts));
    const auto &g = p->geo[p->counts];
    pp.lds_wave_bytes = g.lds_wave_bytes;
    uint64_t blocks = (n + g.waves_per_block - 1) / g.waves_per_block;
    if (blocks > g.max_blocks) blocks = g.max_blocks;
    HIP_TRY(epik_amd::launch_finish_reads(pp, p->counts, dim3((unsigned)blocks), dim3(g.waves_per_block * 64u),
                                          g.lds_block_bytes, static_cast<hipStream_t>(stream)));
    return EPIK_AMD_OK;
}

int epik_amd_placer_place_device(epik_amd_placer *p, const void *d_seqs, const void *d_seq_offsets,
                                 uint64_t n, void *d_rows, void *d_n_rows, void *d_kmer_counts,
                                 void *stream)
{
    if (!p) return fail(EPIK_AMD_ERR_INVALID, "null placer");
    if (n && (!d_seqs || !d_seq_offsets || !d_rows || !d_n_rows))
        return fail(EPIK_AMD_ERR_INVALID, "null device buffer");
    HIP_TRY(hipSetDevice(p->device));
    return launch(p, d_seqs, d_seq_offsets, n, d_rows, d_n_rows, d_kmer_counts,
                  static_cast<hipStream_t>(stream));
}

// Reads per chunk of the host-buffer entry point: about 16 MB of sequence each, so that
// the copy-in of chunk c+1 and the copy-out of chunk c-1 run under the kernel of chunk c.
static uint64_t host_chunk_reads(uint64_t n, size_t seq_bytes)
{
    uint64_t chunks = (seq_bytes + (16u << 20) - 1) / (16u << 20);
    if (const char *e = std::getenv("EPIK_AMD_HOST_CHUNKS")) chunks = std::strtoull(e, nullptr, 10);
    chunks = std::min<uint64_t>(std::max<uint64_t>(chunks, 1), 256);
    return std::max<uint64_t>((n + chunks - 1) / chunks, 1);
}

int epik_amd_placer_place(epik_amd_placer *p, const char *seqs, const uint64_t *seq_offsets,
                          uint64_t n, epik_amd_placement *rows, uint32_t *n_rows,
                          uint32_t *kmer_counts)
{
    if (!p) return fail(EPIK_AMD_ERR_INVALID, "null placer");
    if (n == 0) return EPIK_AMD_OK;
    if (!seqs || !seq_offsets || !rows || !n_rows) return fail(EPIK_AMD_ERR_INVALID, "null host buffer");
    if (seq_offsets[0] != 0) return fail(EPIK_AMD_ERR_INVALID, "seq_offsets[0] must be 0");
    uint64_t longest = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (seq_offsets[i + 1] < seq_offsets[i] || seq_offsets[i + 1] - seq_offsets[i] > 0xffffffffull)
            return fail(EPIK_AMD_ERR_INVALID, "seq_offsets not monotone, or a read of 2^32 characters or more");
        longest = std::max<uint64_t>(longest, seq_offsets[i + 1] - seq_offsets[i]);
    }
    HIP_TRY(hipSetDevice(p->device));
    // The narrowest counts that hold the longest read's k-mers: 16 bits normally, 32 for a read of
    // 32768 k-mers or more, 8 (reads of up to 255 k-mers) when that puts more waves on a CU.
    const int saved_counts = p->counts;
    if (!p->counts_forced) p->counts = counts_for(p, longest);
    struct restore_counts {
        epik_amd_placer *p;
        int v;
        ~restore_counts() { p->counts = v; }
    } restore{p, saved_counts};
    const size_t seq_bytes = (size_t)seq_offsets[n];
    if (seq_bytes + 64 > p->d_seqs_cap) {
        (void)hipFree(p->d_seqs);
        p->d_seqs = nullptr;
        p->d_seqs_cap = 0;
        const size_t cap = seq_bytes + seq_bytes / 4 + 4096;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_seqs), cap));
        p->d_seqs_cap = cap;
    }
    if (n > p->d_reads_cap) {
        (void)hipFree(p->d_seq_offsets);
        (void)hipFree(p->d_rows);
        (void)hipFree(p->d_n_rows);
        (void)hipFree(p->d_counts);
        p->d_seq_offsets = nullptr;
        p->d_rows = nullptr;
        p->d_n_rows = nullptr;
        p->d_counts = nullptr;
        p->d_reads_cap = 0;
        const size_t cap = (size_t)n + (size_t)n / 4 + 64;
        const size_t keep = p->params.keep_at_most;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_seq_offsets), (cap + 1) * sizeof(uint64_t)));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_rows), cap * keep * sizeof(epik_amd_placement)));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_n_rows), cap * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_counts), cap * keep * sizeof(uint32_t)));
        p->d_reads_cap = cap;
    }

    // Chunked pipeline over three streams: this thread copies chunk c in (stream_in) and
    // launches its kernel (stream); a helper thread copies finished chunks out (stream_out),
    // so the two directions of the host link and the kernel all overlap.  Device buffers hold
    // the whole batch (offsets stay absolute), only the transfers are chunked.
    const size_t keep = p->params.keep_at_most;
    const uint64_t per_chunk = host_chunk_reads(n, seq_bytes);
    const uint64_t n_chunks = (n + per_chunk - 1) / per_chunk;
    while (p->ev_in.size() < n_chunks) {
        hipEvent_t a = nullptr, b = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        p->ev_in.push_back(a);
        HIP_TRY(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        p->ev_kernel.push_back(b);
    }
    std::atomic<uint64_t> launched{0};
    std::atomic<bool> abort_out{false};
    hipError_t out_err = hipSuccess;
    const auto copy_out_fn = [&] {
        hipError_t e = hipSetDevice(p->device);
        for (uint64_t c = 0; c < n_chunks && e == hipSuccess; ++c) {
            while (launched.load(std::memory_order_acquire) <= c) {
                if (abort_out.load(std::memory_order_acquire)) return;
                std::this_thread::yield();
            }
            const uint64_t r0 = c * per_chunk, cnt = std::min(per_chunk, n - r0);
            e = hipStreamWaitEvent(p->stream_out, p->ev_kernel[c], 0);
            if (e == hipSuccess)
                e = hipMemcpyAsync(rows + r0 * keep, p->d_rows + r0 * keep, cnt * keep * sizeof(epik_amd_placement),
                                   hipMemcpyDeviceToHost, p->stream_out);
            if (e == hipSuccess)
                e = hipMemcpyAsync(n_rows + r0, p->d_n_rows + r0, cnt * sizeof(uint32_t), hipMemcpyDeviceToHost,
                                   p->stream_out);
            if (e == hipSuccess && kmer_counts)
                e = hipMemcpyAsync(kmer_counts + r0 * keep, p->d_counts + r0 * keep, cnt * keep * sizeof(uint32_t),
                                   hipMemcpyDeviceToHost, p->stream_out);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(p->stream_out);
        out_err = e;
    };
    std::thread copy_out;  // a single chunk has nothing to overlap: no helper thread then
    if (n_chunks > 1) copy_out = std::thread(copy_out_fn);
    int rc = EPIK_AMD_OK;
    hipError_t in_err = hipSuccess;
    for (uint64_t c = 0; c < n_chunks; ++c) {
        const uint64_t r0 = c * per_chunk, cnt = std::min(per_chunk, n - r0);
        const uint64_t b0 = seq_offsets[r0], b1 = seq_offsets[r0 + cnt];
        hipError_t e = hipSuccess;
        if (b1 > b0) e = hipMemcpyAsync(p->d_seqs + b0, seqs + b0, b1 - b0, hipMemcpyHostToDevice, p->stream_in);
        if (e == hipSuccess)
            e = hipMemcpyAsync(p->d_seq_offsets + r0, seq_offsets + r0, (cnt + 1) * sizeof(uint64_t),
                               hipMemcpyHostToDevice, p->stream_in);
        if (e == hipSuccess) e = hipEventRecord(p->ev_in[c], p->stream_in);
        if (e == hipSuccess) e = hipStreamWaitEvent(p->stream, p->ev_in[c], 0);
        // rows beyond n_rows[i] are never written by the kernel: give them a defined value
        if (e == hipSuccess)
            e = hipMemsetAsync(p->d_rows + r0 * keep, 0, cnt * keep * sizeof(epik_amd_placement), p->stream);
        if (e == hipSuccess) e = hipMemsetAsync(p->d_counts + r0 * keep, 0, cnt * keep * sizeof(uint32_t), p->stream);
        if (e != hipSuccess) {
            in_err = e;
            break;
        }
        rc = launch(p, p->d_seqs, p->d_seq_offsets + r0, cnt, p->d_rows + r0 * keep, p->d_n_rows + r0,
                    p->d_counts + r0 * keep, p->stream);
        if (rc != EPIK_AMD_OK) break;
        e = hipEventRecord(p->ev_kernel[c], p->stream);
        if (e != hipSuccess) {
            in_err = e;
            break;
        }
        launched.store(c + 1, std::memory_order_release);
    }
    if (rc != EPIK_AMD_OK || in_err != hipSuccess) abort_out.store(true, std::memory_order_release);
    if (copy_out.joinable())
        copy_out.join();
    else
        copy_out_fn();
    (void)hipStreamSynchronize(p->stream_in);
    (void)hipStreamSynchronize(p->stream);
    if (rc != EPIK_AMD_OK) return rc;
    if (in_err != hipSuccess) return fail_hip(in_err, "copy-in / launch of the host-buffer pipeline");
    if (out_err != hipSuccess) return fail_hip(out_err, "copy-out of the host-buffer pipeline");
    return EPIK_AMD_OK;
}

int epik_amd_placer_algorithmic_bytes(epik_amd_placer *p, const void *d_seqs,
                                      const void *d_seq_offsets, uint64_t n, const void *d_n_rows,
                                      void *stream, uint64_t *bytes_out)
{
    if (!p || !bytes_out) return fail(EPIK_AMD_ERR_INVALID, "null argument");
    *bytes_out = 0;
    if (n == 0) return EPIK_AMD_OK;
    HIP_TRY(hipSetDevice(p->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    epik_amd::PlaceParams pp = p->params;
    pp.seqs = static_cast<const uint8_t *>(d_seqs);
    pp.seq_offsets = static_cast<const uint64_t *>(d_seq_offsets);
    pp.n_reads = n;
    pp.n_rows = const_cast<uint32_t *>(static_cast<const uint32_t *>(d_n_rows));
    HIP_TRY(hipMemsetAsync(p->d_total, 0, sizeof(unsigned long long), s));
    HIP_TRY(epik_amd::launch_algorithmic_bytes(pp, p->layout, p->d_total, s));
    unsigned long long total = 0;
    HIP_TRY(hipMemcpyAsync(&total, p->d_total, sizeof(total), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    *bytes_out = total;
    return EPIK_AMD_OK;
}

int epik_amd_placer_launch_info(const epik_amd_placer *p, uint32_t *waves_per_block,
                                uint32_t *blocks, uint32_t *lds_bytes)
{
    if (!p) return fail(EPIK_AMD_ERR_INVALID, "null placer");
    const auto &g = p->geo[p->last_geo];
    if (waves_per_block) *waves_per_block = g.waves_per_block;
    if (blocks) *blocks = p->last_blocks ? p->last_blocks : g.max_blocks;
    if (lds_bytes) *lds_bytes = g.lds_block_bytes;
    return EPIK_AMD_OK;
}

int epik_amd_placer_set_wide_counts(epik_amd_placer *p, int enabled)
{
    if (!p) return fail(EPIK_AMD_ERR_INVALID, "null placer");
    if (enabled == 2 && p->geo[epik_amd::kCounts8].max_blocks == 0)
        return fail(EPIK_AMD_ERR_UNSUPPORTED, "no 8-bit-count kernel for this tree size");
    p->counts = enabled == 1 ? epik_amd::kCounts32 : enabled == 2 ? epik_amd::kCounts8 : epik_amd::kCounts16;
    p->counts_forced = enabled != 0;
    return EPIK_AMD_OK;
}

int epik_amd_placer_choose_counts(epik_amd_placer *p, uint64_t longest_read)
{
    if (!p) return fail(EPIK_AMD_ERR_INVALID, "null placer");
    p->counts = counts_for(p, longest_read);
    p->counts_forced = true;
    return EPIK_AMD_OK;
}

int epik_amd_placer_set_timing(epik_amd_placer *p, int enabled)
{
    if (!p) return fail(EPIK_AMD_ERR_INVALID, "null placer");
    p->timing = enabled != 0;
    p->ev_recorded = false;
    return EPIK_AMD_OK;
}

int epik_amd_placer_last_kernel_ms(epik_amd_placer *p, float *ms_out)
{
    if (!p || !ms_out) return fail(EPIK_AMD_ERR_INVALID, "null argument");
    *ms_out = -1.0f;
    if (!p->ev_recorded) return EPIK_AMD_OK;
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipEventSynchronize(p->ev_stop));
    HIP_TRY(hipEventElapsedTime(ms_out, p->ev_start, p->ev_stop));
    return EPIK_AMD_OK;
}

}  // extern "C"
