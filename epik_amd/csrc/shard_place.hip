// shard_place.hip -- epik_amd_placer_place_sharded: one database, k-mer-space sharded over several handles
// (devices), placed through the partial lists of include/epik_amd.h -- the native form of
// epik_amd/dist.py: place_kmer_sharded_lists, for a process that drives all the devices itself.
//
// Reference path: the same loop as everything else here (epik::placer::place, place.cpp:201-275; the per-branch
// sums of place.cpp:349-371 split over the shards, correction / top-k / like-weight-ratio of :418-422, :134-199,
// :241-267 on the finisher).  No reference counterpart for the split itself: the reference keeps one database
// in host RAM (main.cpp:277).
//
// Per chunk of the batch (a few ten thousand reads):
//   accumulate  every handle g, on its device and stream: the lists of ALL reads of the chunk against its codes,
//               as G parts (part r = the reads finisher r owns);
//   exchange    part r of every handle goes to the device of handle r: one peer copy per (source, finisher)
//               pair, on the finisher's copy stream -- each pair has its own xGMI link, nothing is a ring --,
//               behind the source's accumulate only;
//   finish      handle r adds the G lists of each of its reads in shard order and finishes the placement; the
//               rows go home.
// The accumulate of chunk c + 1 is enqueued before the exchange of chunk c is: on every device the copies of
// one chunk run under the kernels of the next (two sets of buffers).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "placer_impl.hpp"

namespace {

using epik_amd::fail_with;

#define SHARD_TRY(expr)                                                                                  \
    do {                                                                                                 \
        const hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail_with(EPIK_AMD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

constexpr uint32_t kAmbNone = 0xffffffffu;

// Per branch the average probability of the ambiguous key of smallest order over the shards, 0 where none has
// one (place.cpp:385-388: only the first ambiguous key that reaches a branch scores it -- first over the whole
// database; epik_amd/dist.py: combine_amb).  order / avg: [n_shards] arrays of [rows][num_branches].
struct AmbSources {
    const uint32_t *order[EPIK_AMD_MAX_SHARDS];
    const float *avg[EPIK_AMD_MAX_SHARDS];
    uint32_t n_shards;
};
__global__ void combine_amb_kernel(AmbSources src, uint64_t cells, float *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cells) return;
    uint32_t best = kAmbNone;
    float avg = 0.0f;
    for (uint32_t g = 0; g < src.n_shards; ++g) {
        const uint32_t o = src.order[g][i];
        if (o < best) best = o, avg = src.avg[g][i];  // (a key lives in exactly one shard: no ties)
    }
    out[i] = avg;
}

// a device buffer that only grows
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int device = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return e;
        if (p) (void)hipFree(p);
        p = nullptr, cap = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release()
    {
        if (p) {
            (void)hipSetDevice(device);
            (void)hipFree(p);
        }
        p = nullptr, cap = 0;
    }
    template <typename T>
    T *as() const { return static_cast<T *>(p); }
};

// what one handle needs for a sharded placement, two sets (chunk c and chunk c + 1 overlap)
struct ShardSide {
    epik_amd_placer *h = nullptr;
    int device = 0;
    hipStream_t compute = nullptr, copy = nullptr;
    // as a source: its lists of the chunk
    DevBuf seqs, offsets;                    // the whole batch (every shard sees all reads)
    DevBuf entries[2], index[2], part_entries[2], amb_slot[2], amb_order[2], amb_avg[2];
    uint64_t entries_cap[2] = {0, 0};
    unsigned long long *h_part[2] = {nullptr, nullptr};  // pinned: part sizes of the chunk
    // pinned: the chunk's slot arrays (reads that may hold an ambiguous k-mer) on their way to the device -- a copy
    // from pageable memory would hold the calling thread behind everything queued on the compute stream, once per
    // shard and chunk, and with it the overlap of a chunk's copies with the next chunk's kernels
    int32_t *h_slots[2] = {nullptr, nullptr};
    size_t h_slots_cap[2] = {0, 0};  // in int32
    hipEvent_t accumulated[2] = {nullptr, nullptr};
    // as a finisher: the parts it received, its rows
    DevBuf recv_entries[2][EPIK_AMD_MAX_SHARDS], recv_index[2][EPIK_AMD_MAX_SHARDS];
    DevBuf recv_order[2][EPIK_AMD_MAX_SHARDS], recv_avg[2][EPIK_AMD_MAX_SHARDS], my_avg[2], my_slot[2];
    DevBuf rows[2], n_rows[2], counts[2];
    hipEvent_t arrived[2] = {nullptr, nullptr}, finished[2] = {nullptr, nullptr};
    // its rows on their way home: pinned, so that the copy out of the device does not hold the calling thread (a
    // copy into the caller's pageable arrays would, until everything queued in front of it has run -- the next
    // chunk's accumulate among it); handed to the caller's arrays when the set is waited for
    struct HostOut {
        void *p = nullptr;
        size_t cap = 0;
        hipError_t reserve(size_t bytes)
        {
            if (bytes <= cap) return hipSuccess;
            if (p) (void)hipHostFree(p);
            p = nullptr, cap = 0;
            const hipError_t e = hipHostMalloc(&p, bytes + bytes / 4 + 4096, hipHostMallocDefault);
            if (e == hipSuccess) cap = bytes + bytes / 4 + 4096;
            return e;
        }
        void release()
        {
            if (p) (void)hipHostFree(p);
            p = nullptr, cap = 0;
        }
    } h_rows[2], h_n_rows[2], h_counts[2];
    uint64_t out_at[2] = {0, 0}, out_reads[2] = {0, 0};  // which reads of the batch set b's rows belong to
};

// The sides of a set of handles, with their buffers: kept on the set's first handle from call to call (a driver
// places group after group of batches through the same handles), freed with it.
struct ShardState {
    std::vector<epik_amd_placer *> handles;
    std::vector<uint64_t> generations;  // (of the handles: an address may come back from a later create())
    std::vector<ShardSide> sides;
    bool ready = false;
    ~ShardState()
    {
        for (auto &s : sides) {
            (void)hipSetDevice(s.device);
            s.seqs.release(), s.offsets.release();
            for (int b = 0; b < 2; ++b) {
                s.entries[b].release(), s.index[b].release(), s.part_entries[b].release();
                s.amb_slot[b].release(), s.amb_order[b].release(), s.amb_avg[b].release();
                s.my_avg[b].release(), s.my_slot[b].release(), s.rows[b].release(), s.n_rows[b].release(), s.counts[b].release();
                for (int g = 0; g < EPIK_AMD_MAX_SHARDS; ++g)
                    s.recv_entries[b][g].release(), s.recv_index[b][g].release(), s.recv_order[b][g].release(), s.recv_avg[b][g].release();
                s.h_rows[b].release(), s.h_n_rows[b].release(), s.h_counts[b].release();
                if (s.h_part[b]) (void)hipHostFree(s.h_part[b]);
                if (s.h_slots[b]) (void)hipHostFree(s.h_slots[b]);
                if (s.accumulated[b]) (void)hipEventDestroy(s.accumulated[b]);
                if (s.arrived[b]) (void)hipEventDestroy(s.arrived[b]);
                if (s.finished[b]) (void)hipEventDestroy(s.finished[b]);
            }
        }
    }
};
void free_shard_state(void *p) { delete static_cast<ShardState *>(p); }

// whatever way a call ends, nothing of it is still running on the handles' streams
struct Quiesce {
    std::vector<ShardSide> &sides;
    ~Quiesce()
    {
        for (auto &s : sides) {
            (void)hipSetDevice(s.device);
            if (s.compute) (void)hipStreamSynchronize(s.compute);
            if (s.copy) (void)hipStreamSynchronize(s.copy);
        }
    }
};

// device-to-device, between any two devices (the same one: a plain copy), on `stream` (of the destination's device)
hipError_t peer_copy(void *dst, int dst_dev, const void *src, int src_dev, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return hipSuccess;
    if (dst_dev == src_dev) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream);
    return hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, stream);
}

int place_sharded_impl(epik_amd_placer *const *shards, uint32_t G, const char *seqs, const uint64_t *seq_offsets,
                       uint64_t n, epik_amd_placement *rows, uint32_t *n_rows, uint32_t *kmer_counts)
{
    if (!shards || G == 0 || G > EPIK_AMD_MAX_SHARDS) return fail_with(EPIK_AMD_ERR_INVALID, "n_shards must be in [1, EPIK_AMD_MAX_SHARDS]");
    for (uint32_t g = 0; g < G; ++g)
        if (!shards[g]) return fail_with(EPIK_AMD_ERR_INVALID, "null placer");
    if (n == 0) return EPIK_AMD_OK;
    if (!seqs || !seq_offsets || !rows || !n_rows) return fail_with(EPIK_AMD_ERR_INVALID, "null host buffer");
    if (seq_offsets[0] != 0) return fail_with(EPIK_AMD_ERR_INVALID, "seq_offsets[0] must be 0");
    // ONE shard is the whole of what there is to place against: the one-pass placement, which streams a read's rows
    // once -- accumulate + finish stream them twice (as postings, then as list entries), 2.3 x the time for the same
    // rows (measured, round 4: 16.7 against 38.5 M reads/s at N = 9 999).  EPIK_AMD_SHARD_HALVES=1: the two halves
    // all the same (tests: the machinery below with one handle).
    if (G == 1) {
        const char *halves = std::getenv("EPIK_AMD_SHARD_HALVES");
        if (!(halves && halves[0] == '1')) return epik_amd_placer_place(shards[0], seqs, seq_offsets, n, rows, n_rows, kmer_counts);
    }
    // every handle: the same tree and parameters, kernels that leave partial lists, the same list geometry
    epik_amd_partial_info info0{};
    for (uint32_t g = 0; g < G; ++g) {
        epik_amd_partial_info info{};
        if (const int rc = epik_amd_placer_partial_info(shards[g], &info); rc != EPIK_AMD_OK) return rc;
        if (!info.lists)
            return fail_with(EPIK_AMD_ERR_UNSUPPORTED, "place_sharded needs the kernels of a large tree (partial lists); a small "
                                                       "tree's database fits one device: replicate it (--devices)");
        if (g == 0) info0 = info;
        const auto &a = shards[0]->params, &b = shards[g]->params;
        if (info.slices != info0.slices || info.slice_rows != info0.slice_rows || a.num_branches != b.num_branches ||
            a.kmer_size != b.kmer_size || a.alphabet_size != b.alphabet_size || a.keep_at_most != b.keep_at_most ||
            a.keep_factor != b.keep_factor || a.log_threshold != b.log_threshold)
            return fail_with(EPIK_AMD_ERR_INVALID, "the shards were not created from the same tree and parameters");
    }
    const uint32_t S = info0.slices, N = shards[0]->params.num_branches, keep = shards[0]->params.keep_at_most;
    const uint32_t k = shards[0]->params.kmer_size;
    uint64_t longest = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (seq_offsets[i + 1] < seq_offsets[i] || seq_offsets[i + 1] - seq_offsets[i] > 0xffffffffull)
            return fail_with(EPIK_AMD_ERR_INVALID, "seq_offsets not monotone, or a read of 2^32 characters or more");
        const uint64_t len = seq_offsets[i + 1] - seq_offsets[i];
        longest = std::max(longest, len);
    }
    // the same count width everywhere (it is the format of the entries): what the longest read needs
    for (uint32_t g = 0; g < G; ++g)
        if (const int rc = epik_amd_placer_choose_counts(shards[g], longest); rc != EPIK_AMD_OK) return rc;
    // (every handle chooses by its own geometry and occupancy: handles that ended up apart -- devices of two kinds, a
    // width forced on one of them -- all take the widest, or the finisher would read 8-byte entries as 16-byte ones)
    {
        int widest = shards[0]->counts;
        bool same = true;
        for (uint32_t g = 1; g < G; ++g) same = same && shards[g]->counts == widest, widest = std::max(widest, shards[g]->counts);
        if (!same)
            for (uint32_t g = 0; g < G; ++g) {
                if (shards[g]->geo[widest].max_blocks == 0)
                    return fail_with(EPIK_AMD_ERR_UNSUPPORTED, "the shards chose different count widths and one of them has no kernel of the widest");
                shards[g]->counts = widest;
            }
    }
    uint32_t entry_bytes = 8;
    {
        epik_amd_partial_info info{};
        (void)epik_amd_placer_partial_info(shards[0], &info);
        entry_bytes = info.entry_bytes;
    }
    // which reads may hold an ambiguous k-mer: any character that is not one plain state (dist.py: amb_slots)
    const std::vector<uint32_t> &cls = shards[0]->h_char_class;
    std::vector<uint8_t> dirty(n, 0);
    {
        uint8_t not_plain[256];
        for (int c = 0; c < 256; ++c) not_plain[c] = (cls[c] == 0 || (cls[c] & (cls[c] - 1)) != 0) ? 1 : 0;
        auto scan = [&](uint64_t begin, uint64_t end) {
            for (uint64_t i = begin; i < end; ++i) {
                const unsigned char *s = reinterpret_cast<const unsigned char *>(seqs) + seq_offsets[i];
                const uint64_t len = seq_offsets[i + 1] - seq_offsets[i];
                uint8_t any = 0;
                for (uint64_t c = 0; c < len; ++c) any |= not_plain[s[c]];  // (no early exit: nearly every read is clean)
                dirty[i] = any;
            }
        };
        // (a byte per character at a load each: tens of milliseconds per hundred megabytes on one core -- more than
        // the devices take for the batch -- so a few threads share a large batch)
        const unsigned workers = n >= 65536 ? std::min(8u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
        if (workers <= 1) {
            scan(0, n);
        } else {
            std::vector<std::thread> threads;
            for (unsigned t = 0; t < workers; ++t) threads.emplace_back(scan, n * t / workers, n * (t + 1) / workers);
            for (auto &t : threads) t.join();
        }
    }

    // ---- chunks of the batch: about 16 MB of sequence or 32768 reads, whichever is less (EPIK_AMD_SHARD_CHUNK: tests)
    uint64_t chunk_reads = 32768;
    if (const char *e = std::getenv("EPIK_AMD_SHARD_CHUNK")) chunk_reads = std::max<uint64_t>(1, std::strtoull(e, nullptr, 10));
    {
        const uint64_t mean = std::max<uint64_t>(1, seq_offsets[n] / n);
        chunk_reads = std::max<uint64_t>(1, std::min<uint64_t>(chunk_reads, (16ull << 20) / mean));
    }
    const uint64_t n_chunks = (n + chunk_reads - 1) / chunk_reads;

    // the sides of this set of handles: from the last call, or new
    ShardState *state = static_cast<ShardState *>(shards[0]->shard_state);
    // (the same handles -- and still the same streams: a handle destroyed and another created at its address is not it)
    bool same = state && state->handles.size() == G && std::equal(state->handles.begin(), state->handles.end(), shards);
    for (uint32_t g = 0; same && g < G; ++g) same = state->generations[g] == shards[g]->generation;
    for (uint32_t g = 0; same && state->ready && g < G; ++g)
        same = state->sides[g].compute == shards[g]->stream && state->sides[g].copy == shards[g]->stream_in &&
               state->sides[g].device == shards[g]->device;
    if (state && !same) {
        delete state;
        state = nullptr;
        shards[0]->shard_state = nullptr;
    }
    if (!state) {
        state = new ShardState();
        state->handles.assign(shards, shards + G);
        for (uint32_t g = 0; g < G; ++g) state->generations.push_back(shards[g]->generation);
        state->sides.resize(G);
        shards[0]->shard_state = state;
        shards[0]->shard_state_free = free_shard_state;
    }
    struct HalfBuilt {  // a set-up that failed half way is not kept for the next call
        ShardState *state;
        epik_amd_placer *owner;
        ~HalfBuilt()
        {
            if (!state->ready) {
                owner->shard_state = nullptr;
                delete state;
            }
        }
    } half_built{state, shards[0]};
    std::vector<ShardSide> &sides = state->sides;
    Quiesce quiesce{sides};
    for (uint32_t g = 0; g < G; ++g) {
        ShardSide &s = sides[g];
        if (state->ready) {  // (the batch itself is new every call)
            SHARD_TRY(hipSetDevice(s.device));
            SHARD_TRY(s.seqs.reserve((size_t)seq_offsets[n] + 64));
            SHARD_TRY(s.offsets.reserve((size_t)(n + 1) * sizeof(uint64_t)));
            if (seq_offsets[n]) SHARD_TRY(hipMemcpyAsync(s.seqs.p, seqs, (size_t)seq_offsets[n], hipMemcpyHostToDevice, s.compute));
            SHARD_TRY(hipMemcpyAsync(s.offsets.p, seq_offsets, (size_t)(n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s.compute));
            continue;
        }
        s.h = shards[g];
        s.device = shards[g]->device;
        s.compute = shards[g]->stream;
        s.copy = shards[g]->stream_in;
        DevBuf *all[] = {&s.seqs, &s.offsets};
        for (DevBuf *b : all) b->device = s.device;
        for (int b = 0; b < 2; ++b) {
            DevBuf *bufs[] = {&s.entries[b], &s.index[b], &s.part_entries[b], &s.amb_slot[b], &s.amb_order[b], &s.amb_avg[b],
                              &s.my_avg[b], &s.my_slot[b], &s.rows[b], &s.n_rows[b], &s.counts[b]};
            for (DevBuf *x : bufs) x->device = s.device;
            for (int q = 0; q < EPIK_AMD_MAX_SHARDS; ++q)
                s.recv_entries[b][q].device = s.recv_index[b][q].device = s.recv_order[b][q].device = s.recv_avg[b][q].device = s.device;
        }
        SHARD_TRY(hipSetDevice(s.device));
        for (uint32_t r = 0; r < G; ++r)  // (direct copies over the link where the devices allow it; not fatal otherwise)
            if (shards[r]->device != s.device) {
                (void)hipDeviceEnablePeerAccess(shards[r]->device, 0);
                (void)hipGetLastError();
            }
        for (int b = 0; b < 2; ++b) {
            SHARD_TRY(hipHostMalloc(reinterpret_cast<void **>(&s.h_part[b]), G * sizeof(unsigned long long), hipHostMallocDefault));
            SHARD_TRY(hipEventCreateWithFlags(&s.accumulated[b], hipEventDisableTiming));
            SHARD_TRY(hipEventCreateWithFlags(&s.arrived[b], hipEventDisableTiming));
            SHARD_TRY(hipEventCreateWithFlags(&s.finished[b], hipEventDisableTiming));
        }
        // the whole batch, once, on every device
        SHARD_TRY(s.seqs.reserve((size_t)seq_offsets[n] + 64));
        SHARD_TRY(s.offsets.reserve((size_t)(n + 1) * sizeof(uint64_t)));
        if (seq_offsets[n]) SHARD_TRY(hipMemcpyAsync(s.seqs.p, seqs, (size_t)seq_offsets[n], hipMemcpyHostToDevice, s.compute));
        SHARD_TRY(hipMemcpyAsync(s.offsets.p, seq_offsets, (size_t)(n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s.compute));
    }
    state->ready = true;

    // room for a chunk's entries: the mean postings per k-mer of the shard times the chunk's k-mers, and margin; a
    // chunk that does not fit is accumulated again with what it asked for
    double margin = 1.3;  // (EPIK_AMD_SHARD_MARGIN: tests -- a margin below 1 sends chunks through the overflow round)
    if (const char *e = std::getenv("EPIK_AMD_SHARD_MARGIN")) margin = std::max(0.0, std::atof(e));
    auto entries_wanted = [&](uint32_t g, uint64_t kmers) {
        epik_amd_partial_info info{};
        (void)epik_amd_placer_partial_info(shards[g], &info);
        const double est = (double)kmers * info.postings_per_kmer * margin + 64.0;
        return (uint64_t)std::min<double>(est, 4.0e9);
    };
    struct Chunk {
        uint64_t first = 0, count = 0, per = 0;  // reads; per = reads of a part
        uint64_t amb_per_owner = 0;
        std::vector<int32_t> slot;               // [count]: the read's row among the ambiguous records, -1 none
        std::vector<int32_t> slot_in_part;       // ... counted from its part's first row (what the finisher indexes by)
    };
    auto make_chunk = [&](uint64_t c) {
        Chunk ch;
        ch.first = c * chunk_reads;
        ch.count = std::min(chunk_reads, n - ch.first);
        ch.per = (ch.count + G - 1) / G;
        ch.slot.assign(ch.count, -1);
        ch.slot_in_part.assign(ch.count, -1);
        std::vector<uint64_t> have(G, 0);
        for (uint64_t i = 0; i < ch.count; ++i)
            if (dirty[ch.first + i]) ++have[i / ch.per];
        for (uint64_t v : have) ch.amb_per_owner = std::max(ch.amb_per_owner, v);
        std::fill(have.begin(), have.end(), 0);
        for (uint64_t i = 0; i < ch.count; ++i)
            if (dirty[ch.first + i]) {
                const uint64_t r = i / ch.per;
                ch.slot_in_part[i] = (int32_t)have[r];
                ch.slot[i] = (int32_t)(r * ch.amb_per_owner + have[r]++);
            }
        return ch;
    };

    auto accumulate = [&](const Chunk &ch, int b, uint64_t min_entries) -> int {
        uint64_t kmers = 0;
        for (uint64_t i = ch.first; i < ch.first + ch.count; ++i) {
            const uint64_t len = seq_offsets[i + 1] - seq_offsets[i];
            kmers += len >= k ? len - k + 1 : 0;
        }
        for (uint32_t g = 0; g < G; ++g) {
            ShardSide &s = sides[g];
            SHARD_TRY(hipSetDevice(s.device));
            const uint64_t cap = std::min<uint64_t>(std::max(entries_wanted(g, kmers), min_entries), 0xfffffff0ull);
            // (a buffer still in use by the copies of two chunks ago: they have been waited for in complete())
            SHARD_TRY(s.entries[b].reserve((size_t)cap * entry_bytes));
            s.entries_cap[b] = cap;
            SHARD_TRY(s.index[b].reserve((size_t)ch.per * G * S * 8u));
            SHARD_TRY(s.part_entries[b].reserve(G * sizeof(unsigned long long)));
            SHARD_TRY(hipMemsetAsync(s.index[b].p, 0, (size_t)ch.per * G * S * 8u, s.compute));
            void *d_slot = nullptr, *d_order = nullptr, *d_avg = nullptr;
            if (ch.amb_per_owner) {
                const size_t cells = (size_t)ch.amb_per_owner * G * N;
                SHARD_TRY(s.amb_slot[b].reserve(ch.count * sizeof(int32_t)));
                SHARD_TRY(s.amb_order[b].reserve(cells * 4u));
                SHARD_TRY(s.amb_avg[b].reserve(cells * 4u));
                SHARD_TRY(s.my_slot[b].reserve(ch.count * sizeof(int32_t)));
                if (s.h_slots_cap[b] < 2 * ch.count) {
                    if (s.h_slots[b]) (void)hipHostFree(s.h_slots[b]);
                    s.h_slots[b] = nullptr, s.h_slots_cap[b] = 0;
                    SHARD_TRY(hipHostMalloc(reinterpret_cast<void **>(&s.h_slots[b]), 2 * ch.count * sizeof(int32_t), hipHostMallocDefault));
                    s.h_slots_cap[b] = 2 * ch.count;
                }
                // (set b's copies of two chunks ago have been waited for in complete(), like its device buffers)
                std::memcpy(s.h_slots[b], ch.slot.data(), ch.count * sizeof(int32_t));
                std::memcpy(s.h_slots[b] + ch.count, ch.slot_in_part.data(), ch.count * sizeof(int32_t));
                SHARD_TRY(hipMemcpyAsync(s.amb_slot[b].p, s.h_slots[b], ch.count * sizeof(int32_t), hipMemcpyHostToDevice, s.compute));
                SHARD_TRY(hipMemcpyAsync(s.my_slot[b].p, s.h_slots[b] + ch.count, ch.count * sizeof(int32_t), hipMemcpyHostToDevice, s.compute));
                SHARD_TRY(hipMemsetAsync(s.amb_order[b].p, 0xff, cells * 4u, s.compute));
                SHARD_TRY(hipMemsetAsync(s.amb_avg[b].p, 0, cells * 4u, s.compute));
                d_slot = s.amb_slot[b].p, d_order = s.amb_order[b].p, d_avg = s.amb_avg[b].p;
            }
            if (const int rc = epik_amd_placer_accumulate_lists_device(
                    s.h, s.seqs.p, s.offsets.as<uint64_t>() + ch.first, ch.count, G, s.entries[b].p, cap, s.index[b].p,
                    s.part_entries[b].p, d_slot, d_order, d_avg, s.compute);
                rc != EPIK_AMD_OK)
                return rc;
            SHARD_TRY(hipMemcpyAsync(s.h_part[b], s.part_entries[b].p, G * sizeof(unsigned long long), hipMemcpyDeviceToHost, s.compute));
            SHARD_TRY(hipEventRecord(s.accumulated[b], s.compute));
        }
        return EPIK_AMD_OK;
    };

    // exchange + finish of a chunk whose accumulate has been enqueued in set b
    auto complete = [&](Chunk &ch, int b) -> int {
        for (;;) {  // (until every shard's lists fit: once more at most, with what they asked for)
            uint64_t worst = 0;
            for (uint32_t g = 0; g < G; ++g) {
                SHARD_TRY(hipSetDevice(sides[g].device));
                SHARD_TRY(hipEventSynchronize(sides[g].accumulated[b]));
                uint64_t total = 0;
                for (uint32_t r = 0; r < G; ++r) total += sides[g].h_part[b][r];
                if (total > sides[g].entries_cap[b]) worst = std::max(worst, total);
            }
            if (worst == 0) break;
            if (worst >= 0xfffffff0ull) return fail_with(EPIK_AMD_ERR_UNSUPPORTED, "a chunk's partial lists exceed 2^32 entries: smaller chunks (EPIK_AMD_SHARD_CHUNK)");
            if (const int rc = accumulate(ch, b, worst + worst / 16); rc != EPIK_AMD_OK) return rc;
        }
        for (uint32_t r = 0; r < G; ++r) {
            ShardSide &f = sides[r];
            const uint64_t begin = std::min(ch.count, r * ch.per), end = std::min(ch.count, (r + 1) * ch.per);
            const uint64_t m = end - begin;
            if (m == 0) continue;
            SHARD_TRY(hipSetDevice(f.device));
            const void *entries[EPIK_AMD_MAX_SHARDS], *index[EPIK_AMD_MAX_SHARDS];
            AmbSources amb{};
            amb.n_shards = G;
            const size_t amb_cells = (size_t)ch.amb_per_owner * N;
            for (uint32_t g = 0; g < G; ++g) {
                ShardSide &src = sides[g];
                uint64_t first = 0;
                for (uint32_t q = 0; q < r; ++q) first += src.h_part[b][q];
                const size_t bytes = (size_t)src.h_part[b][r] * entry_bytes, index_bytes = (size_t)m * S * 8u;
                const uint8_t *src_entries = src.entries[b].as<uint8_t>() + first * entry_bytes;
                const uint8_t *src_index = src.index[b].as<uint8_t>() + (size_t)r * ch.per * S * 8u;
                SHARD_TRY(hipStreamWaitEvent(f.copy, src.accumulated[b], 0));
                if (g == r) {  // its own part is where it is
                    entries[g] = src_entries, index[g] = src_index;
                } else {
                    SHARD_TRY(f.recv_entries[b][g].reserve(bytes + 16));
                    SHARD_TRY(f.recv_index[b][g].reserve(index_bytes));
                    SHARD_TRY(peer_copy(f.recv_entries[b][g].p, f.device, src_entries, src.device, bytes, f.copy));
                    SHARD_TRY(peer_copy(f.recv_index[b][g].p, f.device, src_index, src.device, index_bytes, f.copy));
                    entries[g] = f.recv_entries[b][g].p, index[g] = f.recv_index[b][g].p;
                }
                if (ch.amb_per_owner) {  // the records of this finisher's slots
                    const uint8_t *o = src.amb_order[b].as<uint8_t>() + (size_t)r * amb_cells * 4u;
                    const uint8_t *a = src.amb_avg[b].as<uint8_t>() + (size_t)r * amb_cells * 4u;
                    if (g == r) {
                        amb.order[g] = reinterpret_cast<const uint32_t *>(o), amb.avg[g] = reinterpret_cast<const float *>(a);
                    } else {
                        SHARD_TRY(f.recv_order[b][g].reserve(amb_cells * 4u));
                        SHARD_TRY(f.recv_avg[b][g].reserve(amb_cells * 4u));
                        SHARD_TRY(peer_copy(f.recv_order[b][g].p, f.device, o, src.device, amb_cells * 4u, f.copy));
                        SHARD_TRY(peer_copy(f.recv_avg[b][g].p, f.device, a, src.device, amb_cells * 4u, f.copy));
                        amb.order[g] = f.recv_order[b][g].as<uint32_t>(), amb.avg[g] = f.recv_avg[b][g].as<float>();
                    }
                }
            }
            SHARD_TRY(hipEventRecord(f.arrived[b], f.copy));
            SHARD_TRY(hipStreamWaitEvent(f.compute, f.arrived[b], 0));
            void *d_slot = nullptr, *d_avg = nullptr;
            if (ch.amb_per_owner) {
                SHARD_TRY(f.my_avg[b].reserve(amb_cells * 4u));
                hipLaunchKernelGGL(combine_amb_kernel, dim3((unsigned)((amb_cells + 255) / 256)), dim3(256), 0, f.compute, amb,
                                   (uint64_t)amb_cells, f.my_avg[b].as<float>());
                SHARD_TRY(hipGetLastError());
                d_slot = f.my_slot[b].as<int32_t>() + begin, d_avg = f.my_avg[b].p;  // (uploaded with the chunk, accumulate())
            }
            SHARD_TRY(f.rows[b].reserve(m * keep * sizeof(epik_amd_placement)));
            SHARD_TRY(f.n_rows[b].reserve(m * sizeof(uint32_t)));
            SHARD_TRY(f.counts[b].reserve(m * keep * sizeof(uint32_t)));
            SHARD_TRY(hipMemsetAsync(f.rows[b].p, 0, m * keep * sizeof(epik_amd_placement), f.compute));
            SHARD_TRY(hipMemsetAsync(f.counts[b].p, 0, m * keep * sizeof(uint32_t), f.compute));
            if (const int rc = epik_amd_placer_finish_lists_device(
                    f.h, f.offsets.as<uint64_t>() + ch.first + begin, m, G, entries, index, d_slot, d_avg, f.rows[b].p,
                    f.n_rows[b].p, f.counts[b].p, f.compute);
                rc != EPIK_AMD_OK)
                return rc;
            SHARD_TRY(f.h_rows[b].reserve(m * keep * sizeof(epik_amd_placement)));
            SHARD_TRY(f.h_n_rows[b].reserve(m * sizeof(uint32_t)));
            SHARD_TRY(f.h_counts[b].reserve(m * keep * sizeof(uint32_t)));
            SHARD_TRY(hipMemcpyAsync(f.h_rows[b].p, f.rows[b].p, m * keep * sizeof(epik_amd_placement), hipMemcpyDeviceToHost, f.compute));
            SHARD_TRY(hipMemcpyAsync(f.h_n_rows[b].p, f.n_rows[b].p, m * sizeof(uint32_t), hipMemcpyDeviceToHost, f.compute));
            if (kmer_counts)
                SHARD_TRY(hipMemcpyAsync(f.h_counts[b].p, f.counts[b].p, m * keep * sizeof(uint32_t), hipMemcpyDeviceToHost, f.compute));
            f.out_at[b] = ch.first + begin;
            f.out_reads[b] = m;
            SHARD_TRY(hipEventRecord(f.finished[b], f.compute));
        }
        return EPIK_AMD_OK;
    };
    // set b may be written again once everything that read it has run: the finishers' kernels (they read the
    // sources' entries in place or through copies that they waited for)
    auto wait_set = [&](int b) -> int {
        for (uint32_t r = 0; r < G; ++r) {
            ShardSide &f = sides[r];
            SHARD_TRY(hipSetDevice(f.device));
            SHARD_TRY(hipEventSynchronize(f.finished[b]));
            if (const uint64_t m = f.out_reads[b]) {  // its rows are home: into the caller's arrays
                const uint64_t at = f.out_at[b];
                std::memcpy(rows + at * keep, f.h_rows[b].p, m * keep * sizeof(epik_amd_placement));
                std::memcpy(n_rows + at, f.h_n_rows[b].p, m * sizeof(uint32_t));
                if (kmer_counts) std::memcpy(kmer_counts + at * keep, f.h_counts[b].p, m * keep * sizeof(uint32_t));
                f.out_reads[b] = 0;
            }
        }
        return EPIK_AMD_OK;
    };

    for (auto &side : sides) side.out_reads[0] = side.out_reads[1] = 0;
    std::vector<Chunk> chunk(2);
    bool used[2] = {false, false};
    for (uint64_t c = 0; c < n_chunks; ++c) {
        const int b = (int)(c & 1);
        if (used[b])
            if (const int rc = wait_set(b); rc != EPIK_AMD_OK) return rc;
        chunk[b] = make_chunk(c);
        if (const int rc = accumulate(chunk[b], b, 0); rc != EPIK_AMD_OK) return rc;  // chunk c accumulates ...
        used[b] = true;
        if (c > 0)
            if (const int rc = complete(chunk[b ^ 1], b ^ 1); rc != EPIK_AMD_OK) return rc;  // ... while chunk c - 1 crosses and finishes
    }
    if (const int rc = complete(chunk[(n_chunks - 1) & 1], (int)((n_chunks - 1) & 1)); rc != EPIK_AMD_OK) return rc;
    for (int b = 0; b < 2; ++b)
        if (used[b])
            if (const int rc = wait_set(b); rc != EPIK_AMD_OK) return rc;
    for (uint32_t g = 0; g < G; ++g) {
        SHARD_TRY(hipSetDevice(sides[g].device));
        SHARD_TRY(hipStreamSynchronize(sides[g].copy));
        SHARD_TRY(hipStreamSynchronize(sides[g].compute));
    }
    return EPIK_AMD_OK;
}

}  // namespace

extern "C" int epik_amd_placer_place_sharded(epik_amd_placer *const *shards, uint32_t n_shards, const char *seqs,
                                             const uint64_t *seq_offsets, uint64_t n, epik_amd_placement *rows,
                                             uint32_t *n_rows, uint32_t *kmer_counts)
{
    try {  // std::vector: nothing may leave through the C ABI
        return place_sharded_impl(shards, n_shards, seqs, seq_offsets, n, rows, n_rows, kmer_counts);
    } catch (const std::exception &e) {
        return epik_amd::fail_with(EPIK_AMD_ERR_INVALID, std::string("place_sharded: ") + e.what());
    }
}
