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

#include "db_image.hpp"
#include "epik_amd.h"
#include "place_kernel.h"
#include "placer_impl.hpp"

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
// Reads a wave (a workgroup of the streaming kernel: items per wave) places one after the other when a batch has
// more reads than the device holds waves: sets the grid (launch()).
constexpr uint64_t kGridRounds = 24, kMinGridRounds = 8, kMinReadsPerWave = 2, kMinReadsPerStreamBlock = 16;
// ... the grid for `units` work items (reads per wave, reads per workgroup) of a kernel of which `resident`
// workgroups fit the device, `min_per_block` items a workgroup at least: whole rounds of the resident workgroups,
// and the resident ones alone, striding, where that makes fewer than kMinGridRounds (what the last workgroups
// leave idle is up to a round)
inline uint64_t spread_grid(uint64_t units, uint64_t resident, uint64_t min_per_block)
{
    if (units <= resident) return units;
    const uint64_t rounds = std::min<uint64_t>(kGridRounds, units / (resident * min_per_block));
    return rounds >= kMinGridRounds ? resident * rounds : resident;
}
[[maybe_unused]] constexpr size_t kDbgWords = 64 + 4096 * 64;        // diagnostic builds: phase sums + a row of 8 per wave of 4096 workgroups

}  // namespace

namespace epik_amd {
int fail_with(int code, const std::string &msg) { return fail(code, msg); }
}  // namespace epik_amd

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
    if (p->shard_state && p->shard_state_free) p->shard_state_free(p->shard_state);  // (before the streams go)
    p->shard_state = nullptr;
    (void)hipSetDevice(p->device);
#ifdef EPIK_AMD_ABLATION
    if (p->params.dbg) {  // diagnostic build: where the waves spent their cycles, by phase
        unsigned long long t[64] = {0};
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(t, p->params.dbg, sizeof t, hipMemcpyDeviceToHost);
        if (p->team) {
            const char *classic[8] = {"encode+lookup", "wait-tiles", "descriptors", "wait-desc", "stream", "wait-streams",
                                      "epilogue", "wait-epilogues(+merge)"};
            const char *streaming[8] = {"descriptors", "stream", "ambiguous", "epilogue", "wait-slices", "merge", "-", "-"};
            const char **names = p->team_front ? streaming : classic;
            double all = 0;
            for (int w = 0; w < p->team_waves; ++w)
                for (int i = 0; i < 8; ++i) all += (double)t[w * 8 + i];
            for (int w = 0; w < p->team_waves; ++w) {
                std::fprintf(stderr, "team wave %d:", w);
                for (int i = 0; i < 8; ++i)
                    std::fprintf(stderr, "  %s %.1f%%", names[i], 100.0 * t[w * 8 + i] * p->team_waves / all);
                std::fprintf(stderr, "\n");
            }
            if (p->team_front) {  // the timeline of one wave: code and cycles since the previous entry
                std::vector<unsigned long long> tr(200000);
                (void)hipMemcpy(tr.data(), p->params.dbg + 64, tr.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                if (const char *path = std::getenv("EPIK_AMD_TRACE_FILE")) {
                    if (FILE *f = std::fopen(path, "w")) {
                        unsigned long long last = tr[1];  // (entries a call did not reach stay 0)
                        for (size_t i = 1; i < 100000; ++i) {
                            if (tr[2 * i + 1] == 0) continue;
                            std::fprintf(f, "%llu %llu\n", tr[2 * i], tr[2 * i + 1] - last);
                            last = tr[2 * i + 1];
                        }
                        std::fclose(f);
                    }
                }
            }
            (void)hipFree(p->params.dbg);
            p->params.dbg = nullptr;
        }
    }
    if (p->params.dbg) {
        unsigned long long t[8] = {0};
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
    (void)hipFree(p->d_front_hdr);
    (void)hipFree(p->d_finish_hdr);
    (void)hipFree(p->d_slow_list);
    (void)hipFree(p->d_slice_rows);
    (void)hipFree(p->d_slice_sums);
    (void)hipFree(p->d_front_pool);
    (void)hipFree(p->d_front_cursor);
    (void)hipFree(p->d_sparse_cap);
    (void)hipFree(p->d_scan_tiles);
    if (p->h_front_cursor) (void)hipHostFree(p->h_front_cursor);
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

}  // extern "C" (create_impl below is C++)

namespace {

// image::Sink into device memory: bytes go to d_base, front to back, through two pinned staging
// buffers (the copy of one overlaps the filling of the other).  reserve() cannot fail towards the
// builder -- after an error it keeps handing out staging memory and finish() reports the error.
class DeviceSink final : public epik_amd::image::Sink {
public:
    // staging buffer size; EPIK_AMD_STAGE_BYTES (>= 4096) shrinks it so that tests reach the
    // buffer switches and the larger-than-a-buffer path with small databases
    static constexpr size_t kStageMax = 16u << 20;
    DeviceSink()
    {
        if (const char *e = std::getenv("EPIK_AMD_STAGE_BYTES")) {
            const unsigned long long v = std::strtoull(e, nullptr, 10);
            if (v >= 4096 && v <= kStageMax) _stage_bytes = static_cast<size_t>(v);
        }
    }
    ~DeviceSink() override
    {
        for (int i = 0; i < 2; ++i) {
            if (_stage[i]) (void)hipHostFree(_stage[i]);
            if (_event[i]) (void)hipEventDestroy(_event[i]);
        }
    }
    hipError_t init(uint8_t *d_base, uint64_t total, hipStream_t stream)
    {
        _base = d_base;
        _total = total;
        _stream = stream;
        for (int i = 0; i < 2; ++i) {
            hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&_stage[i]), _stage_bytes, hipHostMallocDefault);
            if (e != hipSuccess) return e;
            e = hipEventCreateWithFlags(&_event[i], hipEventDisableTiming);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    uint8_t *reserve(size_t n) override
    {
        flush_large();
        if (n > _stage_bytes) {  // a single posting list of millions of branches: through a buffer of its own,
            flush();       // copied synchronously when the next piece is asked for
            _large.assign(n, 0);  // (may throw std::bad_alloc: create() catches at the boundary)
            return _large.data();
        }
        if (_fill + n > _stage_bytes) flush();
        uint8_t *p = _stage[_cur] + _fill;
        std::memset(p, 0, n);
        _fill += n;
        return p;
    }
    // everything reserved so far is on its way; returns the first error, checks the size
    hipError_t finish(bool *size_ok)
    {
        flush_large();
        flush();
        const hipError_t e = hipStreamSynchronize(_stream);
        if (_error == hipSuccess) _error = e;
        *size_ok = !_overflow && _done == _total;
        return _error;
    }

private:
    void flush_large()
    {
        if (_large.empty()) return;
        if (_done + _large.size() > _total) {
            _overflow = true;
        } else if (_error == hipSuccess) {
            _error = hipStreamSynchronize(_stream);  // keep the order of the bytes: everything before it has left
            if (_error == hipSuccess)
                _error = hipMemcpy(_base + _done, _large.data(), _large.size(), hipMemcpyHostToDevice);
        }
        _done += _large.size();
        std::vector<uint8_t>().swap(_large);
    }
    void flush()
    {
        if (_fill == 0) return;
        if (_done + _fill > _total) {
            _overflow = true;  // plan and build disagree: never write past the allocation
        } else if (_error == hipSuccess) {
            _error = hipMemcpyAsync(_base + _done, _stage[_cur], _fill, hipMemcpyHostToDevice, _stream);
            if (_error == hipSuccess) _error = hipEventRecord(_event[_cur], _stream);
            _used[_cur] = true;
        }
        _done += _fill;
        _fill = 0;
        _cur ^= 1;
        if (_used[_cur] && _error == hipSuccess) _error = hipEventSynchronize(_event[_cur]);  // its last copy has left
    }
    size_t _stage_bytes = kStageMax;
    std::vector<uint8_t> _large;  // a piece larger than a staging buffer, waiting to be copied
    uint8_t *_base = nullptr, *_stage[2] = {nullptr, nullptr};
    hipEvent_t _event[2] = {nullptr, nullptr};
    bool _used[2] = {false, false};
    hipStream_t _stream = nullptr;
    uint64_t _total = 0, _done = 0;
    size_t _fill = 0;
    int _cur = 0;
    bool _overflow = false;
    hipError_t _error = hipSuccess;
};

int create_impl(const epik_amd_placer_desc *d, uint32_t shard_index, uint32_t shard_count, epik_amd_placer **out)
{
    namespace image = epik_amd::image;
    if (!d || !out) return fail(EPIK_AMD_ERR_INVALID, "null argument");
    *out = nullptr;
    std::string err;
    // everything that needs no device first: argument checks, then the lists themselves
    if (const int rc = image::resolve_shard(d, shard_index, shard_count, err); rc != EPIK_AMD_OK) return fail(rc, err);
    if (const int rc = image::validate(d, shard_index, shard_count, err); rc != EPIK_AMD_OK) return fail(rc, err);

    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(EPIK_AMD_ERR_NO_DEVICE, "no HIP device available (libepik_amd has no CPU fallback)");
    if (d->device < 0 || d->device >= n_dev) return fail(EPIK_AMD_ERR_INVALID, "device ordinal out of range");
    HIP_TRY(hipSetDevice(d->device));

    epik_amd_placer *p = new (std::nothrow) epik_amd_placer();
    if (!p) return fail(EPIK_AMD_ERR_INVALID, "out of host memory");
    {
        static std::atomic<uint64_t> created{0};
        p->generation = ++created;
    }
    struct guard {  // whatever way this function is left before the end, the placer goes with it
        epik_amd_placer *p;
        ~guard() { epik_amd_placer_destroy(p); }
    } owner{p};
#define CREATE_TRY(expr)                                       \
    do {                                                       \
        const hipError_t e_ = (expr);                          \
        if (e_ != hipSuccess) return fail_hip(e_, #expr);      \
    } while (0)
    p->device = d->device;
    p->num_keys = d->num_keys;
    p->num_entries = d->num_entries;

    // ---- kernel + layout (db_image.cpp), then the image, streamed into HBM -----------------------------
    size_t free_mem = 0, total_mem = 0;
    CREATE_TRY(hipMemGetInfo(&free_mem, &total_mem));
    const image::Source src{d, shard_index, shard_count};
    image::Plan &plan = p->plan;
    if (const int rc = image::make_plan(src, free_mem, std::getenv("EPIK_AMD_LAYOUT"), std::getenv("EPIK_AMD_KERNEL"),
                                        plan, err);
        rc != EPIK_AMD_OK) {
        return fail(rc, err);
    }
    p->layout = plan.layout;
    p->team = plan.layout == epik_amd::DbLayout::kTeam;
    p->db_bytes = plan.posting_bytes;
    CREATE_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    CREATE_TRY(hipStreamCreateWithFlags(&p->stream_in, hipStreamNonBlocking));
    CREATE_TRY(hipStreamCreateWithFlags(&p->stream_out, hipStreamNonBlocking));
    CREATE_TRY(hipMalloc(&p->d_table, plan.table_bytes));
    if (plan.filter_bytes) CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_filter), plan.filter_bytes));
    CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_postings), plan.posting_bytes));
    {
        DeviceSink table, filter, postings;
        CREATE_TRY(table.init(static_cast<uint8_t *>(p->d_table), plan.table_bytes, p->stream_in));
        if (plan.filter_bytes)
            CREATE_TRY(filter.init(reinterpret_cast<uint8_t *>(p->d_filter), plan.filter_bytes, p->stream_out));
        CREATE_TRY(postings.init(p->d_postings, plan.posting_bytes, p->stream));
        const int rc = image::build(src, plan, table, plan.filter_bytes ? &filter : nullptr, postings, err);
        bool ok_t = true, ok_f = true, ok_p = true;
        const hipError_t e_t = table.finish(&ok_t);
        const hipError_t e_f = plan.filter_bytes ? filter.finish(&ok_f) : hipSuccess;
        const hipError_t e_p = postings.finish(&ok_p);
        if (rc != EPIK_AMD_OK) {
            return fail(rc, err);
        }
        CREATE_TRY(e_t);
        CREATE_TRY(e_f);
        CREATE_TRY(e_p);
        if (!ok_t || !ok_f || !ok_p) {
            return fail(EPIK_AMD_ERR_INVALID, "internal: the database image does not have the planned size");
        }
    }
    CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_char_class), 256 * sizeof(uint32_t)));
    CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_total), sizeof(unsigned long long)));
    CREATE_TRY(hipMemcpy(p->d_char_class, d->char_class, 256 * sizeof(uint32_t), hipMemcpyHostToDevice));
    p->h_char_class.assign(d->char_class, d->char_class + 256);
    CREATE_TRY(hipEventCreate(&p->ev_start));
    CREATE_TRY(hipEventCreate(&p->ev_stop));

    epik_amd::PlaceParams &pp = p->params;
    pp.n_pad = plan.n_pad;
    pp.table = p->d_table;
    pp.filter = p->d_filter;
    pp.filter_rec_bytes = plan.filter_rec_bytes;
    pp.postings = p->d_postings;
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
    if (const char *e = std::getenv("EPIK_AMD_GRID_PERCENT")) p->grid_percent = std::strtoull(e, nullptr, 10);
    if (const char *st = std::getenv("EPIK_AMD_STAMPS"); st && st[0] == '1') {
        CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&pp.dbg), kDbgWords * sizeof(unsigned long long)));
        CREATE_TRY(hipMemset(pp.dbg, 0, kDbgWords * sizeof(unsigned long long)));
    }
#endif
    hipDeviceProp_t prop;
    CREATE_TRY(hipGetDeviceProperties(&prop, d->device));
    if (p->team) {
        // ---- team kernel: one workgroup of W waves per read (team_kernel.hip) -------------------------
        p->team_waves = plan.team_waves;
        p->team_table = static_cast<const uint8_t *>(p->d_table);
        p->team_passes = plan.team_passes;
        p->team_slice_rows = plan.team_slice_rows;
        p->team_rows_pad = plan.team_rows_pad;
        const uint32_t desc_bytes = epik_amd::team_desc_bytes(d->keep_at_most);
        // Placing, the front end runs as a kernel of its own (team_stream.hip) unless the tree has more
        // slices than header words fit a wave, or EPIK_AMD_TEAM_FRONT=0 asks for the one-kernel placement.
        const char *front_env = std::getenv("EPIK_AMD_TEAM_FRONT");
        p->team_front = (uint32_t)plan.team_waves * plan.team_passes <= epik_amd::kFrontMaxSlices &&
                        !(front_env && front_env[0] == '0');
        for (int counts = 0; counts < 3; ++counts) {
            auto &g = p->geo[counts];
            g.waves_per_block = (uint32_t)plan.team_waves;
            g.lds_wave_bytes = epik_amd::team_slice_bytes(plan.team_rows_pad, counts);
            g.lds_block_bytes = (uint32_t)epik_amd::team_lds_bytes(plan.team_waves, plan.team_passes, g.lds_wave_bytes,
                                                                   desc_bytes, d->keep_at_most);
            g.max_blocks = 0;
            // the 8-bit kernel keeps one "seen" bit per row in the slice's descriptor list
            if (counts == epik_amd::kCounts8 && (plan.team_rows_pad + 7u) / 8u > desc_bytes) continue;
            uint32_t per_cu = epik_amd::team_resident_blocks(plan.team_waves, g.lds_block_bytes);
            if (per_cu == 0) continue;  // (make_plan made sure the 32-bit counts fit; narrower ones then do too)
            CREATE_TRY(epik_amd::set_team_lds_limit(plan.team_waves, counts, g.lds_block_bytes));
            // the grid is the workgroups resident at once: of the streaming kernel where it places (a grid
            // larger than team_place_kernel holds at once only queues -- that kernel then gets the few reads
            // the front kernel left it, or the launches of a k-mer-space shard, which size their own)
            int by_query = 0;
            CREATE_TRY(epik_amd::team_occupancy(plan.team_waves, counts, g.lds_block_bytes, &by_query));
            per_cu = std::max<uint32_t>(1u, std::min<uint32_t>(per_cu, (uint32_t)std::max(by_query, 1)));
            g.max_blocks = (uint32_t)prop.multiProcessorCount * per_cu;
            g.resident_waves = per_cu * (uint32_t)plan.team_waves;
            if (p->team_front) {
                // the streaming kernel: its own workgroups (4 waves), LDS and registers; the grid a multiple of
                // the workgroups that share a read
                // [0]: the halves of a sharded placement, workgroups of four; [1]: the one-pass placement, of four or two
                g.stream_bw[0] = 4;
                g.stream_bw[1] = epik_amd::stream_block_waves(plan.team_waves, g.lds_wave_bytes, desc_bytes);
                if (const char *e = std::getenv("EPIK_AMD_STREAM_BLOCK")) g.stream_bw[1] = e[0] == '2' ? 2 : e[0] == '4' ? 4 : g.stream_bw[1];
                for (int v = 0; v < 2; ++v) {
                    const int bw = g.stream_bw[v];
                    g.stream_lds_bytes[v] = (uint32_t)epik_amd::stream_lds_bytes(g.lds_wave_bytes, desc_bytes, bw);
                    uint32_t stream_per_cu = epik_amd::stream_resident_blocks(plan.team_waves, g.stream_lds_bytes[v], bw);
                    for (int mode = 0; mode < 5; ++mode) {
                        if ((v == 1) != (mode == epik_amd::kTeamModePlace)) continue;
                        CREATE_TRY(epik_amd::set_team_stream_lds_limit(plan.team_waves, counts, mode, bw, g.stream_lds_bytes[v]));
                    }
                    CREATE_TRY(epik_amd::team_stream_occupancy(plan.team_waves, counts, v == 1 ? epik_amd::kTeamModePlace : epik_amd::kTeamModeAccumulateLists,
                                                               bw, g.stream_lds_bytes[v], &by_query));
                    stream_per_cu = std::max<uint32_t>(1u, std::min<uint32_t>(stream_per_cu, (uint32_t)std::max(by_query, 1)));
                    const uint32_t parts = epik_amd::stream_parts(plan.team_waves, bw);
                    g.stream_blocks[v] = std::max(parts, (uint32_t)prop.multiProcessorCount * stream_per_cu / parts * parts);
                }
            }
        }
        if (p->team_front) {
            CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_front_cursor), 3 * sizeof(unsigned long long)));
            CREATE_TRY(hipHostMalloc(reinterpret_cast<void **>(&p->h_front_cursor), 3 * sizeof(unsigned long long),
                                     hipHostMallocDefault));
            p->h_front_cursor[0] = p->h_front_cursor[1] = p->h_front_cursor[2] = 0;
            if (const char *e = std::getenv("EPIK_AMD_TEAM_POOL")) p->front_pool_forced = std::strtoull(e, nullptr, 10);
            p->sparse_quads = epik_amd::team_sparse_quads(plan.team_rows_pad);
            // (every item is asked: on a database built from reference sequences the item that streams ALL of a read's
            // chunks touches a clade of a few dozen rows)
            p->sparse_chunks = 0xffffffffu;
            if (const char *e = std::getenv("EPIK_AMD_TEAM_SPARSE")) {
                if (std::strcmp(e, "always") == 0)
                    p->sparse_quads = 256u, p->sparse_chunks = 0xffffffffu;
                else
                    p->sparse_chunks = (uint32_t)std::strtoul(e, nullptr, 10), p->sparse_quads = p->sparse_chunks ? p->sparse_quads : 0u;
            }
            // Grids of the front kernel (workgroups of one wave, 32 resident per CU) and of the merge kernel (four
            // waves, 5 resident): several times what the device holds at once -- see kReadsPerWave in launch().
            // (Front: 32 / 64 / 128 per CU = 2.08 / 1.95 / 1.87 ms per million reads; every wave takes the pool
            // in pieces of its own, so more of them cost pool.)
            uint32_t front_per_cu = 128, merge_per_cu = 32;
#ifdef EPIK_AMD_ABLATION
            if (const char *e = std::getenv("EPIK_AMD_FRONT_PER_CU")) front_per_cu = std::atoi(e) ? (uint32_t)std::atoi(e) : front_per_cu;
            if (const char *e = std::getenv("EPIK_AMD_MERGE_PER_CU")) merge_per_cu = std::atoi(e) ? (uint32_t)std::atoi(e) : merge_per_cu;
#endif
            p->front_blocks = (uint32_t)prop.multiProcessorCount * front_per_cu;
            p->merge_blocks = (uint32_t)prop.multiProcessorCount * merge_per_cu;
        }
    } else {
        for (int counts = 0; counts < 3; ++counts) {
            auto &g = p->geo[counts];
            // per wave: float32 scores + 8/16/32-bit counts + the chunk descriptors of one round
            // + one trip of spare entries (the kernel prefetches a trip ahead)
            const uint32_t desc_bytes = epik_amd::kWaveDescBytes;
            g.lds_wave_bytes = epik_amd::wave_lds_bytes(pp.n_pad, counts);
            if (counts == epik_amd::kCounts8 && (pp.n_pad + 7u) / 8u > desc_bytes) {
                g.max_blocks = 0;  // the 8-bit kernel keeps one flag bit per row in the descriptor area: no room
                continue;
            }
            if (g.lds_wave_bytes > kMaxLdsPerBlock) {
                return fail(EPIK_AMD_ERR_UNSUPPORTED, "num_branches too large for the one-wavefront-per-read kernel");
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
                CREATE_TRY(epik_amd::set_place_reads_lds_limit(p->layout, p->plan.runs, counts, block_bytes));
                int per_cu = 0;
                CREATE_TRY(epik_amd::place_reads_occupancy(p->layout, p->plan.runs, counts, (int)(wpb * 64u), block_bytes, &per_cu));
                const uint32_t lds_units = (block_bytes + epik_amd::kLdsGranule - 1u) / epik_amd::kLdsGranule;
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
            CREATE_TRY(epik_amd::set_place_reads_lds_limit(p->layout, p->plan.runs, counts, g.lds_block_bytes));
            CREATE_TRY(epik_amd::set_finish_reads_lds_limit(counts, g.lds_block_bytes));
        }
    }
    // EPIK_AMD_MAX_BLOCKS=<n> (tests): launch as if the device held only so many workgroups -- small batches then
    // take the grids of the large ones
    if (const char *e = std::getenv("EPIK_AMD_MAX_BLOCKS")) p->max_blocks_cap = (uint32_t)std::strtoul(e, nullptr, 10);
    // EPIK_AMD_WIDE_COUNTS=0|1|2: 16-, 32-, 8-bit counts whatever the reads (tests, experiments)
    if (const char *w = std::getenv("EPIK_AMD_WIDE_COUNTS")) {
        p->counts = w[0] == '1' ? epik_amd::kCounts32 : w[0] == '2' ? epik_amd::kCounts8 : epik_amd::kCounts16;
        if (p->counts == epik_amd::kCounts8 && p->geo[epik_amd::kCounts8].max_blocks == 0) p->counts = epik_amd::kCounts16;
        p->counts_forced = true;
    }
#undef CREATE_TRY

    owner.p = nullptr;
    *out = p;
    return EPIK_AMD_OK;
}

}  // namespace

extern "C" {

int epik_amd_placer_create_sharded(const epik_amd_placer_desc *d, uint32_t shard_index, uint32_t shard_count,
                                   epik_amd_placer **out)
{
    try {  // std::string, std::vector: nothing may leave through the C ABI
        return create_impl(d, shard_index, shard_count, out);
    } catch (const std::exception &e) {
        return fail(EPIK_AMD_ERR_INVALID, std::string("create: ") + e.what());
    }
}

// ---- capacity planning and the image itself, on the host (no device) ---------------------------------
namespace {
// image::Sink into a host buffer (or nowhere): epik_amd_placer_build_image
class HostSink final : public epik_amd::image::Sink {
public:
    HostSink(void *dst, uint64_t total) : _dst(static_cast<uint8_t *>(dst)), _total(total) {}
    uint8_t *reserve(size_t n) override
    {
        uint8_t *p;
        if (_dst && _done + n <= _total) {
            p = _dst + _done;
        } else {  // dropped (or, after a size mismatch, kept off the caller's buffer)
            if (_scratch.size() < n) _scratch.resize(n);
            p = _scratch.data();
        }
        std::memset(p, 0, n);
        _done += n;
        return p;
    }
    bool size_ok() const { return _done == _total; }

private:
    uint8_t *_dst;
    uint64_t _total, _done = 0;
    std::vector<uint8_t> _scratch;
};
}  // namespace

static void fill_plan(const epik_amd::image::Plan &plan, epik_amd_plan *out)
{
    *out = epik_amd_plan{};
    out->kernel = plan.layout == epik_amd::DbLayout::kTeam ? 1u : 0u;
    out->layout = (uint32_t)plan.layout;
    out->team_waves = (uint32_t)plan.team_waves;
    out->team_passes = plan.team_passes;
    out->slice_rows = plan.team_slice_rows;
    for (int c = 0; c < 3; ++c) out->resident_waves[c] = plan.wave_resident[c];
    out->table_bytes = plan.table_bytes;
    out->filter_bytes = plan.filter_bytes;
    out->posting_bytes = plan.posting_bytes;
    out->kept_entries = plan.kept_entries;
    out->run_coded = plan.runs ? 1u : 0u;
}

int epik_amd_placer_plan_sizes(uint32_t kmer_size, uint32_t alphabet_size, uint32_t num_branches, uint32_t keep_at_most,
                               const epik_amd_list_bin *bins, uint64_t n_bins, uint32_t shard_index, uint32_t shard_count,
                               uint64_t free_bytes, epik_amd_plan *out)
{
    namespace image = epik_amd::image;
    if (!out) return fail(EPIK_AMD_ERR_INVALID, "null argument");
    try {
        std::string err;
        image::SizeDesc z;
        z.kmer_size = kmer_size, z.alphabet_size = alphabet_size, z.num_branches = num_branches, z.keep_at_most = keep_at_most;
        z.bins = bins, z.n_bins = n_bins, z.shard_index = shard_index, z.shard_count = shard_count;
        image::Plan plan;
        bool bound = false;
        if (const int rc = image::plan_sizes(z, (size_t)free_bytes, std::getenv("EPIK_AMD_LAYOUT"), std::getenv("EPIK_AMD_KERNEL"), plan, bound, err);
            rc != EPIK_AMD_OK)
            return fail(rc, err);
        fill_plan(plan, out);
        out->posting_bytes_is_bound = bound ? 1u : 0u;
        return EPIK_AMD_OK;
    } catch (const std::exception &e) {
        return fail(EPIK_AMD_ERR_INVALID, std::string("plan_sizes: ") + e.what());
    }
}

int epik_amd_placer_plan(const epik_amd_placer_desc *d, uint32_t shard_index, uint32_t shard_count,
                         uint64_t free_bytes, epik_amd_plan *out)
{
    namespace image = epik_amd::image;
    if (!out) return fail(EPIK_AMD_ERR_INVALID, "null argument");
    try {
        std::string err;
        if (const int rc = image::resolve_shard(d, shard_index, shard_count, err); rc != EPIK_AMD_OK) return fail(rc, err);
        if (const int rc = image::validate(d, shard_index, shard_count, err); rc != EPIK_AMD_OK) return fail(rc, err);
        image::Plan plan;
        if (const int rc = image::make_plan(image::Source{d, shard_index, shard_count}, (size_t)free_bytes,
                                            std::getenv("EPIK_AMD_LAYOUT"), std::getenv("EPIK_AMD_KERNEL"), plan, err);
            rc != EPIK_AMD_OK)
            return fail(rc, err);
        fill_plan(plan, out);
        return EPIK_AMD_OK;
    } catch (const std::exception &e) {
        return fail(EPIK_AMD_ERR_INVALID, std::string("plan: ") + e.what());
    }
}

int epik_amd_placer_build_image(const epik_amd_placer_desc *d, uint32_t shard_index, uint32_t shard_count,
                                uint64_t free_bytes, void *table, void *filter, void *postings)
{
    namespace image = epik_amd::image;
    try {
        std::string err;
        if (const int rc = image::resolve_shard(d, shard_index, shard_count, err); rc != EPIK_AMD_OK) return fail(rc, err);
        if (const int rc = image::validate(d, shard_index, shard_count, err); rc != EPIK_AMD_OK) return fail(rc, err);
        image::Plan plan;
        const image::Source src{d, shard_index, shard_count};
        if (const int rc = image::make_plan(src, (size_t)free_bytes, std::getenv("EPIK_AMD_LAYOUT"),
                                            std::getenv("EPIK_AMD_KERNEL"), plan, err);
            rc != EPIK_AMD_OK)
            return fail(rc, err);
        HostSink t(table, plan.table_bytes), f(filter, plan.filter_bytes), ps(postings, plan.posting_bytes);
        if (const int rc = image::build(src, plan, t, plan.filter_bytes ? &f : nullptr, ps, err); rc != EPIK_AMD_OK)
            return fail(rc, err);
        if (!t.size_ok() || !ps.size_ok() || (plan.filter_bytes && !f.size_ok()))
            return fail(EPIK_AMD_ERR_INVALID, "internal: the database image does not have the planned size");
        return EPIK_AMD_OK;
    } catch (const std::exception &e) {
        return fail(EPIK_AMD_ERR_INVALID, std::string("build_image: ") + e.what());
    }
}

// k-mers a read may have with counts of that width (the top bit of 16- and 32-bit counts is a flag)
static uint64_t max_kmers_of_counts(int counts)
{
    return counts == epik_amd::kCounts8 ? 255u : counts == epik_amd::kCounts16 ? 32767u : 0x7fffffffull;
}

// The narrowest counts that hold the k-mers of a read of `longest` characters, 8 bits only when
// that keeps more waves on a CU than 16.
static int counts_for(const epik_amd_placer *p, uint64_t longest)
{
    const uint64_t kmers = longest >= p->params.kmer_size ? longest - p->params.kmer_size + 1 : 0;
    if (kmers >= 32768u) return epik_amd::kCounts32;
    // (what counts where the streaming kernel places: ITS workgroups on a CU, with 8- and with 16-bit counts)
    const auto &g8 = p->geo[epik_amd::kCounts8], &g16 = p->geo[epik_amd::kCounts16];
    const bool more_waves = p->team && p->team_front && g8.max_blocks != 0
                                ? g8.stream_blocks[1] * (uint32_t)g8.stream_bw[1] > g16.stream_blocks[1] * (uint32_t)g16.stream_bw[1]
                                : g8.resident_waves > g16.resident_waves;
    if (kmers <= 255u && more_waves) return epik_amd::kCounts8;
    return epik_amd::kCounts16;
}

// What a launch works on besides the read batch: the partial vectors of a k-mer-space shard
// (accumulate writes them, finish reads them) and the records of the reads' ambiguous k-mers.
struct shard_buffers {
    float *scores = nullptr;
    uint16_t *counts = nullptr;
    const int32_t *amb_slot = nullptr;
    uint32_t *amb_order = nullptr;
    float *amb_avg = nullptr;
    // ... or its partial lists (accumulate writes entries / index / part_total; finish reads `sources`)
    uint8_t *entries = nullptr;
    uint64_t entries_cap = 0;
    uint2 *index = nullptr;
    unsigned long long *part_total = nullptr;
    uint32_t parts = 0;
    const epik_amd::SparseSources *sources = nullptr;
};
enum launch_mode { kPlace = epik_amd::kTeamModePlace, kAccumulate = epik_amd::kTeamModeAccumulate,
                   kFinish = epik_amd::kTeamModeFinish, kAccumulateLists = epik_amd::kTeamModeAccumulateLists,
                   kFinishLists = epik_amd::kTeamModeFinishLists };
inline bool is_finish(launch_mode m) { return m == kFinish || m == kFinishLists; }
inline bool is_accumulate(launch_mode m) { return m == kAccumulate || m == kAccumulateLists; }

// Grid of the front kernel (workgroups of one wave, a read at a time): up to p->front_blocks, four times what the
// device holds -- but a wave takes the pool in pieces of kFrontPoolChunk descriptors, a few dozen reads' worth: with
// fewer reads per wave than that the descriptors of a small batch would lie spread over a pool many times their size.
uint64_t front_grid(const epik_amd_placer *p, uint64_t n)
{
    const uint64_t resident = (uint64_t)p->front_blocks / 4u;
    return std::min<uint64_t>(n, std::max<uint64_t>(resident, std::min<uint64_t>(p->front_blocks, n / 32u)));
}

// Scratch of the front kernel for a launch of n reads (`total_chars`: their characters when the caller knows,
// else 0).  The pool is sized from what the image says a k-mer's descriptors take and from what earlier
// launches asked for; a read that finds it full is placed by team_place_kernel, so the estimate only
// decides speed.  Grown, never shrunk; growing frees the old buffer (which waits for the device).
static int reserve_front(epik_amd_placer *p, uint64_t n, uint64_t total_chars, bool with_pool, bool lists)
{
    const uint32_t slices = (uint32_t)p->team_waves * p->team_passes;
    if (!with_pool) {  // finish: headers of its own, the slices' results; nothing is looked up
        const size_t bytes = (size_t)n * epik_amd::front_hdr_stride(slices);
        if (bytes > p->finish_hdr_bytes) {
            (void)hipFree(p->d_finish_hdr);
            p->d_finish_hdr = nullptr, p->finish_hdr_bytes = 0;
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_finish_hdr), bytes));
            p->finish_hdr_bytes = bytes;
        }
    }
    if (lists && (size_t)n * slices > p->sparse_cap_items) {
        (void)hipFree(p->d_sparse_cap);
        (void)hipFree(p->d_scan_tiles);
        p->d_sparse_cap = nullptr, p->d_scan_tiles = nullptr, p->sparse_cap_items = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_sparse_cap), (size_t)n * slices * sizeof(uint32_t)));
        // (the scan's tile sums: at most a tile per part more than the reads alone make)
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_scan_tiles),
                          (size_t)(epik_amd::sparse_scan_tiles(n, slices) + EPIK_AMD_MAX_SHARDS) * sizeof(unsigned long long)));
        p->sparse_cap_items = (size_t)n * slices;
    }
    const size_t hdr_bytes = with_pool ? (size_t)n * epik_amd::front_hdr_stride(slices) : 0;
    if (hdr_bytes > p->front_hdr_bytes) {
        (void)hipFree(p->d_front_hdr);
        p->d_front_hdr = nullptr, p->front_hdr_bytes = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_front_hdr), hdr_bytes));
        p->front_hdr_bytes = hdr_bytes;
    }
    if (with_pool && n > p->slow_list_reads) {
        (void)hipFree(p->d_slow_list);
        p->d_slow_list = nullptr, p->slow_list_reads = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_slow_list), (size_t)n * sizeof(uint64_t)));
        p->slow_list_reads = n;
    }
    // (the slices' results: placing and finishing; the accumulate of partial lists has none -- and must not free what a
    // finish launch running beside it on another stream is using)
    if (!(with_pool && lists) && n > p->slice_out_reads) {
        (void)hipFree(p->d_slice_rows);
        (void)hipFree(p->d_slice_sums);
        p->d_slice_rows = p->d_slice_sums = nullptr, p->slice_out_reads = 0;
        HIP_TRY(hipMalloc(&p->d_slice_rows, (size_t)n * slices * p->params.keep_at_most * 16u));
        HIP_TRY(hipMalloc(&p->d_slice_sums, (size_t)n * slices * epik_amd::kTeamPartialBytes));
        p->slice_out_reads = n;
    }
    if (!with_pool) return EPIK_AMD_OK;  // (finish: the totals come from the caller, nothing is looked up)
    uint64_t want;
    if (p->front_pool_forced) {
        want = p->front_pool_forced;
    } else {
        // chunks per k-mer that has a list (all slices and passes together), 10 % on top, and the padding
        // of every slice's list to the ring
        const double per_kmer = (double)p->plan.team_chunks / (double)std::max<uint64_t>(p->plan.present_codes, 1);
        const uint64_t chars = total_chars ? total_chars : n * (p->longest_read_hint ? p->longest_read_hint : 160u);
        // ... and what the waves of the front kernel leave unused of the pieces they take the pool in
        const uint64_t front_waves = front_grid(p, n);
        double est = (double)chars * per_kmer * 1.1 + (double)n * slices * epik_amd::kTeamRing +
                     (double)front_waves * epik_amd::kFrontPoolChunk;
        // what the last finished launch asked for per read, if that is more
        if (p->h_front_cursor[2]) {
            const double asked = (double)p->h_front_cursor[0] / (double)p->h_front_cursor[2] * (double)n * 1.15;
            if (asked > est) est = asked;
        }
        want = (uint64_t)est + 1024u;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const uint64_t room = ((uint64_t)free_b + p->front_pool_cap * 8u) / 2u / 8u;  // at most half of what is free
            want = std::min<uint64_t>(want, std::max<uint64_t>(room, 1024u));
        }
    }
    want = (want + 7u) & ~7ull;
    if (want > p->front_pool_cap || (p->front_pool_forced && want != p->front_pool_cap)) {
        (void)hipFree(p->d_front_pool);
        p->d_front_pool = nullptr, p->front_pool_cap = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_front_pool), (size_t)want * sizeof(uint64_t)));
        p->front_pool_cap = want;
    }
    return EPIK_AMD_OK;
}

static int launch(epik_amd_placer *p, launch_mode mode, const void *d_seqs, const void *d_seq_offsets, uint64_t n,
                  void *d_rows, void *d_n_rows, void *d_counts, hipStream_t stream, const shard_buffers &shard = {},
                  uint64_t total_chars = 0)
{
    if (n == 0) return EPIK_AMD_OK;
    const bool lists = mode == kAccumulateLists || mode == kFinishLists;
    epik_amd::PlaceParams pp = p->params;
    pp.partial_scores = shard.scores;  // non-null in the placement kernels: accumulate only
    pp.partial_counts = shard.counts;
    // (the dense partial counts are uint16: a read of more k-mers is marked, not wrapped, whatever the LDS counts hold)
    pp.max_kmers_cap = (mode == kAccumulate || mode == kFinish) ? 65535u : 0u;
    pp.amb_slot = shard.amb_slot;
    pp.amb_order = shard.amb_order;
    pp.amb_avg = shard.amb_avg;
    pp.seqs = static_cast<const uint8_t *>(d_seqs);
    pp.seq_offsets = static_cast<const uint64_t *>(d_seq_offsets);
    pp.n_reads = n;
    pp.rows = static_cast<epik_amd_placement *>(d_rows);
    pp.n_rows = static_cast<uint32_t *>(d_n_rows);
    pp.kmer_counts = static_cast<uint32_t *>(d_counts);
#ifdef EPIK_AMD_ABLATION
    // (the timeline of one wave holds the LAST launch's: EPIK_AMD_TRACE_HALF=accumulate | finish keeps the other half out)
    if (const char *half = std::getenv("EPIK_AMD_TRACE_HALF"))
        if ((std::strcmp(half, "accumulate") == 0 && is_finish(mode)) || (std::strcmp(half, "finish") == 0 && is_accumulate(mode))) pp.dbg = nullptr;
#endif
    const auto &g = p->geo[p->counts];
    if (g.max_blocks == 0) return fail(EPIK_AMD_ERR_UNSUPPORTED, "no kernel of this count width for this tree size");
    if (lists && !(p->team && p->team_front))
        return fail(EPIK_AMD_ERR_UNSUPPORTED, "partial lists need the front / streaming / merge kernels of a large tree "
                                              "(epik_amd_placer_partial_info says so: use the dense calls)");
    pp.lds_wave_bytes = g.lds_wave_bytes;
    // the team kernel places one read per workgroup, the others one per wave
    uint64_t blocks = p->team ? n : (n + g.waves_per_block - 1) / g.waves_per_block;
    const uint64_t resident = p->max_blocks_cap ? std::min<uint64_t>(g.max_blocks, p->max_blocks_cap) : g.max_blocks;
    if (blocks > resident) {
        // More reads than waves the device holds: every wave places several, one after the other.  NOT as few
        // workgroups as are resident, each striding through the whole batch: the CUs do not progress alike,
        // and the launch ends with its slowest wave.  A grid of kGridRounds times the resident workgroups lets
        // the dispatcher hand the next workgroup to whichever CU has room, and what the last ones leave idle is
        // a 24th of the launch (configs[1]: resident x 1 / 4 / 16 / 32 / 64 / one read per wave = 6.97 / 6.75 /
        // 6.55 / 6.55 / 6.67 / 7.97 ms of the diagnostic build; a workgroup's start costs its share of clearing
        // the LDS, hence a minimum of reads per wave for the small batches).
        // (Tried instead: the resident grid, its waves taking the reads four at a time from a device-wide counter,
        // the question for the next four asked just before a read streams.  N = 1 303 / 1 999: +1.5 / +3 % over
        // this grid; configs[1]: -1.5 % (two more registers in a kernel held to 96, spilled); two reads at a time
        // or one: the counter itself is the limit, 13 ns per question.  Not kept.)
        blocks = p->team ? resident : spread_grid(blocks, resident, kMinReadsPerWave);
#ifdef EPIK_AMD_ABLATION
        // (timing experiments: a given multiple of the resident workgroups)
        if (p->grid_percent) blocks = std::max<uint64_t>(1, resident * p->grid_percent / 100u);
#endif
    }
    p->last_blocks = (uint32_t)blocks;
    p->last_streamed = false;
    p->last_geo = (uint32_t)p->counts;
    const bool timed = p->timing && !is_finish(mode);
    if (timed) HIP_TRY(hipEventRecord(p->ev_start, stream));
    if (p->team) {
        epik_amd::TeamParams tp{};
        tp.base = pp;
        tp.team_table = p->team_table;
        tp.num_keys = p->plan.table_keys;
        tp.shard_index = p->plan.shard_index, tp.shard_count = p->plan.shard_count;
        tp.team_paired = p->plan.team_paired ? 1u : 0u;
        tp.passes = p->team_passes;
        tp.slice_rows = p->team_slice_rows;
        tp.rows_pad = p->team_rows_pad;
        tp.desc_cap = epik_amd::kTeamDescCap;
        tp.slice_bytes = g.lds_wave_bytes;
        tp.desc_bytes = epik_amd::team_desc_bytes(pp.keep_at_most);
        // (the streaming kernel numbers a read's slices in 32 bits)
        // No room for the scratch of the three-kernel placement: the one-kernel placement needs none.  Remembered
        // on the handle -- a launch of as many reads or more does not try again (every attempt is a hipFree and a
        // hipMalloc, i.e. a wait for the device) -- and told by epik_amd_placer_last_path().
        bool streamed = p->team_front && n * ((uint64_t)p->team_waves * p->team_passes) < (1ull << 32) &&
                        !(p->front_failed_reads && n >= p->front_failed_reads);
        if (streamed && reserve_front(p, n, total_chars, !is_finish(mode), mode == kAccumulateLists) != EPIK_AMD_OK) {
            (void)hipGetLastError();
            p->front_failed_reads = n;
            streamed = false;
        }
        if (lists && !streamed)
            return fail(EPIK_AMD_ERR_HIP, "partial lists: no device memory for the scratch of a launch of this size");
        if (streamed) {
            // Placing: front kernel (a wave per read), streaming kernel (a wave per slice of a read), merge
            // kernel (a wave per read), and team_place_kernel for the reads whose descriptors found the pool
            // full.  The halves of a k-mer-space-sharded placement: accumulate = front (+ scan: partial lists) +
            // streaming (+ the other kernel for the rest), finish = headers + streaming + merge.
            tp.front_hdr = is_finish(mode) ? p->d_finish_hdr : p->d_front_hdr;
            tp.front_hdr_stride = epik_amd::front_hdr_stride((uint32_t)p->team_waves * p->team_passes);
            tp.front_pool = p->d_front_pool;
            tp.front_pool_cap = p->front_pool_cap;
            tp.front_cursor = p->d_front_cursor;
            tp.slow_list = p->d_slow_list;
            tp.slice_rows_out = p->d_slice_rows;
            tp.slice_sums_out = p->d_slice_sums;
            tp.sparse_chunks = p->sparse_quads ? p->sparse_chunks : 0u;
            tp.sparse_quads = p->sparse_quads;
            if (mode == kAccumulateLists) {
                tp.sparse_cap = p->d_sparse_cap;
                tp.sparse_index = shard.index;
                tp.sparse_entries = shard.entries;
                tp.sparse_entries_cap = shard.entries_cap;
                tp.sparse_part_total = shard.part_total;
                tp.sparse_parts = shard.parts;
                tp.sparse_part_reads = (uint32_t)((n + shard.parts - 1) / shard.parts);
            }
            const uint64_t front_blocks = front_grid(p, n);
            if (is_finish(mode)) {
                HIP_TRY(epik_amd::launch_team_headers(tp, p->team_waves, p->counts, stream));
            } else {
                HIP_TRY(hipMemsetAsync(p->d_front_cursor, 0, 2 * sizeof(unsigned long long), stream));
                HIP_TRY(epik_amd::launch_team_front(tp, p->team_waves, p->counts, mode == kAccumulateLists,
                                                    dim3((unsigned)front_blocks), stream));
                if (mode == kAccumulateLists) HIP_TRY(epik_amd::launch_team_sparse_scan(tp, p->team_waves, p->d_scan_tiles, stream));
            }
            // (workgroups of bw = four or two waves: W / bw of them share a read, or -- two slices per pass, four waves -- one holds two reads)
            const int sv = mode == kPlace ? 1 : 0, bw = g.stream_bw[sv];
            const uint32_t parts = epik_amd::stream_parts(p->team_waves, bw), per_block = epik_amd::stream_reads_per_block(p->team_waves, bw);
            const uint64_t n_units = (n + per_block - 1) / per_block;  // reads, or pairs of them
            uint64_t stream_blocks = n_units * parts;
            const uint64_t stream_resident = p->max_blocks_cap ? std::min<uint64_t>(g.stream_blocks[sv], (uint64_t)p->max_blocks_cap * parts) : g.stream_blocks[sv];
            // (the dense halves are bound by the partial vectors in HBM: 65 536 reads per step, 16.4 M reads/s on the
            // resident grid against 15.1 spread)
            if (stream_blocks > stream_resident && (mode == kAccumulate || mode == kFinish)) stream_blocks = stream_resident;
            if (stream_blocks > stream_resident) {  // as for `blocks` above (N = 9 999: resident x 1 / 4 / 16 / 32 / 64 = 20.2 / 19.9 / 19.4 / 19.4 / 19.6 ms)
                stream_blocks = spread_grid(n_units, stream_resident / parts, kMinReadsPerStreamBlock) * parts;
#ifdef EPIK_AMD_ABLATION
                if (p->grid_percent) stream_blocks = std::max<uint64_t>(parts, stream_resident * p->grid_percent / 100u / parts * parts);
#endif
            }
            p->last_blocks = (uint32_t)stream_blocks;
            p->last_streamed = true;
            p->geo[p->counts].last_stream = (uint32_t)sv;
            HIP_TRY(epik_amd::launch_team_stream(tp, p->team_waves, p->counts, (int)mode, bw, dim3((unsigned)stream_blocks),
                                                 g.stream_lds_bytes[sv], stream, shard.sources));
            if (!is_accumulate(mode)) {
                const uint64_t merge_blocks = std::min<uint64_t>((n + 3u) / 4u, (uint64_t)p->merge_blocks);
                HIP_TRY(epik_amd::launch_team_merge(tp, p->team_waves, dim3((unsigned)merge_blocks), stream));
            }
            if (!is_finish(mode)) {
                tp.read_list = p->d_slow_list;
                tp.read_list_count = p->d_front_cursor + 1;
                HIP_TRY(epik_amd::launch_team(tp, p->team_waves, p->counts, (int)mode, dim3((unsigned)blocks),
                                              g.lds_block_bytes, stream));
                // what this launch asked of the pool, for the size of the next one's (read whenever it has arrived)
                HIP_TRY(hipMemcpyAsync(p->h_front_cursor, p->d_front_cursor, 3 * sizeof(unsigned long long),
                                       hipMemcpyDeviceToHost, stream));
            }
        } else {
            HIP_TRY(epik_amd::launch_team(tp, p->team_waves, p->counts, (int)mode, dim3((unsigned)blocks), g.lds_block_bytes,
                                          stream));
        }
    } else if (mode == kFinish) {
        HIP_TRY(epik_amd::launch_finish_reads(pp, p->counts, dim3((unsigned)blocks), dim3(g.waves_per_block * 64u),
                                              g.lds_block_bytes, stream));
    } else {
        HIP_TRY(epik_amd::launch_place_reads(pp, p->layout, p->plan.runs, p->counts, dim3((unsigned)blocks),
                                             dim3(g.waves_per_block * 64u), g.lds_block_bytes, stream));
    }
    if (timed) {
        HIP_TRY(hipEventRecord(p->ev_stop, stream));
        p->ev_recorded = true;
    }
    return EPIK_AMD_OK;
}

int epik_amd_placer_accumulate_device(epik_amd_placer *p, const void *d_seqs, const void *d_seq_offsets,
                                      uint64_t n, void *d_scores, void *d_counts, const void *d_amb_slot,
                                      void *d_amb_order, void *d_amb_avg, void *stream)
{
    if (!p) return fail(EPIK_AMD_ERR_INVALID, "null placer");
    if (n && (!d_seqs || !d_seq_offsets || !d_scores || !d_counts))
        return fail(EPIK_AMD_ERR_INVALID, "null device buffer");
    if (d_amb_slot && (!d_amb_order || !d_amb_avg))
        return fail(EPIK_AMD_ERR_INVALID, "d_amb_slot without d_amb_order / d_amb_avg");
    HIP_TRY(hipSetDevice(p->device));
    shard_buffers shard;
    shard.scores = static_cast<float *>(d_scores);
    shard.counts = static_cast<uint16_t *>(d_counts);
    shard.amb_slot = static_cast<const int32_t *>(d_amb_slot);
    shard.amb_order = static_cast<uint32_t *>(d_amb_order);
    shard.amb_avg = static_cast<float *>(d_amb_avg);
    return launch(p, kAccumulate, d_seqs, d_seq_offsets, n, nullptr, nullptr, nullptr, static_cast<hipStream_t>(stream),
                  shard);
}

int epik_amd_placer_finish_device(epik_amd_placer *p, const void *d_seq_offsets, uint64_t n,
                                  const void *d_scores, const void *d_counts, const void *d_amb_slot,
                                  const void *d_amb_avg, void *d_rows, void *d_n_rows, void *d_kmer_counts,
                                  void *stream)
{
    if (!p) return fail(EPIK_AMD_ERR_INVALID, "null placer");
    if (n == 0) return EPIK_AMD_OK;
    if (!d_seq_offsets || !d_scores || !d_counts || !d_rows || !d_n_rows)
        return fail(EPIK_AMD_ERR_INVALID, "null device buffer");
    if (d_amb_slot && !d_amb_avg) return fail(EPIK_AMD_ERR_INVALID, "d_amb_slot without d_amb_avg");
    HIP_TRY(hipSetDevice(p->device));
    shard_buffers shard;
    shard.scores = const_cast<float *>(static_cast<const float *>(d_scores));
    shard.counts = const_cast<uint16_t *>(static_cast<const uint16_t *>(d_counts));
    shard.amb_slot = static_cast<const int32_t *>(d_amb_slot);
    shard.amb_avg = const_cast<float *>(static_cast<const float *>(d_amb_avg));
    return launch(p, kFinish, nullptr, d_seq_offsets, n, d_rows, d_n_rows, d_kmer_counts,
                  static_cast<hipStream_t>(stream), shard);
}

int epik_amd_placer_partial_info(const epik_amd_placer *p, epik_amd_partial_info *out)
{
    if (!p || !out) return fail(EPIK_AMD_ERR_INVALID, "null argument");
    *out = epik_amd_partial_info{};
    out->num_branches = p->params.num_branches;
    out->lists = p->team && p->team_front ? 1u : 0u;
    out->slices = p->team ? (uint32_t)p->team_waves * p->team_passes : 1u;
    out->slice_rows = p->team ? p->team_slice_rows : p->params.num_branches;
    out->entry_bytes = epik_amd::sparse_entry_bytes(p->counts);
    out->postings_per_kmer = (double)p->plan.kept_entries / (double)std::max<uint64_t>(p->num_keys, 1);
    return EPIK_AMD_OK;
}

int epik_amd_placer_accumulate_lists_device(epik_amd_placer *p, const void *d_seqs, const void *d_seq_offsets,
                                            uint64_t n, uint32_t n_parts, void *d_entries, uint64_t entries_cap,
                                            void *d_index, void *d_part_entries, const void *d_amb_slot,
                                            void *d_amb_order, void *d_amb_avg, void *stream)
{
    if (!p) return fail(EPIK_AMD_ERR_INVALID, "null placer");
    if (n_parts == 0 || n_parts > EPIK_AMD_MAX_SHARDS) return fail(EPIK_AMD_ERR_INVALID, "n_parts must be in [1, EPIK_AMD_MAX_SHARDS]");
    if (!d_part_entries) return fail(EPIK_AMD_ERR_INVALID, "null device buffer");
    HIP_TRY(hipSetDevice(p->device));
    if (n == 0) {
        HIP_TRY(hipMemsetAsync(d_part_entries, 0, n_parts * sizeof(uint64_t), static_cast<hipStream_t>(stream)));
        return EPIK_AMD_OK;
    }
    if (!d_seqs || !d_seq_offsets || !d_index || (!d_entries && entries_cap)) return fail(EPIK_AMD_ERR_INVALID, "null device buffer");
    if (entries_cap >= (1ull << 32)) return fail(EPIK_AMD_ERR_INVALID, "entries_cap must be below 2^32 (offsets inside a part are 32-bit)");
    if (d_amb_slot && (!d_amb_order || !d_amb_avg))
        return fail(EPIK_AMD_ERR_INVALID, "d_amb_slot without d_amb_order / d_amb_avg");
    shard_buffers shard;
    shard.entries = static_cast<uint8_t *>(d_entries);
    shard.entries_cap = entries_cap;
    shard.index = static_cast<uint2 *>(d_index);
    shard.part_total = static_cast<unsigned long long *>(d_part_entries);
    shard.parts = n_parts;
    shard.amb_slot = static_cast<const int32_t *>(d_amb_slot);
    shard.amb_order = static_cast<uint32_t *>(d_amb_order);
    shard.amb_avg = static_cast<float *>(d_amb_avg);
    return launch(p, kAccumulateLists, d_seqs, d_seq_offsets, n, nullptr, nullptr, nullptr, static_cast<hipStream_t>(stream),
                  shard);
}

int epik_amd_placer_finish_lists_device(epik_amd_placer *p, const void *d_seq_offsets, uint64_t n, uint32_t n_shards,
                                        const void *const *d_entries, const void *const *d_index,
                                        const void *d_amb_slot, const void *d_amb_avg, void *d_rows, void *d_n_rows,
                                        void *d_kmer_counts, void *stream)
{
    if (!p) return fail(EPIK_AMD_ERR_INVALID, "null placer");
    if (n == 0) return EPIK_AMD_OK;
    if (n_shards == 0 || n_shards > EPIK_AMD_MAX_SHARDS) return fail(EPIK_AMD_ERR_INVALID, "n_shards must be in [1, EPIK_AMD_MAX_SHARDS]");
    if (!d_seq_offsets || !d_entries || !d_index || !d_rows || !d_n_rows) return fail(EPIK_AMD_ERR_INVALID, "null device buffer");
    if (d_amb_slot && !d_amb_avg) return fail(EPIK_AMD_ERR_INVALID, "d_amb_slot without d_amb_avg");
    epik_amd::SparseSources src{};
    src.n_shards = n_shards;
    for (uint32_t g = 0; g < n_shards; ++g) {
        if (!d_index[g]) return fail(EPIK_AMD_ERR_INVALID, "null index of a shard");
        src.entries[g] = static_cast<const uint8_t *>(d_entries[g]);  // (may be null when the shard sent no entry)
        src.index[g] = static_cast<const uint2 *>(d_index[g]);
    }
    HIP_TRY(hipSetDevice(p->device));
    shard_buffers shard;
    shard.sources = &src;
    shard.amb_slot = static_cast<const int32_t *>(d_amb_slot);
    shard.amb_avg = const_cast<float *>(static_cast<const float *>(d_amb_avg));
    return launch(p, kFinishLists, nullptr, d_seq_offsets, n, d_rows, d_n_rows, d_kmer_counts,
                  static_cast<hipStream_t>(stream), shard);
}

int epik_amd_placer_stream_build(const epik_amd_placer *p, uint32_t *wide, uint32_t *sparse_quads)
{
    if (!p || !wide || !sparse_quads) return fail(EPIK_AMD_ERR_INVALID, "null argument");
    *wide = *sparse_quads = 0;
    if (!p->team || !p->team_front) return EPIK_AMD_OK;
    const auto &g = p->geo[p->counts];
    const bool is_wide = epik_amd::team_stream_is_wide(p->team_waves, g.stream_lds_bytes[1], g.stream_bw[1]);
    *wide = is_wide ? 1u : 0u;
    // (the touched-quad epilogue is compiled into the wide build, for 8- and 16-bit counts)
    *sparse_quads = is_wide && p->counts != epik_amd::kCounts32 ? p->sparse_quads : 0u;
    return EPIK_AMD_OK;
}

int epik_amd_placer_last_path(const epik_amd_placer *p, uint32_t *path)
{
    if (!p || !path) return fail(EPIK_AMD_ERR_INVALID, "null argument");
    *path = !p->team ? EPIK_AMD_PATH_WAVE : p->last_streamed ? EPIK_AMD_PATH_TEAM_STREAMED : EPIK_AMD_PATH_TEAM_ONE_KERNEL;
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
    return launch(p, kPlace, d_seqs, d_seq_offsets, n, d_rows, d_n_rows, d_kmer_counts,
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

static int place_impl(epik_amd_placer *p, const char *seqs, const uint64_t *seq_offsets, uint64_t n,
                      epik_amd_placement *rows, uint32_t *n_rows, uint32_t *kmer_counts);

int epik_amd_placer_place(epik_amd_placer *p, const char *seqs, const uint64_t *seq_offsets,
                          uint64_t n, epik_amd_placement *rows, uint32_t *n_rows,
                          uint32_t *kmer_counts)
{
    try {  // std::thread, std::vector: nothing may leave through the C ABI
        return place_impl(p, seqs, seq_offsets, n, rows, n_rows, kmer_counts);
    } catch (const std::exception &e) {
        if (p) {
            (void)hipStreamSynchronize(p->stream_in);
            (void)hipStreamSynchronize(p->stream);
            (void)hipStreamSynchronize(p->stream_out);
        }
        return fail(EPIK_AMD_ERR_INVALID, std::string("place: ") + e.what());
    }
}

static int place_impl(epik_amd_placer *p, const char *seqs, const uint64_t *seq_offsets, uint64_t n,
                      epik_amd_placement *rows, uint32_t *n_rows, uint32_t *kmer_counts)
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
    {
        // (a width forced by the caller or the environment is kept unless it cannot hold the longest read's
        // k-mers: a read must never come back as EPIK_AMD_ROWS_COUNTS_TOO_NARROW from THIS entry point, whose
        // consumers take n_rows as a row count)
        const int wanted = counts_for(p, longest);
        const uint64_t kmers = longest >= p->params.kmer_size ? longest - p->params.kmer_size + 1 : 0;
        if (!p->counts_forced) {
            p->counts = wanted;
        } else if (kmers > max_kmers_of_counts(p->counts)) {
            p->counts = kmers > max_kmers_of_counts(epik_amd::kCounts16) ? epik_amd::kCounts32 : epik_amd::kCounts16;
        }
    }
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
    p->ev_in.reserve(n_chunks);  // (may throw: caught at the boundary below)
    p->ev_kernel.reserve(n_chunks);
    while (p->ev_in.size() < n_chunks) {  // the two vectors always grow together
        hipEvent_t a = nullptr, b = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        if (const hipError_t e = hipEventCreateWithFlags(&b, hipEventDisableTiming); e != hipSuccess) {
            (void)hipEventDestroy(a);
            return fail_hip(e, "hipEventCreateWithFlags");
        }
        p->ev_in.push_back(a);
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
        rc = launch(p, kPlace, p->d_seqs, p->d_seq_offsets + r0, cnt, p->d_rows + r0 * keep, p->d_n_rows + r0,
                    p->d_counts + r0 * keep, p->stream, {}, b1 - b0);
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
    // whatever happened, nothing of this call may still be writing into the caller's buffers on return
    (void)hipStreamSynchronize(p->stream_in);
    (void)hipStreamSynchronize(p->stream);
    (void)hipStreamSynchronize(p->stream_out);
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
    if (p->team) {
        epik_amd::TeamParams tp{};
        tp.base = pp;
        tp.team_table = p->team_table;
        tp.num_keys = p->plan.table_keys;
        tp.shard_index = p->plan.shard_index, tp.shard_count = p->plan.shard_count;
        tp.team_paired = p->plan.team_paired ? 1u : 0u;
        tp.passes = p->team_passes;
        HIP_TRY(epik_amd::launch_team_algorithmic_bytes(tp, p->team_waves, p->d_total, s));
    } else {
        HIP_TRY(epik_amd::launch_algorithmic_bytes(pp, p->layout, p->plan.runs, p->d_total, s));
    }
    unsigned long long total = 0;
    HIP_TRY(hipMemcpyAsync(&total, p->d_total, sizeof(total), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    *bytes_out = total;
    return EPIK_AMD_OK;
}

int epik_amd_placer_release_scratch(epik_amd_placer *p)
{
    if (!p) return fail(EPIK_AMD_ERR_INVALID, "null placer");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipDeviceSynchronize());
    if (p->shard_state && p->shard_state_free) p->shard_state_free(p->shard_state);
    p->shard_state = nullptr;
    void **bufs[] = {reinterpret_cast<void **>(&p->d_front_hdr), reinterpret_cast<void **>(&p->d_finish_hdr),
                     reinterpret_cast<void **>(&p->d_slow_list), &p->d_slice_rows,
                     &p->d_slice_sums, reinterpret_cast<void **>(&p->d_front_pool), reinterpret_cast<void **>(&p->d_sparse_cap),
                     reinterpret_cast<void **>(&p->d_scan_tiles), reinterpret_cast<void **>(&p->d_seqs),
                     reinterpret_cast<void **>(&p->d_seq_offsets), reinterpret_cast<void **>(&p->d_rows),
                     reinterpret_cast<void **>(&p->d_n_rows), reinterpret_cast<void **>(&p->d_counts)};
    for (void **b : bufs) {
        if (*b) (void)hipFree(*b);
        *b = nullptr;
    }
    p->front_hdr_bytes = 0, p->finish_hdr_bytes = 0, p->slow_list_reads = 0, p->slice_out_reads = 0, p->front_pool_cap = 0, p->sparse_cap_items = 0;
    p->d_seqs_cap = 0, p->d_reads_cap = 0, p->front_failed_reads = 0;
    return EPIK_AMD_OK;
}

int epik_amd_placer_launch_info(const epik_amd_placer *p, uint32_t *waves_per_block,
                                uint32_t *blocks, uint32_t *lds_bytes)
{
    if (!p) return fail(EPIK_AMD_ERR_INVALID, "null placer");
    const auto &g = p->geo[p->last_geo];
    // (the team placement as front + streaming + merge kernels reports its streaming kernel)
    const bool streaming = p->team && p->team_front && p->last_streamed;
    if (waves_per_block) *waves_per_block = streaming ? (uint32_t)g.stream_bw[g.last_stream] : g.waves_per_block;
    if (blocks) *blocks = p->last_blocks ? p->last_blocks : streaming ? g.stream_blocks[g.last_stream] : g.max_blocks;
    if (lds_bytes) *lds_bytes = streaming ? g.stream_lds_bytes[g.last_stream] : g.lds_block_bytes;
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
    // (for the device-pointer entry points only: epik_amd_placer_place() keeps choosing from its batch)
    p->counts = counts_for(p, longest_read);
    p->longest_read_hint = longest_read;
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
