// team_stream.hip -- large trees, the front end as a kernel of its own.
//
// Reference path: as team_kernel.hip (epik/src/epik/place.cpp:278-440, 134-199, 241-267).
//
// team_place_kernel keeps three workgroups on a CU (a read's per-branch vectors fill the LDS), and each
// of them walks through the phases of a read one after the other: characters -> classes -> table entries ->
// chunk descriptors (four dependent trips to memory and two meetings of the waves) before the first posting
// is fetched.  With twelve waves on a CU nothing hides that latency.  Here the placement is three kernels:
//
//   team_front_kernel   one WAVE per read, no per-branch vectors, so the CU is full of waves: encode
//                       (i2l::to_kmers, place.cpp:294), lookup (phylo_kmer_db::search, :300) and the chunk
//                       descriptors of every slice of the branch range, in read order, into a pool in HBM
//                       (8 bytes per chunk of <= 64 postings: ~3 % on top of the postings themselves) plus
//                       a header per read;
//   team_stream_kernel  one workgroup per read as before, but a wave now only loads ITS slice's descriptor
//                       list (coalesced, the header a read ahead), streams it into its rows (:349-371) and
//                       runs the slice epilogue (:418-422, its share of :134-184), whose results -- the
//                       slice's best rows and its share of sum_scores -- go to HBM.  The waves of a
//                       workgroup share nothing but the LDS allocation: no barrier, no counter;
//   team_merge_kernel   one wave per read: the slices' rows ranked together, sum_scores, like-weight
//                       ratios, filter, rows out (:164-199, :241-264).
//
// A read whose descriptors did not fit the pool is put on a list and placed by team_place_kernel afterwards.
// The arithmetic and its order are those of the other kernels: every branch receives its float32 adds from
// one wave, in the k-mer order of the read -- bit-identical scores.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "team_epilogue.hpp"

namespace epik_amd {

// ---------------------------------------------------------------------------------
// Front end.  A read of up to kTilesPerPass tiles (150 bp: three) on a one-pass tree keeps its table
// entries in registers: its characters, then their classes, then the entries each go out together, the
// chunks of every slice are counted, the read takes its descriptors' place in the pool, and the descriptors
// are written.  Any other read is swept twice per pass: once to count, once to write (the table entries
// come out of the L2 the second time).  The pool is handed out to the waves kFrontPoolChunk descriptors at
// a time: one atomic add per read on a single counter costs more than everything else here (measured: 8 ns
// each, device-wide).
// ---------------------------------------------------------------------------------
// (Workgroups of one wave, 3 KB of LDS each.  With 4 slices per pass the kernel is held to the 64 registers of
// eight waves per SIMD -- it needed 65.)
#ifndef EPIK_AMD_FRONT_OCC
#define EPIK_AMD_FRONT_OCC 8
#endif
// kLists (the first half of a k-mer-space-sharded placement that leaves partial LISTS): the front end also says
// how many entries the list of every (read, slice) may take -- the postings of the slice's sublists, at most
// the slice's rows -- into tp.sparse_cap; team_sparse_scan_kernel lays the lists out from that.
template <int W, bool kLists>
// (kLists: the four extra sums do not fit the 64 registers of eight waves per SIMD: six)
__global__ __launch_bounds__(64, W <= 4 ? (kLists ? 6 : EPIK_AMD_FRONT_OCC) : 1) void team_front_kernel(TeamParams tp, uint64_t max_kmers, uint32_t held_passes)
{
    const PlaceParams &p = tp.base;
    const int lane = lane_id();
    const uint32_t k = p.kmer_size;
    const uint32_t sigma = p.alphabet_size;
    const uint32_t stride = kWave - (k - 1);  // windows per 64-character tile
    const uint32_t n_slices = (uint32_t)W * tp.passes;
    const uint64_t waves_per_block = blockDim.x >> 6;
    const uint64_t n_waves = (uint64_t)gridDim.x * waves_per_block;
    const uint64_t null_desc = null_chunk(p);
    constexpr int T = kTilesPerPass;
    typedef typename TeamEntry<W>::raw_t raw_t;
    extern __shared__ __align__(16) unsigned char held_bytes[];
    raw_t *held = reinterpret_cast<raw_t *>(held_bytes);  // [held_passes][T][kQuads][64]: the entries of a short read's tiles, by lane
    constexpr uint32_t kHeldPerPass = T * TeamEntry<W>::kQuads * kWave;
    unsigned long long chunk_at = 0;  // this wave's piece of the pool: next free descriptor, how many are left
    uint32_t chunk_left = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) tp.front_cursor[2] = p.n_reads;  // (the host sizes the next launch's pool by it)
    for (uint64_t read = (uint64_t)blockIdx.x * waves_per_block + (threadIdx.x >> 6); read < p.n_reads; read += n_waves) {
        const uint64_t seq_begin = p.seq_offsets[read];
        const uint64_t len = p.seq_offsets[read + 1] - seq_begin;
        const uint8_t *__restrict__ seq = p.seqs + seq_begin;
        uint32_t *hdr = reinterpret_cast<uint32_t *>(tp.front_hdr + read * tp.front_hdr_stride);
        // header words, one per lane: 0 = first descriptor / 8, 1 = flags, 2 = length, 3 + s = chunks of slice s
        uint32_t word = 0;
        if (len < k || len - k + 1 > max_kmers) {  // no placement / counts too narrow: the consumer reports it
            word = lane == 1 ? (len < k ? kFrontNoRows : kFrontTooNarrow) : lane == 2 ? (uint32_t)len : 0u;
            if ((uint32_t)lane < kFrontHdrWords + n_slices) hdr[lane] = word;
            if constexpr (kLists)
                if ((uint32_t)lane < n_slices) tp.sparse_cap[read * n_slices + (uint32_t)lane] = 0u;
            continue;
        }
        const uint64_t n_kmers = len - k + 1;  // :322
        bool any_amb = false;
        // one tile: which windows are exact k-mers, and (cold) whether any is ambiguous (:306-313)
        auto exact_windows = [&](const Tile &tl) {
            bool exact = tl.in_range;
            if ((tl.inv_mask | tl.amb_mask) != 0) {  // wave-uniform, cold
                const uint64_t wmask = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
                const uint64_t inv_w = (tl.inv_mask >> lane) & wmask;
                const uint64_t amb_w = (tl.amb_mask >> lane) & wmask;
                const bool is_amb = tl.in_range && inv_w == 0 && __popcll(amb_w) == 1;
                exact = tl.in_range && inv_w == 0 && amb_w == 0;
                any_amb = any_amb || __ballot(is_amb) != 0;
            }
            return exact;
        };
        const bool one_group = tp.passes <= held_passes && n_kmers <= (uint64_t)T * stride;  // wave-uniform
        uint32_t postings = 0;  // kLists, lane 3 + s: the postings of slice s's sublists
        // ---- chunks per slice -------------------------------------------------------------------------
        if (one_group) {
            uint32_t ch[T], cls[T];
#pragma unroll
            for (int t = 0; t < T; ++t) ch[t] = tile_char(seq, len, (uint64_t)t * stride);
#pragma unroll
            for (int t = 0; t < T; ++t) cls[t] = tile_class(ch[t], len, (uint64_t)t * stride, p.char_class);
            // the entries wait in LDS for the second half (in registers, unrolled over tiles and slices, they
            // cost the kernel three quarters of its waves); the tiles are encoded once for all passes
            Tile tiles[T];
            bool exact[T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                tiles[t] = tile_from_class(cls[t], len, (uint64_t)t * stride, n_kmers, k, sigma, stride);
                exact[t] = exact_windows(tiles[t]);
            }
            for (uint32_t pass = 0; pass < tp.passes; ++pass) {
                uint32_t acc[W];
#pragma unroll
                for (int s = 0; s < W; ++s) acc[s] = 0;
                raw_t raw[T][TeamEntry<W>::kQuads];
#pragma unroll
                for (int t = 0; t < T; ++t) {
#pragma unroll
                    for (int q = 0; q < TeamEntry<W>::kQuads; ++q) raw[t][q] = TeamEntry<W>::zero_raw();
                    if (exact[t]) TeamEntry<W>::fetch(tp, pass, tiles[t].key, (uint32_t)t * stride + (uint32_t)lane, raw[t]);
                }
                uint32_t plen[W];
#pragma unroll
                for (int s = 0; s < W; ++s) plen[s] = 0;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    TeamEntry<W> e;
                    e.unpack(raw[t]);
#pragma unroll
                    for (int s = 0; s < W; ++s) {
                        acc[s] += (e.len[s] + (uint32_t)kWave - 1u) >> 6;
                        if constexpr (kLists) plen[s] += e.len[s];
                    }
#pragma unroll
                    for (int q = 0; q < TeamEntry<W>::kQuads; ++q)
                        held[pass * kHeldPerPass + (t * TeamEntry<W>::kQuads + q) * kWave + lane] = raw[t][q];
                }
#pragma unroll
                for (int s = 0; s < W; ++s) {
                    const uint32_t total = wave_sum_u32(acc[s]);
                    word = ((uint32_t)lane == kFrontHdrWords + pass * W + (uint32_t)s) ? total : word;
                    if constexpr (kLists) {
                        const uint32_t all = wave_sum_u32(plen[s]);
                        postings = ((uint32_t)lane == kFrontHdrWords + pass * W + (uint32_t)s) ? all : postings;
                    }
                }
            }
        } else {
            for (uint32_t pass = 0; pass < tp.passes; ++pass) {
                uint32_t acc[W], plen[W];
#pragma unroll
                for (int s = 0; s < W; ++s) acc[s] = plen[s] = 0;
                for (uint64_t tile_pos = 0; tile_pos < n_kmers; tile_pos += stride) {
                    const Tile tl = encode_tile(seq, len, tile_pos, n_kmers, k, sigma, stride, p.char_class);
                    if (exact_windows(tl)) {
                        TeamEntry<W> e;
                        e.load(tp, pass, tl.key, (uint32_t)tile_pos + (uint32_t)lane);
#pragma unroll
                        for (int s = 0; s < W; ++s) {
                            acc[s] += (e.len[s] + (uint32_t)kWave - 1u) >> 6;
                            // (saturating: a very long read's sublists may name more postings than 32 bits hold;
                            // the bound below is the slice's rows then)
                            if constexpr (kLists) plen[s] = plen[s] + e.len[s] < plen[s] ? 0xffffffffu : plen[s] + e.len[s];
                        }
                    }
                }
#pragma unroll
                for (int s = 0; s < W; ++s) {
                    const uint32_t total = wave_sum_u32(acc[s]);
                    word = ((uint32_t)lane == kFrontHdrWords + pass * W + (uint32_t)s) ? total : word;
                    if constexpr (kLists) {
                        // (lane sums below 2^32 / 64 add up without wrapping; anything larger is "all rows")
                        const bool huge = __ballot(plen[s] >= (1u << 25)) != 0;
                        const uint32_t all = huge ? 0xffffffffu : wave_sum_u32(plen[s]);
                        postings = ((uint32_t)lane == kFrontHdrWords + pass * W + (uint32_t)s) ? all : postings;
                    }
                }
            }
        }
        // ---- the read's place in the pool: every slice's list rounded up to the ring ---------------
        const bool is_count = (uint32_t)lane >= kFrontHdrWords && (uint32_t)lane < kFrontHdrWords + n_slices;
        const uint32_t padded = is_count ? (word + kTeamRing - 1u) & ~(kTeamRing - 1u) : 0u;
        const uint32_t incl = wave_incl_scan_u32(padded);
        const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
        const uint32_t first_of_slice = incl - padded;  // lane 3 + s: where slice s starts, in descriptors from the read's first
        unsigned long long off = 0;
        if (total > kFrontPoolChunk) {  // a long read: exactly what it needs
            if (lane == 0) off = atomicAdd(tp.front_cursor, (unsigned long long)total);
            off = readlane_u64(off, 0);
        } else if (total != 0) {
            if (total > chunk_left) {  // the rest of the old piece is lost (on average half a read's worth)
                if (lane == 0) off = atomicAdd(tp.front_cursor, (unsigned long long)kFrontPoolChunk);
                chunk_at = readlane_u64(off, 0);
                chunk_left = kFrontPoolChunk;
            }
            off = chunk_at;
            chunk_at += total;
            chunk_left -= total;
        }
        // A read with an ambiguous k-mer (place.cpp:306-313, 373-415) is left to team_place_kernel: its sweep
        // of the resolved keys (double-precision pow) needs more registers than everything else together, and
        // the streaming kernel without it leaves room on a SIMD for the kernels that run beside it.
        const bool fits = off + total <= tp.front_pool_cap && !any_amb;
        uint32_t flags = any_amb ? kFrontAmbiguous : 0u;
        if (!fits) {
            flags |= kFrontSlow;
            if (lane == 0) tp.slow_list[atomicAdd(tp.front_cursor + 1, 1ull)] = read;
        }
        if constexpr (kLists) {
            // Room for the list of (read, slice): a posting touches one row, and the slice has only so many.  A
            // read whose ambiguous k-mers are ADDED by this shard (no slot: one shard only, epik_amd.h) may touch
            // any row.
            if (is_count) {
                const uint32_t slice = (uint32_t)lane - kFrontHdrWords;
                const uint32_t first = slice * tp.slice_rows;
                const uint32_t rows = first >= p.num_branches ? 0u : min(tp.slice_rows, p.num_branches - first);
                const bool adds_amb = any_amb && !(p.amb_slot && p.amb_slot[read] >= 0);
                tp.sparse_cap[read * n_slices + slice] = adds_amb ? rows : min(postings, rows);
            }
        }
        word = lane == 0 ? (uint32_t)(off >> 3) : lane == 1 ? flags : lane == 2 ? (uint32_t)len : word;
        if ((uint32_t)lane < kFrontHdrWords + n_slices) hdr[lane] = word;
        if (!fits || total == 0) continue;
        // ---- the descriptors ---------------------------------------------------------------------------
        uint64_t *__restrict__ out = tp.front_pool + off;
        // the chunks of one tile's sublists of slice s, from descriptor `run` of the read on; returns their number
        auto write_sublists = [&](uint32_t llen, uint64_t start, uint32_t run) {
            const uint32_t nch = (llen + (uint32_t)kWave - 1u) >> 6;
            const uint32_t scan = wave_incl_scan_u32(nch);
            const uint32_t at = run + scan - nch;
            // every lane writes the first kOwnChunks chunks of its own sublist; the rest of a longer one is
            // written by the whole wave, lane j writing chunk kOwnChunks + j
            constexpr uint32_t kOwnChunks = 3;
#pragma unroll
            for (uint32_t c = 0; c < kOwnChunks; ++c) {
                if (nch > c) {
                    const uint32_t rest = llen - (c << 6);
                    const uint64_t cnt = rest < (uint32_t)kWave ? rest : (uint32_t)kWave;
                    out[at + c] = chunk_address<TeamChunks>(p, start, c) | (cnt << 48);
                }
            }
            uint64_t long_lists = __ballot(nch > kOwnChunks);
            while (long_lists) {
                const int m = __builtin_ctzll(long_lists);
                long_lists &= long_lists - 1;
                const uint32_t l_first = __builtin_amdgcn_readlane(at, m);
                const uint32_t l_len = __builtin_amdgcn_readlane(llen, m);
                const uint64_t l_start = readlane_u64(start, m);
                for (uint32_t c = (uint32_t)lane + kOwnChunks; (c << 6) < l_len; c += kWave) {
                    const uint32_t rest = l_len - (c << 6);
                    const uint64_t cnt = rest < (uint32_t)kWave ? rest : (uint32_t)kWave;
                    out[l_first + c] = chunk_address<TeamChunks>(p, l_start, c) | (cnt << 48);
                }
            }
            return __builtin_amdgcn_readlane(scan, 63);
        };
        for (uint32_t pass = 0; pass < tp.passes; ++pass) {
            uint32_t run[W];  // where the next tile's chunks of slice s go (scalar)
#pragma unroll
            for (int s = 0; s < W; ++s)
                run[s] = __builtin_amdgcn_readlane(first_of_slice, (int)kFrontHdrWords + (int)(pass * W) + s);
            if (one_group) {
#pragma unroll 1
                for (int t = 0; t < T; ++t) {
                    raw_t raw[TeamEntry<W>::kQuads];
#pragma unroll
                    for (int q = 0; q < TeamEntry<W>::kQuads; ++q)
                        raw[q] = held[pass * kHeldPerPass + (t * TeamEntry<W>::kQuads + q) * kWave + lane];
                    TeamEntry<W> e;
                    e.unpack(raw);
#pragma unroll
                    for (int s = 0; s < W; ++s) run[s] += write_sublists(e.len[s], e.start(s), run[s]);
                }
            } else {
                for (uint64_t tile_pos = 0; tile_pos < n_kmers; tile_pos += stride) {
                    const Tile tl = encode_tile(seq, len, tile_pos, n_kmers, k, sigma, stride, p.char_class);
                    TeamEntry<W> e;
                    e.line = 0;
#pragma unroll
                    for (int s = 0; s < W; ++s) e.len[s] = 0;
                    if (exact_windows(tl)) e.load(tp, pass, tl.key, (uint32_t)tile_pos + (uint32_t)lane);
#pragma unroll
                    for (int s = 0; s < W; ++s) run[s] += write_sublists(e.len[s], e.start(s), run[s]);
                }
            }
            // the padding behind every list: chunks of zero bytes
#pragma unroll
            for (int s = 0; s < W; ++s) {
                const int slot = (int)kFrontHdrWords + (int)(pass * W) + s;
                const uint32_t end = __builtin_amdgcn_readlane(first_of_slice, slot) + __builtin_amdgcn_readlane(padded, slot);
                if (run[s] + (uint32_t)lane < end) out[run[s] + (uint32_t)lane] = null_desc;
            }
        }
    }
}

// The slice epilogue.  (While the streaming kernel still held the ambiguous sweep it was a function out of
// line, and then with at most 16 argument registers: a larger set of arguments goes through the stack --
// scratch memory --, a round trip to the caches at every call.  Without that callee the kernel takes 96
// vector registers with the epilogue inline, and no scratch.)
// (the dense epilogue of place_device.hpp; an item that touched few rows takes team_epilogue.hpp's instead)
template <int W, typename CountT>
__device__ __forceinline__ void slice_epilogue(const TeamParams *__restrict__ ktp, WaveLds<CountT> lds,
                                                         uint32_t n_kmers, SliceArgs a)
{
    TeamCtx<W, true> ctx;
    ctx.rows_pad_ = a.rows_pad;
    ctx.rows_ = a.rows;
    ctx.base_ = a.base;
    ctx.slice_ = ctx.pass_ = 0;  // (what the ambiguous sweep looks lists up by: not used here)
    ctx.kmer_size_ = a.kmer_size;
    ctx.keep_ = a.keep;
    ctx.log_threshold_ = a.log_threshold;
    ctx.cand = static_cast<v4u *>(ktp->slice_rows_out) + (uint64_t)a.slice_at * a.keep;
    ctx.partial = static_cast<TeamPartial *>(ktp->slice_sums_out) + a.slice_at;
    ctx.trace_at_ = a.trace_at;
    ctx.untouched_ = a.untouched != 0;
    place_epilogue_body<TeamChunks, CountT>(&ktp->base, lds, 0ull, (uint64_t)n_kmers, ctx);
}

// ---------------------------------------------------------------------------------
// Stream + slice epilogue.  LDS: the rows of the workgroup's kStreamWaves slices | their descriptor lists (a
// wave's list also holds its top-k candidates and, with 8-bit counts, the "seen" bits of the ambiguous sweep).
// A wave takes one slice of every read of its workgroup and shares nothing with the other waves: no barrier,
// no counter.  (A workgroup is four waves whatever the number of slices: what counts is how many waves a CU
// holds, and with 8 slices per pass the slices are small enough for twenty.)
// ---------------------------------------------------------------------------------
// kMode: kTeamModePlace; kTeamModeAccumulate / kTeamModeFinish = the two halves of a k-mer-space-sharded
// placement (include/epik_amd.h): stream only, the slice's raw sums and counts to HBM / the totals back from
// HBM, slice epilogue (the headers then come from team_header_kernel: there is no front end).
// kTeamModeAccumulateLists / kTeamModeFinishLists: the same halves with partial LISTS in place of the dense
// vectors -- the rows a slice really touched, compacted (emit_partial_list), and the shards' lists of a slice
// added back in shard order (merge_partial_lists; `src` says where they lie).
// kWide: the build for slices so large that LDS keeps a CU to twelve waves (db_layout.h: stream_wide) -- 168 vector
// registers, and with them the slice epilogue over the touched quads.
// kBW: waves of a workgroup, 4 or 2 (db_layout.h: stream_block_waves)
template <int W, typename CountT, int kMode, bool kWide, int kBW>
__global__ __launch_bounds__(kBW * 64, stream_waves_per_simd(W)) void team_stream_kernel(TeamParams tp, SparseSources src)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    typedef WaveLds<CountT> Lds;
    const PlaceParams &p = tp.base;
    const int lane = lane_id();
    const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // W slices per pass, kStreamWaves waves per workgroup: W / kStreamWaves consecutive workgroups share a read -- or,
    // with two slices per pass, a workgroup holds two reads
    constexpr uint32_t kParts = stream_parts(W, kBW), kReadsPerBlock = stream_reads_per_block(W, kBW);
    static_assert(W % kBW == 0 || kBW % W == 0, "slices per pass: a multiple or a divisor of the workgroup's waves");
    const uint32_t wave = kReadsPerBlock > 1 ? wave_in_block % (uint32_t)W
                                             : (blockIdx.x % kParts) * kBW + wave_in_block;  // this wave's slice of a pass
    // (the grid is a multiple of kParts)
    const uint64_t first_read = kReadsPerBlock > 1 ? (uint64_t)blockIdx.x * kReadsPerBlock + wave_in_block / (uint32_t)W : blockIdx.x / kParts;
    const uint64_t read_stride = kReadsPerBlock > 1 ? (uint64_t)gridDim.x * kReadsPerBlock : gridDim.x / kParts;
    const uint32_t rows_pad = tp.rows_pad;
    Lds lds;
    unsigned char *desc_base = lds_raw + (size_t)kBW * tp.slice_bytes;
    lds.score = (typename Lds::f32_t *)reinterpret_cast<float *>(lds_raw + (size_t)wave_in_block * tp.slice_bytes);
    lds.count = (typename Lds::count_t *)reinterpret_cast<CountT *>(lds_raw + (size_t)wave_in_block * tp.slice_bytes + (size_t)rows_pad * 4);
    lds.desc = (typename Lds::u64_t *)reinterpret_cast<uint64_t *>(desc_base + (size_t)wave_in_block * tp.desc_bytes);
    // items that touched few rows take the epilogue over their touched quads (team_epilogue.hpp); 32-bit counts
    // (reads of 32 768 k-mers or more) always the dense one
    constexpr bool kSparseCounts = kWide && sizeof(CountT) <= 2;
    const uint32_t n_slices = W * tp.passes;
    const uint32_t score_top = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds.score + (rows_pad - 1u) * 4u);
    const uint32_t count_top = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds.count +
                                                              (rows_pad - 1u) * (uint32_t)sizeof(CountT));
    const PlaceParams *kp = (const PlaceParams *)__builtin_amdgcn_kernarg_segment_ptr();  // = &tp.base
    for (uint32_t i = lane; i < rows_pad; i += kWave) lds.store(i, 0u, 0u);
    // partial lists out: where every part's entries begin (lane r: part r), from the scan kernel's totals
    uint64_t part_first = 0;
    if constexpr (kMode == kTeamModeAccumulateLists) {
        uint64_t mine = (uint32_t)lane < tp.sparse_parts ? (uint64_t)tp.sparse_part_total[lane] : 0ull;
        for (uint32_t r = 0; r < tp.sparse_parts; ++r) {  // (a handful of parts: a serial sum of uniform values)
            const uint64_t t = readlane_u64(mine, (int)r);
            if ((uint32_t)lane > r) part_first += t;
        }
    }

    const uint32_t k = p.kmer_size;
    const uint32_t keep = p.keep_at_most;
    const uint32_t cap = tp.desc_cap;  // descriptors per round (a multiple of the ring; + one trip of spare entries <= 64)
    const uint64_t null_desc = null_chunk(p);
    v4u *rows_out = static_cast<v4u *>(tp.slice_rows_out);
    TeamPartial *sums_out = static_cast<TeamPartial *>(tp.slice_sums_out);
#ifdef EPIK_AMD_ABLATION
    // the timeline of one wave (wave 1 of workgroup 0), EPIK_AMD_STAMPS=1: see EPI_STAMP in place_device.hpp
    const bool traced = p.dbg && blockIdx.x == 0 && wave_in_block == 1;
    uint32_t trace_at = 0;
#define STREAM_STAMP(k)                                                       \
    if (traced) {                                                             \
        if (lane == 0 && trace_at < 100000u) {                                \
            p.dbg[64 + 2 * (size_t)trace_at] = 100ull + (k);                  \
            p.dbg[65 + 2 * (size_t)trace_at] = __builtin_amdgcn_s_memtime();  \
        }                                                                     \
        ++trace_at;                                                           \
    }
#else
#define STREAM_STAMP(k)
#endif

    // A read's header (one word per lane: 0 first descriptor / 8, 1 flags, 2 length, 3 + s chunks of slice s)
    // is loaded two reads ahead, and this wave's first descriptors of a read one read ahead -- just before the
    // read in front of it streams, so that they arrive under its postings (loads return in order: the ring's
    // first wait covers them) and nothing is in flight when the out-of-line functions are entered.
    auto load_header = [&](uint64_t r) {
        const uint32_t *hdr = reinterpret_cast<const uint32_t *>(tp.front_hdr + r * tp.front_hdr_stride);
        return (uint32_t)lane < kFrontHdrWords + n_slices ? hdr[lane] : 0u;
    };
    // where the descriptors of slice `slot - kFrontHdrWords` of the read with header `word` lie, and how many
    auto slice_list = [&](uint32_t word, int slot, uint32_t &my_padded, uint32_t *my_chunks = nullptr) {
        const bool is_count = (uint32_t)lane >= kFrontHdrWords && (uint32_t)lane < kFrontHdrWords + n_slices;
        const uint32_t padded = is_count ? (word + kTeamRing - 1u) & ~(kTeamRing - 1u) : 0u;
        const uint32_t first_of_slice = wave_incl_scan_u32(padded) - padded;
        my_padded = __builtin_amdgcn_readlane(padded, slot);
        if (my_chunks) *my_chunks = __builtin_amdgcn_readlane(word, slot);  // ... of which this many are chunks, the rest padding
        const uint64_t first_desc = (uint64_t)__builtin_amdgcn_readlane(word, 0) << 3;
        return (const uint64_t *)(tp.front_pool + first_desc + __builtin_amdgcn_readlane(first_of_slice, slot));
    };
    auto first_round = [&](uint32_t word) {
        const uint32_t flags = __builtin_amdgcn_readlane(word, 1);
        uint64_t d = null_desc;
        if (kMode != kTeamModeFinish && kMode != kTeamModeFinishLists && !(flags & (kFrontSlow | kFrontNoRows | kFrontTooNarrow))) {
            uint32_t my_padded;
            const uint64_t *list = slice_list(word, (int)(kFrontHdrWords + wave), my_padded);
            const uint32_t n_first = my_padded < cap ? my_padded : cap;
            if ((uint32_t)lane < n_first) d = list[lane];
        }
        return d;
    };
    uint32_t word_cur = 0, word_next = 0;
    uint64_t desc_cur = null_desc;
    // ... and, in the accumulate halves, the postings of its first chunks (the ring's first fill) are asked for behind
    // this read's stream: a trip to HBM that the wave would otherwise sit out at the start of every read
    Preloaded<(int)kTeamRing> pre;
    // (the accumulate halves only: there the wave's next read would otherwise start on an idle memory system behind a
    // list emission that asks nothing of it; in a placement the ring of the next read starts under the epilogue's LDS
    // work anyway, and since an empty ring's first trip only issues loads the fetch ahead costs more instructions
    // than it hides: 64.1 M reads/s without against 62.6 M with, same box, N = 9 999)
    constexpr bool kFetchAhead = kMode == kTeamModeAccumulate || kMode == kTeamModeAccumulateLists;
    // partial lists of an item that streamed one trip of the ring: listed from the cells the ring held
#ifndef EPIK_AMD_CELL_LISTS
#define EPIK_AMD_CELL_LISTS 1
#endif
    constexpr bool kCellLists = EPIK_AMD_CELL_LISTS != 0 && kMode == kTeamModeAccumulateLists;
    auto preload = [&](uint64_t descs) {
        if constexpr (kFetchAhead) {
#pragma unroll
            for (int i = 0; i < (int)kTeamRing; ++i) preload_chunk<TeamChunks>(readlane_u64(descs, i), pre.cell[i], pre.score[i]);
        }
    };
    // partial lists in (finish): the walk over the shards' lists of this item and of the one behind it
    [[maybe_unused]] ListWalk<CountT> walk, walk_next;
    [[maybe_unused]] ListSources lists;
    if constexpr (kMode == kTeamModeFinishLists) lists.load(src);
    // partial lists out (accumulate), lane pass: where the list of (read, pass * W + wave) begins in its part and how
    // much room it has -- loaded a read ahead, like the descriptors
    [[maybe_unused]] uint32_t room_cur = 0, first_cur = 0;
    auto load_room = [&](uint64_t r, uint32_t &first_out, uint32_t &room_out) {
        if constexpr (kMode == kTeamModeAccumulateLists) {
            first_out = room_out = 0;
            if ((uint32_t)lane < tp.passes) {
                const uint64_t at = r * n_slices + (uint32_t)lane * W + wave;
                first_out = tp.sparse_index[at].x;
                room_out = tp.sparse_cap[at];
            }
        }
    };
    if constexpr (kMode == kTeamModeAccumulateLists)
        if (first_read < p.n_reads) load_room(first_read, first_cur, room_cur);
    if (first_read < p.n_reads) {
        word_cur = load_header(first_read);
        if (first_read + read_stride < p.n_reads) word_next = load_header(first_read + read_stride);
        desc_cur = first_round(word_cur);
    }
    if constexpr (kMode != kTeamModeFinish && kMode != kTeamModeFinishLists) preload(desc_cur);
    for (uint64_t read = first_read; read < p.n_reads; read += read_stride) {
        const uint32_t word = word_cur;
        const uint64_t my_desc = desc_cur;
        word_cur = word_next;
        desc_cur = read + read_stride < p.n_reads ? first_round(word_cur) : null_desc;
        if (read + 2ull * read_stride < p.n_reads) word_next = load_header(read + 2ull * read_stride);
        [[maybe_unused]] const uint32_t my_first = first_cur, my_room = room_cur;
        if constexpr (kMode == kTeamModeAccumulateLists)
            if (read + read_stride < p.n_reads) load_room(read + read_stride, first_cur, room_cur);
        STREAM_STAMP(9)  // next read
        const uint32_t flags = __builtin_amdgcn_readlane(word, 1);
        // place.cpp:322 underflows for len < k; we report "no placement".  A read with more k-mers than this
        // launch's counts hold is marked (the caller chose the count width).  team_merge_kernel writes both.
        // A read whose descriptors are not in the pool is placed by team_place_kernel after this launch.
        if (kMode == kTeamModeAccumulate && (flags & (kFrontNoRows | kFrontTooNarrow))) {  // an all-zero partial vector
            for (uint32_t pass = 0; pass < tp.passes; ++pass) {
                const uint32_t base = (pass * W + wave) * tp.slice_rows;
                const uint32_t rows = base >= p.num_branches ? 0u : min(tp.slice_rows, p.num_branches - base);
                for (uint32_t i = lane; i < rows; i += kWave) {
                    p.partial_scores[read * p.num_branches + base + i] = 0.0f;
                    p.partial_counts[read * p.num_branches + base + i] = 0u;
                }
            }
            preload(desc_cur);  // (every turn of the loop leaves the first postings of the wave's next read on their way)
            continue;
        }
        if (flags & (kFrontNoRows | kFrontTooNarrow | kFrontSlow)) {
            if constexpr (kMode != kTeamModeFinish && kMode != kTeamModeFinishLists) preload(desc_cur);
            continue;
        }
        const uint64_t len = __builtin_amdgcn_readlane(word, 2);
        const uint64_t n_kmers = len - k + 1;  // :322

        for (uint32_t pass = 0; pass < tp.passes; ++pass) {
            TeamCtx<W, true> ctx;
            ctx.rows_pad_ = rows_pad;
            ctx.kmer_size_ = k;
            ctx.keep_ = keep;
            ctx.log_threshold_ = p.log_threshold;
            ctx.slice_ = wave;
            ctx.pass_ = pass;
            ctx.base_ = (pass * W + wave) * tp.slice_rows;
            ctx.rows_ = ctx.base_ >= p.num_branches ? 0u : min(tp.slice_rows, p.num_branches - ctx.base_);
            const uint64_t slice_at = read * n_slices + pass * W + wave;
            ctx.cand = rows_out + slice_at * keep;
            ctx.partial = sums_out + slice_at;
            // nothing reached this slice's rows: they are as the last reset left them (wave-uniform)
            [[maybe_unused]] bool slice_untouched = false;
            // ... or an estimate of how much did, in chunks of 64 postings (0: no idea)
            [[maybe_unused]] uint32_t sparse_hint = 0u, my_chunks = 0, my_padded = 0;
            // AccumulateLists, an item whose stream is one trip of the ring: the cells the ring held
            [[maybe_unused]] uint32_t trip_cells[kTeamRing];
            if constexpr (kMode == kTeamModeFinish) {
                // second half of a k-mer-space-sharded placement: the slice's totals come back from HBM, with
                // the read's ambiguous record (the average of the first ambiguous key that reached the branch
                // over all shards, place.cpp:385-388) added as the one-pass loop adds it (:409-410)
                const int64_t slot = p.amb_slot ? (int64_t)p.amb_slot[read] : -1;
                for (uint32_t i = lane; i < ctx.rows_; i += kWave) {
                    const uint64_t at = read * p.num_branches + ctx.base_ + i;
                    float sc = p.partial_scores[at];
                    uint32_t c = p.partial_counts[at];
                    if (slot >= 0) {
                        const float avg = p.amb_avg[(uint64_t)slot * p.num_branches + ctx.base_ + i];
                        if (avg > 0.0f) {
                            sc = __fadd_rn(sc, avg);
                            c += 1u;
                        }
                    }
                    lds.store(i, __float_as_uint(sc), c);
                }
            } else if constexpr (kMode == kTeamModeFinishLists) {
                // ... the same from the shards' partial lists, in shard order; then the ambiguous record.  The
                // item behind this one (the read's next pass, or the wave's next read) has its index asked for
                // before this one's lists are added and its first entries before this one's epilogue: the two
                // dependent trips to memory of an item lie under the work on the item in front of it.
                // (the copy of the prepared walk HERE, behind the epilogue of the item in front: its first entries are
                // on their way until then, and a copy is a use)
                if (walk_next.item == slice_at) walk = walk_next;
                // (the read's slot is asked for here, not behind the requests below: vector memory operations
                // return in order, and waiting for this one would wait for those)
                const int64_t slot = p.amb_slot ? (int64_t)p.amb_slot[read] : -1;
                if (walk.item != slice_at) walk.request(lists, slice_at);
                uint64_t next_item = ~0ull;
                if (pass + 1 < tp.passes)
                    next_item = slice_at + W;
                else if (read + read_stride < p.n_reads &&
                         !(__builtin_amdgcn_readlane(word_cur, 1) & (kFrontSlow | kFrontNoRows | kFrontTooNarrow)))
                    next_item = (read + read_stride) * n_slices + wave;
                walk_next.item = ~0ull;
                if (next_item != ~0ull) walk_next.request(lists, next_item);
                walk.run(lists, lds, rows_pad - 1u);
                if (next_item != ~0ull) walk_next.start(lists, rows_pad - 1u);
                STREAM_STAMP(5)  // lists added
                {   // (lanes without a shard hold 0; an overflowed list counts as a long one)
                    const uint32_t entries = wave_sum_u32(walk.count > 0xffffu ? 0xffffu : walk.count);
                    slice_untouched = slot < 0 && entries == 0;
                    sparse_hint = (entries + 31u) >> 5;  // (a chunk streams about 32 postings)
                }
                if (slot >= 0) {
                    for (uint32_t i = lane; i < ctx.rows_; i += kWave) {
                        const float avg = p.amb_avg[(uint64_t)slot * p.num_branches + ctx.base_ + i];
                        if (avg > 0.0f) {
                            const uint2 cv = lds.load(i);
                            lds.store(i, __float_as_uint(__fadd_rn(__uint_as_float(cv.x), avg)), cv.y + 1u);
                        }
                    }
                }
            } else {
            // ---- exact k-mers, read order (place.cpp:349-371): this slice's descriptor list, a round at a time
            const uint64_t *__restrict__ my_list = slice_list(word, (int)(kFrontHdrWords + pass * W + wave), my_padded, &my_chunks);
            slice_untouched = my_chunks == 0;
            sparse_hint = my_chunks;
            for (uint32_t r0 = 0; r0 < my_padded; r0 += cap) {
                const uint32_t n_round = min(my_padded - r0, cap);  // a multiple of the ring
                const uint32_t n_chunks = min(my_chunks - r0, n_round);  // (r0 < my_chunks: the padding is less than a trip)
                // all 64 entries: the round's descriptors, then null chunks (the ring reads one trip ahead)
                uint64_t d = my_desc;
                if (pass != 0 || r0 != 0) d = (uint32_t)lane < n_round ? my_list[r0 + (uint32_t)lane] : null_desc;
                lds.desc[lane] = d;
                STREAM_STAMP(0)  // descriptors
#ifdef EPIK_AMD_ABLATION
                if (p.ablate & 8u) continue;  // (timing experiments: nothing streamed)
#endif
                // (partial lists: the cells the ring held stay -- of a stream of one trip, the rows it touched)
                if constexpr (kCellLists) {
                    if (kFetchAhead && pass == 0 && r0 == 0)
                        stream_round<TeamChunks, CountT, (int)kTeamRing, true>(p, lds.desc, n_round, score_top, count_top, &pre, n_chunks, trip_cells);
                    else
                        stream_round<TeamChunks, CountT, (int)kTeamRing, false>(p, lds.desc, n_round, score_top, count_top, nullptr, n_chunks, trip_cells);
                } else if (kFetchAhead && pass == 0 && r0 == 0) {
                    stream_round<TeamChunks, CountT, (int)kTeamRing, true>(p, lds.desc, n_round, score_top, count_top, &pre, n_chunks);
                } else {
                    stream_round<TeamChunks, CountT, (int)kTeamRing, false>(p, lds.desc, n_round, score_top, count_top, nullptr, n_chunks);
                }
                STREAM_STAMP(1)  // stream
            }
            // the wave's next read: its first chunks' postings, on their way under what follows
            if (pass + 1 == tp.passes) preload(desc_cur);
            // (no ambiguous k-mers here: the front kernel leaves such a read to team_place_kernel)
            }
            if constexpr (kMode == kTeamModeAccumulate) {
                // k-mer-space shard: the slice's raw sums and counts leave for HBM; the rows are reset
                for (uint32_t i = lane; i < rows_pad; i += kWave) {
                    const uint2 cv = lds.load(i);
                    if (i < ctx.rows_) {
                        const uint64_t at = read * p.num_branches + ctx.base_ + i;
                        p.partial_scores[at] = __uint_as_float(cv.x);
                        p.partial_counts[at] = (uint16_t)(cv.y & ~(uint32_t)Lds::kSeen);
                    }
                    lds.store(i, 0u, 0u);
                }
                continue;
            }
            if constexpr (kMode == kTeamModeAccumulateLists) {
                // ... as a list of the rows that received a k-mer, where the scan kernel made room for it
                const uint32_t room = (uint32_t)__builtin_amdgcn_readlane(my_room, (int)pass);
                const uint32_t part = (uint32_t)(read / tp.sparse_part_reads);
                const uint64_t first = readlane_u64(part_first, (int)part) + (uint32_t)__builtin_amdgcn_readlane(my_first, (int)pass);
                const bool fits = first + room <= tp.sparse_entries_cap;
                uint32_t n_out = 0;
                bool done = slice_untouched;  // (a shard's lists reach a small part of a slice: nothing, or a few dozen quads)
                if constexpr (kCellLists) {
                    if (!done && my_padded == kTeamRing) {
                        if (lane == 0) lds.store(rows_pad - 1u, 0u, 0u);  // the dummy row of the out-of-range lanes
                        n_out = emit_partial_list_cells<CountT, (int)kTeamRing>(lds, rows_pad, trip_cells, my_chunks,
                                                                               tp.sparse_entries + first * PartialEntry<CountT>::kBytes, fits ? room : 0u);
                        done = true;
                    }
                }
                if constexpr (kSparseCounts) {
                    if (!done && my_chunks <= tp.sparse_chunks) {
                        if (lane == 0) lds.store(rows_pad - 1u, 0u, 0u);  // the dummy row of the out-of-range lanes
                        done = emit_partial_list_sparse<CountT>(lds, rows_pad, tp.sparse_quads, tp.sparse_entries + first * PartialEntry<CountT>::kBytes,
                                                                fits ? room : 0u, &n_out);
                    }
                }
                if (!done)
                    n_out = emit_partial_list<CountT>(lds, rows_pad, ctx.rows_, tp.sparse_entries + first * PartialEntry<CountT>::kBytes,
                                                      fits ? room : 0u, p.ablate);
                if (lane == 0) tp.sparse_index[slice_at].y = fits ? n_out : kSparseOverflow;
                STREAM_STAMP(4)  // list emitted
                continue;
            }
            // ---- correction, the slice's best rows and share of sum_scores (to HBM: team_merge_kernel),
            //      reset of the rows
            if constexpr (kWide) {
                // On a database built from reference sequences a read's lists fall into one slice of the four
                // (bench.py --clades): the other three items are done here.  (The lean build says so to its dense
                // epilogue instead, args.untouched: a second way out of the loop here costs it 13 registers -- 109
                // instead of the 96 of five waves per SIMD.)
                if (slice_untouched) {
                    publish_empty_slice(ctx, (uint32_t)n_kmers, k, p.log_threshold, keep);
                    continue;
                }
            }
            if (lane == 0 && !slice_untouched) lds.store(rows_pad - 1u, 0u, 0u);  // the dummy row of the out-of-range lanes
#ifdef EPIK_AMD_ABLATION
            if (p.ablate & 2u) {  // (timing experiments: no slice epilogue)
                lds.clear(rows_pad);
                continue;
            }
#endif
            SliceArgs args;
            args.rows_pad = rows_pad;
            args.rows = ctx.rows_;
            args.base = ctx.base_;
            args.kmer_size = k;
            args.keep = keep;
            args.log_threshold = p.log_threshold;
            args.slice_at = (uint32_t)slice_at;  // (the host keeps reads * slices below 2^32 for this kernel)
            args.trace_at = 0xffffffffu;
            args.untouched = slice_untouched ? 1u : 0u;
#ifdef EPIK_AMD_ABLATION
            if (traced) args.trace_at = trace_at, trace_at += 10;  // the epilogue's entries
#endif
            if constexpr (kSparseCounts) {
                // (what was streamed says how many rows may hold a count: with few, the touched quads are listed)
                if (sparse_hint <= tp.sparse_chunks &&
                    slice_epilogue_sparse<W, CountT>(reinterpret_cast<const TeamParams *>(kp), lds, (uint32_t)n_kmers, args, tp.sparse_quads)) {
                    STREAM_STAMP(3)
                    continue;
                }
            }
            slice_epilogue<W, CountT>(reinterpret_cast<const TeamParams *>(kp), lds, (uint32_t)n_kmers, args);
            STREAM_STAMP(3)  // slice epilogue
        }
    }
}

// Partial lists: room for every (read, slice) list of a part, one after the other in read order, from the front
// kernel's bounds (a part = the reads of one finisher): an exclusive prefix sum per part, in two kernels -- the
// sum of every tile of kScanTile items, then, per tile, the tiles in front of it added up and the tile scanned.
constexpr uint32_t kScanTile = 4096, kScanThreads = 256;  // 16 items per thread
__device__ __forceinline__ uint64_t scan_block_sum(uint64_t v, unsigned long long *wave_sums /*[kScanThreads / 64]*/)
{
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_u64(v, m);
    if ((threadIdx.x & 63u) == 0) wave_sums[threadIdx.x >> 6] = v;
    __syncthreads();
    uint64_t total = 0;
    for (uint32_t w = 0; w < kScanThreads / 64u; ++w) total += wave_sums[w];
    __syncthreads();
    return total;
}
// grid (tiles per part, parts)
__global__ __launch_bounds__(kScanThreads) void team_sparse_sums_kernel(TeamParams tp, uint32_t n_slices, unsigned long long *tile_sums)
{
    __shared__ unsigned long long wave_sums[kScanThreads / 64];
    const PlaceParams &p = tp.base;
    const uint64_t r0 = (uint64_t)blockIdx.y * tp.sparse_part_reads;
    const uint64_t r1 = r0 + tp.sparse_part_reads < p.n_reads ? r0 + tp.sparse_part_reads : p.n_reads;
    const uint64_t items = r1 > r0 ? (r1 - r0) * n_slices : 0;
    const uint32_t *__restrict__ room = tp.sparse_cap + r0 * n_slices;
    const uint64_t t0 = (uint64_t)blockIdx.x * kScanTile;
    uint64_t mine = 0;
    for (uint32_t u = 0; u < kScanTile / kScanThreads; ++u) {
        const uint64_t i = t0 + (uint64_t)u * kScanThreads + threadIdx.x;  // (coalesced: the order does not matter for a sum)
        if (i < items) mine += room[i];
    }
    const uint64_t total = scan_block_sum(mine, wave_sums);
    if (threadIdx.x == 0) tile_sums[(uint64_t)blockIdx.y * gridDim.x + blockIdx.x] = total;
}
__global__ __launch_bounds__(kScanThreads) void team_sparse_scan_kernel(TeamParams tp, uint32_t n_slices, const unsigned long long *tile_sums)
{
    __shared__ unsigned long long wave_sums[kScanThreads / 64];
    const PlaceParams &p = tp.base;
    const uint32_t lane = (uint32_t)lane_id(), wave = threadIdx.x >> 6;
    const uint64_t r0 = (uint64_t)blockIdx.y * tp.sparse_part_reads;
    const uint64_t r1 = r0 + tp.sparse_part_reads < p.n_reads ? r0 + tp.sparse_part_reads : p.n_reads;
    const uint64_t items = r1 > r0 ? (r1 - r0) * n_slices : 0;
    const uint32_t *__restrict__ room = tp.sparse_cap + r0 * n_slices;
    uint2 *__restrict__ index = tp.sparse_index + r0 * n_slices;
    // the tiles of this part in front of this one (the first tile's workgroup adds up all of them: the part's size)
    const unsigned long long *sums = tile_sums + (uint64_t)blockIdx.y * gridDim.x;
    const uint32_t n_before = blockIdx.x == 0 ? gridDim.x : blockIdx.x;
    uint64_t mine = 0;
    for (uint32_t t = threadIdx.x; t < n_before; t += kScanThreads) mine += sums[t];
    const uint64_t before = scan_block_sum(mine, wave_sums);
    if (blockIdx.x == 0 && threadIdx.x == 0) tp.sparse_part_total[blockIdx.y] = before;
    // this tile: 16 consecutive items per thread
    constexpr uint32_t kPer = kScanTile / kScanThreads;
    const uint64_t i0 = (uint64_t)blockIdx.x * kScanTile + (uint64_t)kPer * threadIdx.x;
    uint32_t v[kPer];
    uint32_t own = 0;
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) {
        v[u] = i0 + u < items ? room[i0 + u] : 0u;
        own += v[u];  // (a list has at most the slice's rows, a tile 4096 lists: no wrap)
    }
    const uint32_t incl = wave_incl_scan_u32(own);
    if (lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    uint64_t at = (blockIdx.x == 0 ? 0ull : before) + (incl - own);
    for (uint32_t w = 0; w < wave; ++w) at += wave_sums[w];
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) {
        // (offsets inside a part are 32-bit: the host keeps a part's entries below 2^32, see capi.hip)
        if (i0 + u < items) index[i0 + u] = make_uint2((uint32_t)at, 0u);
        at += v[u];
    }
}

// The headers of a batch whose descriptors nobody needs (finish): length and the two "no placement" flags.
__global__ __launch_bounds__(256) void team_header_kernel(TeamParams tp, uint32_t n_slices, uint64_t max_kmers)
{
    const PlaceParams &p = tp.base;
    const uint64_t read = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (read >= p.n_reads) return;
    const uint64_t len = p.seq_offsets[read + 1] - p.seq_offsets[read];
    uint32_t *hdr = reinterpret_cast<uint32_t *>(tp.front_hdr + read * tp.front_hdr_stride);
    hdr[0] = 0u;
    hdr[1] = len < p.kmer_size ? kFrontNoRows : len - p.kmer_size + 1 > max_kmers ? kFrontTooNarrow : 0u;
    hdr[2] = (uint32_t)len;
    for (uint32_t s2 = 0; s2 < n_slices; ++s2) hdr[kFrontHdrWords + s2] = 0u;
}

// ---------------------------------------------------------------------------------
// The merge: one wave per read ranks the slices' rows together and finishes the placement (team_merge:
// sum_scores :164-184 from the partial sums, like-weight-ratios :241-264, filter_by_ratio :188-199, rows out).
// In the streaming kernel, by one of the read's four waves, it was a fifth of the read's time.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void team_merge_kernel(TeamParams tp, uint32_t n_slices)
{
    const PlaceParams &p = tp.base;
    const int lane = lane_id();
    const uint64_t waves_per_block = blockDim.x >> 6;
    const uint64_t n_waves = (uint64_t)gridDim.x * waves_per_block;
    MergeParams mp;
    mp.keep_at_most = p.keep_at_most;
    mp.kmer_size = p.kmer_size;
    mp.num_branches = p.num_branches;
    mp.log_threshold = p.log_threshold;
    mp.keep_factor = p.keep_factor;
    mp.rows = p.rows;
    mp.n_rows = p.n_rows;
    mp.kmer_counts = p.kmer_counts;
    v4u *rows_out = static_cast<v4u *>(tp.slice_rows_out);  // (more than 64 slots: team_merge keeps the ranks in them)
    TeamPartial *sums_out = static_cast<TeamPartial *>(tp.slice_sums_out);
    const uint64_t first = (uint64_t)blockIdx.x * waves_per_block + (threadIdx.x >> 6);
    auto settled = [&](uint64_t read, uint32_t flags) {  // reads without rows to merge
        if (flags & kFrontSlow) return true;  // team_place_kernel's
        if (flags & (kFrontNoRows | kFrontTooNarrow)) {
            if (lane == 0) p.n_rows[read] = (flags & kFrontNoRows) ? 0u : kCountsTooNarrow;
            return true;
        }
        return false;
    };
    if (n_slices * p.keep_at_most <= (uint32_t)kWave) {
        // A slot per lane.  A read is a chain of three memory round trips (header, slots and sums, rows out) around
        // a few hundred instructions: the wave asks for the next read's header, slots and sums before it works on
        // this one's.  (Slots and sums of a read that has none to merge are whatever the buffers hold: not used.)
        struct Fetched {
            uint32_t flags, len;
            MergeInputs in;
        };
        auto fetch = [&](uint64_t read) {
            const uint32_t *hdr = reinterpret_cast<const uint32_t *>(tp.front_hdr + read * tp.front_hdr_stride);
            Fetched f;
            f.flags = hdr[1], f.len = hdr[2];
            f.in = load_merge_inputs(rows_out + read * n_slices * p.keep_at_most, p.keep_at_most, sums_out + read * n_slices,
                                     n_slices, p.keep_at_most);
            return f;
        };
        if (first >= p.n_reads) return;
        Fetched cur = fetch(first);
        for (uint64_t read = first; read < p.n_reads; read += n_waves) {
            const uint64_t next_read = read + n_waves < p.n_reads ? read + n_waves : read;  // (the last one: once more, unused)
            const Fetched next = fetch(next_read);
            if (!settled(read, cur.flags))
                team_merge_body<true>(mp, rows_out, p.keep_at_most, sums_out, n_slices, read, (uint64_t)cur.len - p.kmer_size + 1u,
                                      cur.in);
            cur = next;
        }
        return;
    }
    for (uint64_t read = first; read < p.n_reads; read += n_waves) {
        const uint32_t *hdr = reinterpret_cast<const uint32_t *>(tp.front_hdr + read * tp.front_hdr_stride);
        const uint32_t flags = hdr[1], len = hdr[2];
        if (settled(read, flags)) continue;
        team_merge(mp, rows_out + read * n_slices * p.keep_at_most, p.keep_at_most, sums_out + read * n_slices, n_slices,
                   read, (uint64_t)len - p.kmer_size + 1u);
    }
}

// The merge with 4 or 2 reads to a wave (team_device.hpp: team_merge_packed): a group of L lanes per read.
template <int L>
__global__ __launch_bounds__(256) void team_merge_packed_kernel(TeamParams tp, uint32_t n_slices)
{
    const PlaceParams &p = tp.base;
    constexpr uint32_t kPerWave = 64u / (uint32_t)L;
    const uint64_t waves_per_block = blockDim.x >> 6;
    const uint64_t n_waves = (uint64_t)gridDim.x * waves_per_block;
    MergeParams mp;
    mp.keep_at_most = p.keep_at_most;
    mp.kmer_size = p.kmer_size;
    mp.num_branches = p.num_branches;
    mp.log_threshold = p.log_threshold;
    mp.keep_factor = p.keep_factor;
    mp.rows = p.rows;
    mp.n_rows = p.n_rows;
    mp.kmer_counts = p.kmer_counts;
    const v4u *rows_out = static_cast<const v4u *>(tp.slice_rows_out);
    const TeamPartial *sums_out = static_cast<const TeamPartial *>(tp.slice_sums_out);
    const uint32_t group = (uint32_t)lane_id() / (uint32_t)L;
    const uint64_t first = ((uint64_t)blockIdx.x * waves_per_block + (threadIdx.x >> 6)) * kPerWave + group;
    const uint64_t stride = n_waves * kPerWave;
    if (first - group >= p.n_reads) return;  // (wave-uniform: the wave's first read)
    // (a read is a chain of memory round trips around a few hundred instructions: the next reads' inputs are asked for
    // before this wave works on the ones it has)
    PackedMergeInputs cur = load_packed_merge_inputs<L>(tp, rows_out, sums_out, first, first < p.n_reads, n_slices, p.keep_at_most);
    for (uint64_t read = first; read - group < p.n_reads; read += stride) {
        const uint64_t next_read = read + stride;
        const PackedMergeInputs next = load_packed_merge_inputs<L>(tp, rows_out, sums_out, next_read, next_read < p.n_reads, n_slices, p.keep_at_most);
        team_merge_packed<L>(mp, read, read < p.n_reads, cur);
        cur = next;
    }
}

namespace {

template <typename F>
hipError_t stream_dispatch(int waves, int counts, int mode, bool wide, int bw, F &&f)
{
#define EPIK_STREAM_CASE(W, C, M, WIDE, BW) \
    if (waves == W && counts == C && mode == M && wide == WIDE && bw == BW) \
        return f.template operator()<W, std::conditional_t<C == kCounts8, uint8_t, std::conditional_t<C == kCounts16, uint16_t, uint32_t>>, M, WIDE, BW>();
#define EPIK_STREAM_MODES(W, C, WIDE, BW) EPIK_STREAM_CASE(W, C, kTeamModePlace, WIDE, BW) EPIK_STREAM_CASE(W, C, kTeamModeAccumulate, WIDE, BW) \
    EPIK_STREAM_CASE(W, C, kTeamModeFinish, WIDE, BW) EPIK_STREAM_CASE(W, C, kTeamModeAccumulateLists, WIDE, BW) EPIK_STREAM_CASE(W, C, kTeamModeFinishLists, WIDE, BW)
#define EPIK_STREAM_COUNTS(W, WIDE, BW) EPIK_STREAM_MODES(W, kCounts8, WIDE, BW) EPIK_STREAM_MODES(W, kCounts16, WIDE, BW) EPIK_STREAM_MODES(W, kCounts32, WIDE, BW)
    EPIK_STREAM_COUNTS(2, false, 4) EPIK_STREAM_COUNTS(2, true, 4) EPIK_STREAM_COUNTS(4, false, 4) EPIK_STREAM_COUNTS(4, true, 4) EPIK_STREAM_COUNTS(8, false, 4)
    // (two-wave workgroups: the one-pass placement only -- the halves of a sharded placement keep four)
    if (mode == kTeamModePlace) {
#define EPIK_STREAM_PLACE(W, WIDE) EPIK_STREAM_CASE(W, kCounts8, kTeamModePlace, WIDE, 2) EPIK_STREAM_CASE(W, kCounts16, kTeamModePlace, WIDE, 2) \
    EPIK_STREAM_CASE(W, kCounts32, kTeamModePlace, WIDE, 2)
        EPIK_STREAM_PLACE(2, false) EPIK_STREAM_PLACE(2, true) EPIK_STREAM_PLACE(4, false) EPIK_STREAM_PLACE(4, true) EPIK_STREAM_PLACE(8, false)
#undef EPIK_STREAM_PLACE
    }
#undef EPIK_STREAM_COUNTS
#undef EPIK_STREAM_MODES
#undef EPIK_STREAM_CASE
    return hipErrorInvalidValue;
}

uint64_t max_kmers_of(int counts)
{
    return counts == kCounts8 ? WaveLds<uint8_t>::kMaxKmers : counts == kCounts16 ? WaveLds<uint16_t>::kMaxKmers
                                                                                : WaveLds<uint32_t>::kMaxKmers;
}

}  // namespace

namespace {
// LDS of the front kernel: the table entries of a short read's tiles, for every pass -- up to 24 KB per
// workgroup of one wave; trees of more passes than that take the two-sweep path.
uint32_t front_held_passes(int waves, uint32_t passes, size_t *lds)
{
    const uint32_t per_pass = (uint32_t)kTilesPerPass * (uint32_t)team_entry_bytes(waves) * 64u;
    const uint32_t held_passes = std::max(1u, std::min(passes, 24576u / per_pass));
    *lds = (size_t)held_passes * per_pass;
    return held_passes;
}
}  // namespace

hipError_t launch_team_front(const TeamParams &tp, int waves, int counts, bool lists, dim3 grid, hipStream_t stream)
{
    // reads with more k-mers than the consumer's counts hold get no descriptors (it marks them)
    size_t lds = 0;
    const uint32_t held_passes = front_held_passes(waves, tp.passes, &lds);
    uint64_t max_kmers = max_kmers_of(counts);
    if (tp.base.max_kmers_cap) max_kmers = std::min<uint64_t>(max_kmers, tp.base.max_kmers_cap);
    if (waves == 2 && !lists)
        hipLaunchKernelGGL((team_front_kernel<2, false>), grid, dim3(64), lds, stream, tp, max_kmers, held_passes);
    else if (waves == 2)
        hipLaunchKernelGGL((team_front_kernel<2, true>), grid, dim3(64), lds, stream, tp, max_kmers, held_passes);
    else if (waves == 4 && !lists)
        hipLaunchKernelGGL((team_front_kernel<4, false>), grid, dim3(64), lds, stream, tp, max_kmers, held_passes);
    else if (waves == 4)
        hipLaunchKernelGGL((team_front_kernel<4, true>), grid, dim3(64), lds, stream, tp, max_kmers, held_passes);
    else if (waves == 8 && !lists)
        hipLaunchKernelGGL((team_front_kernel<8, false>), grid, dim3(64), lds, stream, tp, max_kmers, held_passes);
    else if (waves == 8)
        hipLaunchKernelGGL((team_front_kernel<8, true>), grid, dim3(64), lds, stream, tp, max_kmers, held_passes);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

uint64_t sparse_scan_tiles(uint64_t part_reads, uint32_t slices) { return std::max<uint64_t>(1, (part_reads * slices + kScanTile - 1) / kScanTile); }

hipError_t launch_team_sparse_scan(const TeamParams &tp, int waves, unsigned long long *tile_sums, hipStream_t stream)
{
    const uint32_t slices = (uint32_t)waves * tp.passes;
    const dim3 grid((unsigned)sparse_scan_tiles(tp.sparse_part_reads, slices), tp.sparse_parts);
    hipLaunchKernelGGL(team_sparse_sums_kernel, grid, dim3(kScanThreads), 0, stream, tp, slices, tile_sums);
    hipLaunchKernelGGL(team_sparse_scan_kernel, grid, dim3(kScanThreads), 0, stream, tp, slices, tile_sums);
    return hipGetLastError();
}

hipError_t launch_team_headers(const TeamParams &tp, int waves, int counts, hipStream_t stream)
{
    const dim3 grid((unsigned)((tp.base.n_reads + 255) / 256));
    uint64_t max_kmers = max_kmers_of(counts);
    if (tp.base.max_kmers_cap) max_kmers = std::min<uint64_t>(max_kmers, tp.base.max_kmers_cap);
    hipLaunchKernelGGL(team_header_kernel, grid, dim3(256), 0, stream, tp, (uint32_t)waves * tp.passes, max_kmers);
    return hipGetLastError();
}

// Which build of the streaming kernel a geometry gets: db_layout.h's rule, or -- EPIK_AMD_STREAM_WIDE=1 / 0, tests --
// the wide build whatever the slices' size (2 and 4 slices per pass have one) / the lean one.  The wide build holds
// the slice epilogue over the touched quads (team_epilogue.hpp), which the small trees of the parity tests would
// otherwise never reach.
bool team_stream_is_wide(int waves, size_t lds_bytes, int bw)
{
    // (read at every call: a handful per launch, and the tests create placers of several kinds in one process)
    const char *e = std::getenv("EPIK_AMD_STREAM_WIDE");
    const int forced = e && e[0] == '1' ? 1 : e && e[0] == '0' ? 0 : -1;
    if (forced == 1) return waves <= 4;
    if (forced == 0) return false;
    return stream_wide(waves, lds_bytes, bw);
}

hipError_t launch_team_stream(const TeamParams &tp, int waves, int counts, int mode, int bw, dim3 grid, size_t lds_bytes,
                              hipStream_t stream, const SparseSources *sources)
{
    const SparseSources src = sources ? *sources : SparseSources{};
    return stream_dispatch(waves, counts, mode, team_stream_is_wide(waves, lds_bytes, bw), bw, [&]<int W, typename C, int M, bool kWide, int kBW>() {
        hipLaunchKernelGGL((team_stream_kernel<W, C, M, kWide, kBW>), grid, dim3(kBW * 64), lds_bytes, stream, tp, src);
        return hipGetLastError();
    });
}

hipError_t launch_team_merge(const TeamParams &tp, int waves, dim3 grid, hipStream_t stream)
{
    // (grid: workgroups of four waves sized for a read per wave; with several reads to a wave fewer do)
    const uint32_t n_slices = (uint32_t)waves * tp.passes, slots = n_slices * tp.base.keep_at_most;
    const char *forced = std::getenv("EPIK_AMD_MERGE_PACKED");  // 0: a wave per read (tests, measurements)
    const bool packed = !(forced && forced[0] == '0');
    if (packed && slots <= 16u)
        hipLaunchKernelGGL(team_merge_packed_kernel<16>, dim3((grid.x + 3u) / 4u), dim3(256), 0, stream, tp, n_slices);
    else if (packed && slots <= 32u)
        hipLaunchKernelGGL(team_merge_packed_kernel<32>, dim3((grid.x + 1u) / 2u), dim3(256), 0, stream, tp, n_slices);
    else
        hipLaunchKernelGGL(team_merge_kernel, grid, dim3(256), 0, stream, tp, n_slices);
    return hipGetLastError();
}

hipError_t set_team_stream_lds_limit(int waves, int counts, int mode, int bw, size_t lds_bytes)  // (always the whole CU: see place_kernel.hip)
{
    return stream_dispatch(waves, counts, mode, team_stream_is_wide(waves, lds_bytes, bw), bw, [&]<int W, typename C, int M, bool kWide, int kBW>() {
            return hipFuncSetAttribute(reinterpret_cast<const void *>(&team_stream_kernel<W, C, M, kWide, kBW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu);
        });
}

hipError_t team_stream_occupancy(int waves, int counts, int mode, int bw, size_t lds_bytes, int *blocks_per_cu)
{
    return stream_dispatch(waves, counts, mode, team_stream_is_wide(waves, lds_bytes, bw), bw, [&]<int W, typename C, int M, bool kWide, int kBW>() {
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, team_stream_kernel<W, C, M, kWide, kBW>, kBW * 64,
                                                            lds_bytes);
    });
}

}  // namespace epik_amd
