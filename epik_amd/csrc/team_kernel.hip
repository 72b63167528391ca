// team_kernel.hip -- large trees: ONE WORKGROUP PLACES ONE READ, the branch range split over
// its wavefronts.
//
// Reference path: the same as place_kernel.hip (epik/src/epik/place.cpp:278-440, 134-199, 241-267).
// place.cpp:92-96 sizes the per-thread score and count vectors by tree.get_node_count() without any
// bound.  With one wavefront per read (place_kernel.hip) a wave owns a whole vector in LDS: at
// N ~ 10 000 branches three waves fit a CU, and beyond ~20 000 none does.  Here the W waves of a
// workgroup place ONE read together: wave w owns the rows of slice w of the branch range, and the
// database is stored pre-split the same way -- every posting list as W sublists, one per slice,
// in the list's original order (db_image.cpp).  A branch still receives its float32 adds in
// exactly the k-mer order of place.cpp:349-371, because every wave walks the read's k-mers in
// order and the postings of one list are distinct branches: bit-identical sums, as before.
// Trees too large even for that are placed in P passes over S = W * P slices (pass p = slices
// p*W .. p*W + W-1, a table of its own), so there is no upper bound on the tree.
//
//   front end   wave t encodes tile t of a group of W tiles, looks the k-mers up (one table entry
//               {line, len[W]} gives all W sublists) and, after the tile totals have crossed LDS,
//               writes the chunk descriptors of every slice into that slice's list, in read order;
//   stream      wave w runs its own list through the ring of place_device.hpp into its rows;
//   ambiguous   (cold) every wave sweeps the read's ambiguous keys for its slice;
//   epilogue    every wave: correction, its slice's best rows and share of sum_scores -> LDS;
//   merge       the last wave (idle in the front end of short reads): the slices' rows ranked
//               together, sum_scores, like-weight-ratios, filter, rows out -- while the other
//               waves already encode the next read.
//
// Gather / scatter-add, no MFMA.  Unlike the one-wavefront kernel this one is not bound by bandwidth but by
// the time a workgroup takes per read (LDS leaves room for three reads per CU at N = 9 999) and by vector
// instruction issue.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "team_device.hpp"

namespace epik_amd {

enum : int { kTeamPlace = kTeamModePlace, kTeamAccumulate = kTeamModeAccumulate, kTeamFinish = kTeamModeFinish,
             kTeamAccumulateLists = kTeamModeAccumulateLists };  // (lists: the reads the front kernel left to this kernel)

template <int W, typename CountT, int kMode>
__global__ __launch_bounds__(W * 64, team_waves_per_simd(W)) void team_place_kernel(TeamParams tp)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    typedef WaveLds<CountT> Lds;
    const PlaceParams &p = tp.base;
    const int lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t rows_pad = tp.rows_pad;
    // ---- LDS: the W slices' rows | the W descriptor lists | tile totals + flags | partial sums |
    //      (more than one pass only) the slices' ranked rows.  With one pass a slice's ranked rows
    //      lie at the start of its wave's descriptor list, idle by then.
    Lds lds;
    unsigned char *desc_base = lds_raw + (size_t)W * tp.slice_bytes;
    lds.score = (typename Lds::f32_t *)reinterpret_cast<float *>(lds_raw + (size_t)wave * tp.slice_bytes);
    lds.count = (typename Lds::count_t *)reinterpret_cast<CountT *>(lds_raw + (size_t)wave * tp.slice_bytes + (size_t)rows_pad * 4);
    lds.desc = (typename Lds::u64_t *)reinterpret_cast<uint64_t *>(desc_base + (size_t)wave * tp.desc_bytes);
    lds_u32 *totals = (lds_u32 *)reinterpret_cast<uint32_t *>(desc_base + (size_t)W * tp.desc_bytes);  // [W tiles][W slices]
    lds_u32 *flags = totals + W * W;                                                                      // [2]: any ambiguous k-mer, alternating over the workgroup's reads
    lds_partial *partials = (lds_partial *)reinterpret_cast<TeamPartial *>(reinterpret_cast<uint32_t *>(desc_base + (size_t)W * tp.desc_bytes) + W * W + 4);
    const uint32_t n_slices = W * tp.passes;
    // the slices' ranked rows for the merge: an area of their own (not the idle descriptor lists: the tile
    // waves fill those for the next read while the last wave still merges this one)
    lds_u32x4 *merge_cand = (lds_u32x4 *)reinterpret_cast<v4u *>(reinterpret_cast<TeamPartial *>(reinterpret_cast<uint32_t *>(desc_base + (size_t)W * tp.desc_bytes) + W * W + 4) + n_slices);
    const uint32_t merge_stride = p.keep_at_most;  // in entries of 16 bytes, from one slice's rows to the next
    lds_u32 *tiles_done = flags + 2;  // how many tiles of this workgroup have published their totals so far
    uint32_t tiles_expected = 0;      // ... and how many must have before this group's descriptors can be laid out
    const uint32_t score_top = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds.score + (rows_pad - 1u) * 4u);
    const uint32_t count_top = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds.count +
                                                              (rows_pad - 1u) * (uint32_t)sizeof(CountT));
    const PlaceParams *kp = (const PlaceParams *)__builtin_amdgcn_kernarg_segment_ptr();  // = &tp.base
    for (uint32_t i = lane; i < rows_pad; i += kWave) lds.store(i, 0u, 0u);
    uint64_t part_first = 0;  // partial lists out: where every part's entries begin (lane r: part r), as in team_stream_kernel
    if constexpr (kMode == kTeamAccumulateLists) {
        const uint64_t mine = (uint32_t)lane < tp.sparse_parts ? (uint64_t)tp.sparse_part_total[lane] : 0ull;
        for (uint32_t r = 0; r < tp.sparse_parts; ++r) {
            const uint64_t t = readlane_u64(mine, (int)r);
            if ((uint32_t)lane > r) part_first += t;
        }
    }
    // (no list entry is ever read before it was written; should that ever break, an entry is at least a chunk
    // of zero bytes at a valid address and not what the LDS happened to hold)
    for (uint32_t i = lane; i < tp.desc_bytes / 8u; i += kWave) lds.desc[i] = null_chunk(p);
    if (threadIdx.x < 4) flags[threadIdx.x] = 0u;  // the two flags, the tile counter, a spare
    __syncthreads();

    const uint32_t k = p.kmer_size;
    const uint32_t sigma = p.alphabet_size;
    const uint32_t stride = kWave - (k - 1);  // windows per 64-character tile
    const uint32_t cap = tp.desc_cap;
#ifdef EPIK_AMD_ABLATION
    // where the waves of a team spend their time: cycles per (wave, phase), EPIK_AMD_STAMPS=1
    unsigned long long dbg_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long dbg_last = p.dbg ? __builtin_amdgcn_s_memtime() : 0;
#define TEAM_STAMP(k)                                                    \
    if (p.dbg) {                                                         \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();    \
        dbg_t[k] += now_ - dbg_last;                                     \
        dbg_last = now_;                                                 \
    }
#else
#define TEAM_STAMP(k)
#endif

    // the bounds of a read are loaded one read ahead: they arrive under the read in front of them
    // (a launch over a list of reads -- the ones team_front_kernel left to this kernel -- takes entry `at`
    // of the list where the others take read `at`)
    const uint64_t n_todo = tp.read_list ? (uint64_t)*tp.read_list_count : p.n_reads;
    uint64_t bounds[2] = {0, 0};
    uint64_t next_read = 0;
    if (blockIdx.x < n_todo) {
        next_read = tp.read_list ? tp.read_list[blockIdx.x] : (uint64_t)blockIdx.x;
        bounds[0] = p.seq_offsets[next_read], bounds[1] = p.seq_offsets[next_read + 1];
    }
    uint32_t placed_here = 0;  // reads this workgroup has really placed (the skipped ones below meet at no barrier)
    for (uint64_t at = blockIdx.x; at < n_todo; at += gridDim.x) {
        const uint64_t read = readlane_u64(next_read, 0);
        const uint64_t seq_begin = readlane_u64(bounds[0], 0);
        const uint64_t len = readlane_u64(bounds[1], 0) - seq_begin;
        const uint8_t *__restrict__ seq = p.seqs + seq_begin;
        if (at + gridDim.x < n_todo) {
            next_read = tp.read_list ? tp.read_list[at + gridDim.x] : at + gridDim.x;
            bounds[0] = p.seq_offsets[next_read], bounds[1] = p.seq_offsets[next_read + 1];
        }
        // place.cpp:322 underflows for len < k; we report "no placement".  A read with more k-mers than
        // this launch's counts hold is marked (the caller chose the count width).  Uniform over the
        // workgroup: nobody is left waiting at a barrier.
        if (len < k || len - k + 1 > Lds::kMaxKmers || (p.max_kmers_cap && len - k + 1 > p.max_kmers_cap)) {
            if (kMode == kTeamAccumulateLists) {
                // (its lists are empty: the front kernel made no room, the scan kernel wrote "0 entries")
            } else if (kMode == kTeamAccumulate) {  // an all-zero partial vector
                for (uint32_t i = threadIdx.x; i < p.num_branches; i += W * kWave) {
                    p.partial_scores[read * p.num_branches + i] = 0.0f;
                    p.partial_counts[read * p.num_branches + i] = 0u;
                }
            } else if (threadIdx.x == 0) {
                p.n_rows[read] = len < k ? 0u : kCountsTooNarrow;
            }
            continue;
        }
        const uint64_t n_kmers = len - k + 1;  // :322
        // which of the two flags this read uses: the reads the workgroup PLACES alternate (NOT read & 1: with an
        // even grid every read of a workgroup has the same parity; and not the iteration index either: a read
        // skipped above executes no barrier, so the reads before and after it would share a flag and wave 0's
        // clear of the first could land on the second's set)
        const uint32_t parity = placed_here++ & 1u;
        bool any_amb = false;  // the same in every wave of the workgroup

        for (uint32_t pass = 0; pass < tp.passes; ++pass) {
            TeamCtx<W> ctx;
            ctx.rows_pad_ = rows_pad;
            ctx.kmer_size_ = k;
            ctx.keep_ = p.keep_at_most;
            ctx.log_threshold_ = p.log_threshold;
            ctx.slice_ = wave;
            ctx.pass_ = pass;
            ctx.base_ = (pass * W + wave) * tp.slice_rows;
            ctx.rows_ = ctx.base_ >= p.num_branches ? 0u : min(tp.slice_rows, p.num_branches - ctx.base_);
            ctx.cand = merge_cand + (size_t)(pass * W + wave) * merge_stride;
            ctx.partial = partials + (pass * W + wave);

            if (kMode == kTeamFinish) {
                // second half of a k-mer-space-sharded placement: the slice's totals come back from HBM
                // (see finish_reads_kernel in place_kernel.hip for the ambiguous record)
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
            } else {
                // ---- exact k-mers, read order (place.cpp:294-305, 349-371) ------------------------
                for (uint64_t group_pos = 0; group_pos < n_kmers; group_pos += (uint64_t)W * stride) {
                    const uint64_t tile_pos = group_pos + (uint64_t)wave * stride;
                    const bool has_tile = tile_pos < n_kmers;  // wave-uniform
                    uint64_t start[W];
                    uint32_t llen[W], nch[W], first[W], tile_total[W];
#pragma unroll
                    for (int s = 0; s < W; ++s) {
                        start[s] = 0;
                        llen[s] = nch[s] = first[s] = tile_total[s] = 0;
                    }
                    if (has_tile) {
                        const Tile tl = encode_tile(seq, len, tile_pos, n_kmers, k, sigma, stride, p.char_class);
                        bool exact = tl.in_range;
                        if ((tl.inv_mask | tl.amb_mask) != 0) {  // wave-uniform, cold
                            const uint64_t wmask = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
                            const uint64_t inv_w = (tl.inv_mask >> lane) & wmask;
                            const uint64_t amb_w = (tl.amb_mask >> lane) & wmask;
                            const bool is_amb = tl.in_range && inv_w == 0 && __popcll(amb_w) == 1;
                            exact = tl.in_range && inv_w == 0 && amb_w == 0;
                            if (__ballot(is_amb) != 0 && lane == 0) flags[parity] = 1u;
                        }
                        if (exact) {
                            TeamEntry<W> e;
                            e.load(tp, pass, tl.key, (uint32_t)tile_pos + (uint32_t)lane);
#pragma unroll
                            for (int s = 0; s < W; ++s) {
                                llen[s] = e.len[s];
                                start[s] = e.start(s);
                            }
                        }
#pragma unroll
                        for (int s = 0; s < W; ++s) {
                            nch[s] = (llen[s] + (uint32_t)kWave - 1u) >> 6;
                            const uint32_t incl = wave_incl_scan_u32(nch[s]);
                            first[s] = incl - nch[s];
                            tile_total[s] = __builtin_amdgcn_readlane(incl, 63);
                        }
                    }
                    TEAM_STAMP(0)  // encode, lookup, scans
                    {   // chunks per (tile, slice) across the workgroup
                        uint32_t mine = tile_total[0];
#pragma unroll
                        for (int s = 1; s < W; ++s) mine = (lane == s) ? tile_total[s] : mine;
                        if (has_tile && lane < W) totals[wave * W + (uint32_t)lane] = mine;
                    }
                    uint32_t n_tiles;  // tiles of this group: the rows of the table that were published for it
                    // Not a barrier: only the waves that HAVE a tile publish, and nobody waits for one that has
                    // none -- the last wave, still merging the previous read while the others encode this one,
                    // finds the counter already there when it arrives.  (Every wave counts the same tiles: the
                    // counter only grows, and the two real barriers of the round keep a fast wave from
                    // publishing the next group before everybody has read this one's totals.)
                    {
                        const uint64_t tiles_left = (n_kmers - group_pos + stride - 1) / stride;
                        n_tiles = tiles_left < (uint64_t)W ? (uint32_t)tiles_left : (uint32_t)W;
                        tiles_expected += n_tiles;
                        if (has_tile && lane == 0)
                            __hip_atomic_fetch_add((uint32_t *)tiles_done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        while ((int32_t)(__hip_atomic_load((uint32_t *)tiles_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) -
                                         tiles_expected) < 0)
                            __builtin_amdgcn_s_sleep(1);
                    }
                    TEAM_STAMP(1)  // waiting for the other tiles
                    any_amb = any_amb || flags[parity] != 0u;  // set before the publication; cleared two barriers later at the earliest
                    uint32_t base[W];       // chunks of the earlier tiles of this group, per slice
                    uint32_t my_total = 0;  // chunks of this wave's slice in the group
                    uint32_t max_total = 0;
                    {   // the W x W table comes out of LDS once, one entry per lane; the sums are scalar work
                        constexpr int kRegs = (W * W + kWave - 1) / kWave;
                        uint32_t held[kRegs];
#pragma unroll
                        for (int q = 0; q < kRegs; ++q)
                            held[q] = (q * kWave + lane < W * W) ? totals[q * kWave + lane] : 0u;
#pragma unroll
                        for (int s = 0; s < W; ++s) {
                            uint32_t sum = 0;
                            base[s] = 0;
#pragma unroll
                            for (int t = 0; t < W; ++t) {
                                // (a wave without a tile published nothing: its row is whatever an earlier group left)
                                const uint32_t v = (uint32_t)t < n_tiles ? __builtin_amdgcn_readlane(held[(t * W + s) / kWave], (t * W + s) % kWave) : 0u;
                                base[s] += (uint32_t)t < wave ? v : 0u;
                                sum += v;
                            }
                            my_total = (uint32_t)s == wave ? sum : my_total;
                            max_total = max(max_total, sum);
                        }
                    }
                    // rounds of at most `cap` chunks per slice (one, unless a list is very long)
                    for (uint32_t w0 = 0; w0 < max_total || w0 == 0; w0 += cap) {
                        if (has_tile) {
                            // Every lane writes the first kOwnChunks chunks of its own sublists; the rest of
                            // a longer sublist is written by the whole wave, lane j writing chunk kOwnChunks + j.
                            constexpr uint32_t kOwnChunks = 3;
#pragma unroll
                            for (int s = 0; s < W; ++s) {
                                auto *list = (typename Lds::u64_t *)reinterpret_cast<uint64_t *>(desc_base + (size_t)s * tp.desc_bytes);
                                const uint32_t at = base[s] + first[s];
#pragma unroll
                                for (uint32_t c = 0; c < kOwnChunks; ++c) {
                                    const uint32_t idx = at + c - w0;  // wraps when in front of the window
                                    if (nch[s] > c && idx < cap) {
                                        const uint32_t rest = llen[s] - (c << 6);
                                        const uint64_t cnt = rest < (uint32_t)kWave ? rest : (uint32_t)kWave;
                                        list[idx] = chunk_address<TeamChunks>(p, start[s], c) | (cnt << 48);
                                    }
                                }
                                uint64_t long_lists = __ballot(nch[s] > kOwnChunks);
                                while (long_lists) {
                                    const int m = __builtin_ctzll(long_lists);
                                    long_lists &= long_lists - 1;
                                    const uint32_t l_first = __builtin_amdgcn_readlane(at, m);
                                    const uint32_t l_len = __builtin_amdgcn_readlane(llen[s], m);
                                    const uint64_t l_start = readlane_u64(start[s], m);
                                    for (uint32_t c = (uint32_t)lane + kOwnChunks; (c << 6) < l_len; c += kWave) {
                                        const uint32_t idx = l_first + c - w0;
                                        if (idx < cap) {
                                            const uint32_t rest = l_len - (c << 6);
                                            const uint64_t cnt = rest < (uint32_t)kWave ? rest : (uint32_t)kWave;
                                            list[idx] = chunk_address<TeamChunks>(p, l_start, c) | (cnt << 48);
                                        }
                                    }
                                }
                            }
                        }
                        TEAM_STAMP(2)  // descriptors
                        __syncthreads();
                        TEAM_STAMP(3)
                        if (my_total > w0) {
                            const uint32_t n_round = min(my_total - w0, cap);
                            const uint32_t n_padded = (n_round + kTeamRing - 1u) & ~(kTeamRing - 1u);
                            if ((uint32_t)lane < n_padded - n_round) lds.desc[n_round + lane] = null_chunk(p);
                            stream_round<TeamChunks, CountT, (int)kTeamRing, false>(p, lds.desc, n_padded, score_top, count_top, nullptr, n_round);
                        }
                        TEAM_STAMP(4)  // stream
                        __syncthreads();
                        TEAM_STAMP(5)  // waiting for the other slices' streams
                    }
                }
                // ---- ambiguous k-mers (place.cpp:306-313, 373-415), after all exact ones ----------
                if (any_amb) {
                    const int64_t amb_slot = ((kMode == kTeamAccumulate || kMode == kTeamAccumulateLists) && p.amb_slot) ? (int64_t)p.amb_slot[read] : -1;
                    place_ambiguous<TeamChunks, CountT>(kp, lds, seq, len, n_kmers, amb_slot, ctx);
                }
            }
            if constexpr (kMode == kTeamAccumulateLists) {
                // k-mer-space shard, partial lists: the rows that received a k-mer, where the scan kernel made room
                const uint64_t slice_at = read * n_slices + pass * W + wave;
                const uint2 ix = tp.sparse_index[slice_at];
                const uint32_t room = tp.sparse_cap[slice_at];
                const uint32_t part = (uint32_t)(read / tp.sparse_part_reads);
                const uint64_t first = readlane_u64(part_first, (int)part) + ix.x;
                const bool fits = first + room <= tp.sparse_entries_cap;
                const uint32_t n_out = emit_partial_list<CountT>(lds, rows_pad, ctx.rows_,
                                                                 tp.sparse_entries + first * PartialEntry<CountT>::kBytes,
                                                                 fits ? room : 0u);
                if (lane == 0) tp.sparse_index[slice_at].y = fits ? n_out : kSparseOverflow;
            } else if (kMode == kTeamAccumulate) {
                // k-mer-space shard: the slice's raw sums and counts leave for HBM (see place_kernel.hip)
                for (uint32_t i = lane; i < rows_pad; i += kWave) {
                    const uint2 cv = lds.load(i);
                    if (i < ctx.rows_) {
                        const uint64_t at = read * p.num_branches + ctx.base_ + i;
                        p.partial_scores[at] = __uint_as_float(cv.x);
                        p.partial_counts[at] = (uint16_t)(cv.y & ~(uint32_t)Lds::kSeen);
                    }
                    lds.store(i, 0u, 0u);
                }
            } else {
                // ---- correction, the slice's best rows and share of sum_scores, reset of the rows ----
                if (lane == 0) lds.store(rows_pad - 1u, 0u, 0u);  // the dummy row of the out-of-range lanes
                place_epilogue<TeamChunks, CountT>(kp, lds, read, n_kmers, ctx);
            }
            // A wave's descriptor list is its scratch in the ambiguous sweep and the epilogue (seen bits,
            // top-k candidates), and the tile waves of the next pass / read write into every list as soon as
            // THEY have met: nobody may start that before everybody is through here.  Placing, the barrier
            // in front of the merge below does it after the last pass.
            if (kMode == kTeamAccumulate || kMode == kTeamAccumulateLists || pass + 1 < tp.passes) __syncthreads();
        }
        TEAM_STAMP(6)  // slice epilogue
        if (kMode != kTeamAccumulate && kMode != kTeamAccumulateLists) {
            __syncthreads();  // every slice's rows and sums are in LDS
            if (wave == W - 1) {
                MergeParams mp;
                mp.keep_at_most = p.keep_at_most;
                mp.kmer_size = k;
                mp.num_branches = p.num_branches;
                mp.log_threshold = p.log_threshold;
                mp.keep_factor = p.keep_factor;
                mp.rows = p.rows;
                mp.n_rows = p.n_rows;
                mp.kmer_counts = p.kmer_counts;
                team_merge(mp, merge_cand, merge_stride, partials, n_slices, read, n_kmers);
            }
            TEAM_STAMP(7)  // waiting for the other slices' epilogues (+ the merge, in the last wave)
            // placing, the barriers of the next read's front end keep the other waves' next epilogue
            // off the merge area until the merge is done; finishing, there is no front end
            if (kMode == kTeamFinish) __syncthreads();
        }
        if (kMode != kTeamFinish && wave == 0 && lane == 0) flags[parity] = 0u;  // read again two reads on
    }
#ifdef EPIK_AMD_ABLATION
    if (p.dbg && lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&p.dbg[wave * 8 + i], dbg_t[i]);
#endif
}

// Algorithmic bytes of SURVEY.md 8(d) on the sliced database: a code's list is its sublists of all passes.
template <int W>
__global__ void team_algorithmic_bytes_kernel(TeamParams tp, unsigned long long *total)
{
    algorithmic_bytes_block(tp.base, total, [&](uint32_t key) {
        uint32_t n = 0;
        for (uint32_t pass = 0; pass < tp.passes; ++pass) {
            TeamEntry<W> e;
            e.load(tp, pass, key);
#pragma unroll
            for (int s = 0; s < W; ++s) n += e.len[s];
        }
        return n;
    });
}

namespace {

template <typename F>
hipError_t team_dispatch(int waves, int counts, int mode, F &&f)
{
#define EPIK_TEAM_CASE(W, C, M) \
    if (waves == W && counts == C && mode == M) \
        return f.template operator()<W, std::conditional_t<C == kCounts8, uint8_t, std::conditional_t<C == kCounts16, uint16_t, uint32_t>>, M>();
#define EPIK_TEAM_MODES(W, C) EPIK_TEAM_CASE(W, C, kTeamPlace) EPIK_TEAM_CASE(W, C, kTeamAccumulate) EPIK_TEAM_CASE(W, C, kTeamFinish) \
    EPIK_TEAM_CASE(W, C, kTeamAccumulateLists)
#define EPIK_TEAM_COUNTS(W) EPIK_TEAM_MODES(W, kCounts8) EPIK_TEAM_MODES(W, kCounts16) EPIK_TEAM_MODES(W, kCounts32)
    EPIK_TEAM_COUNTS(2)
    EPIK_TEAM_COUNTS(4)
    EPIK_TEAM_COUNTS(8)
#undef EPIK_TEAM_COUNTS
#undef EPIK_TEAM_MODES
#undef EPIK_TEAM_CASE
    return hipErrorInvalidValue;
}

}  // namespace

hipError_t launch_team(const TeamParams &tp, int waves, int counts, int mode, dim3 grid, size_t lds_bytes,
                       hipStream_t stream)
{
    return team_dispatch(waves, counts, mode, [&]<int W, typename C, int M>() {
        hipLaunchKernelGGL((team_place_kernel<W, C, M>), grid, dim3(W * 64), lds_bytes, stream, tp);
        return hipGetLastError();
    });
}

hipError_t set_team_lds_limit(int waves, int counts, size_t /*lds_bytes*/)  // (always the whole CU: see place_kernel.hip)
{
    hipError_t err = hipSuccess;
    for (int mode = 0; mode < 4 && err == hipSuccess; ++mode)
        err = team_dispatch(waves, counts, mode, [&]<int W, typename C, int M>() {
            return hipFuncSetAttribute(reinterpret_cast<const void *>(&team_place_kernel<W, C, M>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu);
        });
    return err;
}

hipError_t team_occupancy(int waves, int counts, size_t lds_bytes, int *blocks_per_cu)
{
    return team_dispatch(waves, counts, kTeamPlace, [&]<int W, typename C, int M>() {
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, team_place_kernel<W, C, M>, W * 64, lds_bytes);
    });
}

hipError_t launch_team_algorithmic_bytes(const TeamParams &tp, int waves, unsigned long long *d_total, hipStream_t stream)
{
    const dim3 block(256), grid((unsigned)((tp.base.n_reads + 255) / 256));
    if (waves == 2)
        hipLaunchKernelGGL((team_algorithmic_bytes_kernel<2>), grid, block, 0, stream, tp, d_total);
    else if (waves == 4)
        hipLaunchKernelGGL((team_algorithmic_bytes_kernel<4>), grid, block, 0, stream, tp, d_total);
    else if (waves == 8)
        hipLaunchKernelGGL((team_algorithmic_bytes_kernel<8>), grid, block, 0, stream, tp, d_total);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace epik_amd
