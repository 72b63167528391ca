// place_kernel.hip -- the epik::placer hot loop as CDNA4 (gfx950) kernels.
//
// Reference path (all /root/reference/epik/src/epik/place.cpp):
//   query_kmers            :278-316   i2l rolling k-mer encode + phylo_kmer_db::search
//   place_seq              :320-440   per-branch float32 log-score accumulate + correction
//   sum_scores             :164-184   double sum of 10^score over touched branches
//   select_best_placements :134-159   top keep_at_most by score
//   LWR + filter_by_ratio  :241-267, :188-199
//
// Mapping: ONE WAVEFRONT (64 lanes) PLACES ONE READ.  The per-branch score and
// count vectors (`_scores[thread]`, `_counts[thread]`, place.h:126-131) live in a
// wave-private slice of LDS.  K-mers are consumed strictly in read order; the 64
// lanes stride one posting list at a time (6-byte {score, row} postings, coalesced),
// and because branches are distinct inside a list no two lanes of an instruction
// touch the same LDS cell, while LDS executes a wave's instructions in order -- so
// every branch receives its float32 adds in exactly the k-mer order of
// place.cpp:349-371 and the sums are bit-identical to the CPU loop.
//
// This is gather / scatter-add: HBM-bandwidth bound, no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "place_device.hpp"

namespace epik_amd {

template <typename Layout, typename CountT>
__global__ __launch_bounds__(256, 5) void place_reads_kernel(PlaceParams p)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int lane = lane_id();
    const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t waves_per_block = blockDim.x >> 6;

    WaveLds<CountT> lds;
    {
        unsigned char *base = lds_raw + (size_t)wave_in_block * p.lds_wave_bytes;
        lds.carve(base, p.n_pad);
    }
    // LDS byte addresses of the dummy row (cell 0) in the two vectors: row = n_pad - 1 - cell
    const uint32_t score_top = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds.score + (p.n_pad - 1u) * 4u);
    const uint32_t count_top = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds.count +
                                                              (p.n_pad - 1u) * (uint32_t)sizeof(CountT));
    // the argument block itself, for the out-of-line parts (no private copy of `p`)
    const PlaceParams *kp = (const PlaceParams *)__builtin_amdgcn_kernarg_segment_ptr();
    for (uint32_t i = lane; i < p.n_pad; i += kWave) lds.store(i, 0u, 0u);

    const uint32_t k = p.kmer_size;
    const uint32_t sigma = p.alphabet_size;
    const uint32_t stride = kWave - (k - 1);  // windows per 64-character tile
    const uint64_t wave_global = (uint64_t)blockIdx.x * waves_per_block + wave_in_block;
    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_block;

#ifdef EPIK_AMD_ABLATION
    unsigned long long dbg_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define EPIK_STAMP(k)                                                        \
    if (p.dbg) {                                                             \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();        \
        dbg_t[k] += now_ - dbg_last;                                         \
        dbg_last = now_;                                                     \
    }
    unsigned long long dbg_last = p.dbg ? __builtin_amdgcn_s_memtime() : 0;
#else
#define EPIK_STAMP(k)
#endif
    for (uint64_t read = wave_global; read < p.n_reads; read += total_waves) {
        // the read's bounds are the same in every lane: keep them (and every length, position and
        // loop bound derived from them) in scalar registers
        const uint64_t seq_begin = readlane_u64(p.seq_offsets[read], 0);
        const uint64_t len = readlane_u64(p.seq_offsets[read + 1], 0) - seq_begin;
        const uint8_t *__restrict__ seq = p.seqs + seq_begin;
        // place.cpp:322 underflows for len < k; we report "no placement" -- also for a read whose
        // k-mers could overflow this kernel's count type (the host then uses the wide kernel)
        // (n_rows = kCountsTooNarrow tells the caller, who chose the count width, which case it was)
        if (len < k || len - k + 1 > WaveLds<CountT>::kMaxKmers || (p.max_kmers_cap && len - k + 1 > p.max_kmers_cap)) {
            if (p.partial_scores) {  // accumulate-only launch: an all-zero partial vector
                for (uint32_t i = lane; i < p.num_branches; i += kWave) {
                    p.partial_scores[read * p.num_branches + i] = 0.0f;
                    p.partial_counts[read * p.num_branches + i] = 0u;
                }
            } else if (lane == 0) {
                p.n_rows[read] = len < k ? 0u : kCountsTooNarrow;
            }
            continue;
        }
        const uint64_t n_kmers = len - k + 1;  // :322
        bool any_amb = false;

        // ---- exact k-mers, read order (place.cpp:294-305, 349-371) -------------------
        // A pass covers kTilesPerPass 64-character tiles: (1) encode every window and
        // issue all table lookups, (2) compact the found lists, in read order, into the
        // wave's LDS array of chunk descriptors, (3) stream their postings through a ring
        // of kRing chunks in flight.
        for (uint64_t pass_pos = 0; pass_pos < n_kmers; pass_pos += (uint64_t)kTilesPerPass * stride) {
            uint64_t start[kTilesPerPass];
            uint32_t llen[kTilesPerPass];
            // the characters of all tiles, then their classes: two round trips per pass, not two per tile
            uint32_t tile_cls[kTilesPerPass];
#pragma unroll
            for (int t = 0; t < kTilesPerPass; ++t) tile_cls[t] = tile_char(seq, len, pass_pos + (uint64_t)t * stride);
#pragma unroll
            for (int t = 0; t < kTilesPerPass; ++t)
                tile_cls[t] = tile_class(tile_cls[t], len, pass_pos + (uint64_t)t * stride, p.char_class);
#pragma unroll
            for (int t = 0; t < kTilesPerPass; ++t) {
                start[t] = 0;
                llen[t] = 0;
                const uint64_t tile_pos = pass_pos + (uint64_t)t * stride;
                if (tile_pos < n_kmers) {  // wave-uniform
                    const Tile tl = tile_from_class(tile_cls[t], len, tile_pos, n_kmers, k, sigma, stride);
                    bool exact = tl.in_range;
                    if ((tl.inv_mask | tl.amb_mask) != 0) {  // wave-uniform, cold
                        const uint64_t wmask = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
                        const uint64_t inv_w = (tl.inv_mask >> lane) & wmask;
                        const uint64_t amb_w = (tl.amb_mask >> lane) & wmask;
                        const bool is_amb = tl.in_range && inv_w == 0 && __popcll(amb_w) == 1;
                        exact = tl.in_range && inv_w == 0 && amb_w == 0;
                        any_amb = any_amb || (__ballot(is_amb) != 0);
                    }
#ifdef EPIK_AMD_ABLATION
                    // bit 16: every lookup falls into the first 4096 table entries (L2-resident):
                    // what the kernel would cost without the table's HBM traffic (wrong lists)
                    // bits 32 / 64: the table shrunk to a half / a quarter (key >> 1, key >> 2)
                    if (p.ablate & (16u | 32u | 64u)) {
                        if (exact)
                            Layout::lookup(p, (p.ablate & 16u) ? (tl.key & 4095u) : (tl.key >> ((p.ablate >> 5) & 3u)),
                                           (uint32_t)tile_pos + (uint32_t)lane, start[t], llen[t]);
                    } else {
                        Layout::lookup_window(p, tl, (uint32_t)tile_pos + (uint32_t)lane, exact, start[t], llen[t]);
                    }
#else
                    Layout::lookup_window(p, tl, (uint32_t)tile_pos + (uint32_t)lane, exact, start[t], llen[t]);
#endif
                }
            }
            EPIK_STAMP(0)  // front end: encode + lookups issued
            // (2) lists -> chunks of <= 64 postings, in read order.  Each lane knows how many
            // chunks its k-mers need; a prefix sum over (tile, lane) gives every chunk its
            // position in the stream, and the lanes write the chunk descriptors
            //     address (48 bits) | count << 48
            // into LDS themselves.  The streaming loop below then spends only a few scalar
            // instructions per chunk (the scalar unit is shared by the whole CU and was the
            // bottleneck of a scalar list-walking generator).
            uint32_t nch[kTilesPerPass], first[kTilesPerPass];
            uint32_t total = 0;
            bool found_any[kTilesPerPass];  // wave-uniform: a sparse database leaves half the tiles of a read without a list
#pragma unroll
            for (int t = 0; t < kTilesPerPass; ++t) {
                nch[t] = (Layout::length(llen[t]) + (uint32_t)kWave - 1u) >> 6;
                found_any[t] = __ballot(nch[t] != 0) != 0;
                first[t] = total;
                if (found_any[t]) {
                    const uint32_t incl = wave_incl_scan_u32(nch[t]);
                    first[t] = total + incl - nch[t];
                    total += __builtin_amdgcn_readlane(incl, 63);
                }
            }
#ifdef EPIK_AMD_ABLATION
            if (p.ablate & 8u) total = 0;   // lookups done, nothing streamed
#endif
            auto *chunks = lds.desc;
            EPIK_STAMP(1)  // lookups landed, scan done
            for (uint32_t w0 = 0; w0 < total; w0 += kChunkCap) {  // one round unless > kChunkCap chunks
                const uint32_t n_round = min(total - w0, kChunkCap);
                const uint32_t n_padded = (n_round + (uint32_t)kRing - 1u) & ~((uint32_t)kRing - 1u);
                // Every lane writes the first kOwnChunks chunks of its own lists (lists of up to 192
                // postings: all but a few percent); the rest of a longer list is written by the
                // whole wave at once, lane j writing chunk kOwnChunks + j, one list at a time.
                constexpr uint32_t kOwnChunks = 3;
#pragma unroll
                for (int t = 0; t < kTilesPerPass; ++t) {
                    if (!found_any[t]) continue;
#pragma unroll
                    for (uint32_t c = 0; c < kOwnChunks; ++c) {
                        const uint32_t idx = first[t] + c - w0;  // wraps when in front of the window
                        if (nch[t] > c && idx < kChunkCap) {
                            const uint32_t rest = Layout::length(llen[t]) - (c << 6);
                            const uint64_t cnt = rest < (uint32_t)kWave ? rest : (uint32_t)kWave;
                            chunks[idx] = Layout::descriptor(p, start[t], llen[t], c, cnt);
                        }
                    }
                    uint64_t long_lists = __ballot(nch[t] > kOwnChunks);
                    while (long_lists) {
                        const int m = __builtin_ctzll(long_lists);
                        long_lists &= long_lists - 1;
                        const uint32_t l_first = __builtin_amdgcn_readlane(first[t], m);
                        const uint32_t l_w = __builtin_amdgcn_readlane(llen[t], m);
                        const uint32_t l_len = Layout::length(l_w);
                        const uint64_t l_start = readlane_u64(start[t], m);
                        // lists of more than 64 + kOwnChunks chunks: the lanes take further turns
                        for (uint32_t c = (uint32_t)lane + kOwnChunks; (c << 6) < l_len; c += kWave) {
                            const uint32_t idx = l_first + c - w0;
                            if (idx < kChunkCap) {
                                const uint32_t rest = l_len - (c << 6);
                                const uint64_t cnt = rest < (uint32_t)kWave ? rest : (uint32_t)kWave;
                                chunks[idx] = Layout::descriptor(p, l_start, l_w, c, cnt);
                            }
                        }
                    }
                }
                if ((uint32_t)lane < n_padded - n_round) chunks[n_round + lane] = Layout::null_descriptor(p);

                // (3) stream the chunks through the ring of kRing in-flight loads (place_device.hpp)
                stream_round<Layout, CountT, kRing, false>(p, chunks, n_padded, score_top, count_top, nullptr, n_round);
            }
        }

        EPIK_STAMP(2)  // expansion + stream
        // ---- ambiguous k-mers (place.cpp:306-313, 373-415), after all exact ones ------
        if (any_amb) {
            const int64_t amb_slot = (p.partial_scores && p.amb_slot) ? (int64_t)p.amb_slot[read] : -1;
            place_ambiguous<Layout, CountT>(kp, lds, seq, len, n_kmers, amb_slot, WaveCtx{});
        }

#ifdef EPIK_AMD_ABLATION
        if (p.ablate & 2u) {  // skip the whole epilogue
            if (lane == 0) p.n_rows[read] = 0;
            if (!(p.ablate & 4u))
                for (uint32_t i = lane; i < p.n_pad; i += kWave) lds.store(i, 0u, 0u);
            continue;
        }
#endif
        if (p.partial_scores) {
            // Accumulate-only launch (k-mer-space shard of the database): this GPU
            // holds part of the lists, so the raw per-branch sums and counts leave for HBM, to be
            // added across GPUs before finish_reads_kernel runs the epilogue on the totals.
            for (uint32_t i = lane; i < p.n_pad; i += kWave) {
                const uint2 cv = lds.load(i);
                if (i < p.num_branches) {
                    p.partial_scores[read * p.num_branches + i] = __uint_as_float(cv.x);
                    p.partial_counts[read * p.num_branches + i] = (uint16_t)(cv.y & ~(uint32_t)WaveLds<CountT>::kSeen);
                }
                lds.store(i, 0u, 0u);
            }
            continue;
        }
        // ---- correction, sum_scores, top-k, LWR, rows out, reset of the wave's vectors ----------
        if (lane == 0) lds.store(p.n_pad - 1u, 0u, 0u);  // the dummy row of the out-of-range lanes
        place_epilogue<Layout, CountT>(kp, lds, read, n_kmers, WaveCtx{});
        EPIK_STAMP(4)  // top-k, LWR, rows out, reset
    }
#ifdef EPIK_AMD_ABLATION
    if (p.dbg && lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&p.dbg[i], dbg_t[i]);
#endif
}

// Second half of a k-mer-space-sharded placement: the per-branch sums and counts of every
// read, already added over the shards, come back from HBM into the wave's LDS vectors and
// go through the same epilogue (place.cpp:418-422, 134-199, 241-267) as a one-GPU placement.
template <typename CountT>
__global__ __launch_bounds__(256, 5) void finish_reads_kernel(PlaceParams p)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int lane = lane_id();
    const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t waves_per_block = blockDim.x >> 6;
    WaveLds<CountT> lds;
    {
        unsigned char *base = lds_raw + (size_t)wave_in_block * p.lds_wave_bytes;
        lds.carve(base, p.n_pad);
    }
    const PlaceParams *kp = (const PlaceParams *)__builtin_amdgcn_kernarg_segment_ptr();
    for (uint32_t i = lane; i < p.n_pad; i += kWave) lds.store(i, 0u, 0u);
    const uint32_t k = p.kmer_size;
    const uint64_t wave_global = (uint64_t)blockIdx.x * waves_per_block + wave_in_block;
    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_block;
    for (uint64_t read = wave_global; read < p.n_reads; read += total_waves) {
        const uint64_t len = p.seq_offsets[read + 1] - p.seq_offsets[read];
        if (len < k || len - k + 1 > WaveLds<CountT>::kMaxKmers || (p.max_kmers_cap && len - k + 1 > p.max_kmers_cap)) {
            if (lane == 0) p.n_rows[read] = len < k ? 0u : kCountsTooNarrow;
            continue;
        }
        // the read's one ambiguous-key record per branch (place.cpp:400-410), already reduced over the
        // shards to the key of smallest order: its average probability joins the sum after all exact
        // scores, as in the one-pass loop, and counts as one k-mer
        const int64_t slot = p.amb_slot ? (int64_t)p.amb_slot[read] : -1;
        for (uint32_t i = lane; i < p.num_branches; i += kWave) {
            float sc = p.partial_scores[read * p.num_branches + i];
            uint32_t c = p.partial_counts[read * p.num_branches + i];
            if (slot >= 0) {
                const float avg = p.amb_avg[(uint64_t)slot * p.num_branches + i];
                if (avg > 0.0f) {
                    sc = __fadd_rn(sc, avg);
                    c += 1u;
                }
            }
            lds.store(i, __float_as_uint(sc), c);
        }
        place_epilogue<PackedLayout<kPlainTable>, CountT>(kp, lds, read, len - k + 1, WaveCtx{});  // also clears the vectors
    }
}

// Algorithmic bytes of SURVEY.md 8(d): one thread per read, plain loops (place_device.hpp).
template <typename Layout>
__global__ void algorithmic_bytes_kernel(PlaceParams p, unsigned long long *total)
{
    algorithmic_bytes_block(p, total, [&](uint32_t key) {
        uint64_t addr;
        uint32_t llen;
        Layout::lookup(p, key, 0u, addr, llen);
        return Layout::length(llen);
    });
}

namespace {

// Calls f.template operator()<Layout, CountT>() for the runtime variant (counts: CountBits).
template <typename Layout, typename F>
hipError_t dispatch_counts(int counts, F &&f)
{
    switch (counts) {
        case kCounts8: return f.template operator()<Layout, uint8_t>();
        case kCounts16: return f.template operator()<Layout, uint16_t>();
        case kCounts32: return f.template operator()<Layout, uint32_t>();
    }
    return hipErrorInvalidValue;
}
// runs: the packed lists in their run-coded form (place_device.hpp, kRuns; chosen by the image builder for
// databases well beyond the Infinity Cache)
template <typename F>
hipError_t dispatch(DbLayout layout, bool runs, int counts, F &&f)
{
    switch (layout) {
        case DbLayout::kCompact32: return dispatch_counts<CompactLayout<uint32_t>>(counts, f);
        case DbLayout::kCompact64: return dispatch_counts<CompactLayout<uint64_t>>(counts, f);
        case DbLayout::kPacked:
            return runs ? dispatch_counts<PackedLayout<kPlainTable, true>>(counts, f) : dispatch_counts<PackedLayout<kPlainTable>>(counts, f);
        case DbLayout::kPaired:
            return runs ? dispatch_counts<PackedLayout<kPairedTable, true>>(counts, f) : dispatch_counts<PackedLayout<kPairedTable>>(counts, f);
        case DbLayout::kFiltered:
            return runs ? dispatch_counts<PackedLayout<kFilteredTable, true>>(counts, f) : dispatch_counts<PackedLayout<kFilteredTable>>(counts, f);
        case DbLayout::kTeam: break;  // team_kernel.hip
    }
    return hipErrorInvalidValue;
}

}  // namespace

hipError_t launch_place_reads(const PlaceParams &p, DbLayout layout, bool runs, int counts, dim3 grid, dim3 block,
                              size_t lds_bytes, hipStream_t stream)
{
    return dispatch(layout, runs, counts, [&]<typename L, typename C>() {
        hipLaunchKernelGGL((place_reads_kernel<L, C>), grid, block, lds_bytes, stream, p);
        return hipGetLastError();
    });
}

// The attribute is a cap per kernel and process, not a reservation (what a launch occupies is what it
// asks for): it is always raised to the whole LDS of a CU, so that placers of different trees alive in
// one process can never lower it under each other's launches.
hipError_t set_place_reads_lds_limit(DbLayout layout, bool runs, int counts, size_t /*lds_bytes*/)
{
    return dispatch(layout, runs, counts, [&]<typename L, typename C>() {
        return hipFuncSetAttribute(reinterpret_cast<const void *>(&place_reads_kernel<L, C>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu);
    });
}

hipError_t place_reads_occupancy(DbLayout layout, bool runs, int counts, int block_threads, size_t lds_bytes,
                                 int *blocks_per_cu)
{
    return dispatch(layout, runs, counts, [&]<typename L, typename C>() {
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, place_reads_kernel<L, C>,
                                                            block_threads, lds_bytes);
    });
}

hipError_t launch_finish_reads(const PlaceParams &p, int counts, dim3 grid, dim3 block, size_t lds_bytes,
                               hipStream_t stream)
{
    return dispatch_counts<PackedLayout<kPlainTable>>(counts, [&]<typename L, typename C>() {
        hipLaunchKernelGGL((finish_reads_kernel<C>), grid, block, lds_bytes, stream, p);
        return hipGetLastError();
    });
}

hipError_t set_finish_reads_lds_limit(int counts, size_t /*lds_bytes*/)
{
    return dispatch_counts<PackedLayout<kPlainTable>>(counts, [&]<typename L, typename C>() {
        return hipFuncSetAttribute(reinterpret_cast<const void *>(&finish_reads_kernel<C>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu);
    });
}

hipError_t launch_algorithmic_bytes(const PlaceParams &p, DbLayout layout, bool runs, unsigned long long *d_total,
                                    hipStream_t stream)
{
    const dim3 block(256);
    const dim3 grid((unsigned)((p.n_reads + 255) / 256));
    return dispatch(layout, runs, kCounts16, [&]<typename L, typename C>() {
        hipLaunchKernelGGL((algorithmic_bytes_kernel<L>), grid, block, 0, stream, p, d_total);
        return hipGetLastError();
    });
}

}  // namespace epik_amd
