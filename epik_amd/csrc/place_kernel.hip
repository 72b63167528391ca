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
// lanes stride one posting list at a time (8-byte {branch, score} postings,
// coalesced), and because branches are distinct inside a list no two lanes of an
// instruction touch the same LDS cell, while LDS executes a wave's instructions in
// order -- so every branch receives its float32 adds in exactly the k-mer order of
// place.cpp:349-371 and the sums are bit-identical to the CPU loop.
//
// This is gather / scatter-add: HBM-bandwidth bound, no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "place_kernel.h"

namespace epik_amd {

namespace {

constexpr int kWave = 64;
constexpr uint32_t kAmbSeen = 0x80000000u;  // counts[] bit: branch already scored by an ambiguous key

__device__ __forceinline__ int lane_id() { return (int)__lane_id(); }

// float -> unsigned that sorts like the float
__device__ __forceinline__ uint32_t ord_f32(float f)
{
    const uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m)
{
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m);
    const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const uint64_t o = shfl_xor_u64(v, m);
        v = o > v ? o : v;
    }
    return v;
}

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        v += __longlong_as_double((long long)shfl_xor_u64((uint64_t)__double_as_longlong(v), m));
    }
    return v;
}

__device__ __forceinline__ double readlane0_f64(double v)
{
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)u, 0);
    const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(u >> 32), 0);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += (uint32_t)__shfl_xor((int)v, m);
    return v;
}

// 10^x in double.  The reference calls glibc pow(10.0, x) (place.cpp:46,181,254);
// the device routine differs from it by ulps, which is far inside the 1e-5 LWR bar.
__device__ __forceinline__ double pow10_f64(double x) { return exp10(x); }

template <typename OffT>
__device__ __forceinline__ void load_range(const OffT *__restrict__ offsets, uint32_t key,
                                           uint64_t &start, uint32_t &len)
{
    const OffT b = offsets[key];
    const OffT e = offsets[(uint64_t)key + 1];
    start = (uint64_t)b;
    len = (uint32_t)(e - b);
}

// Everything a wave knows about the 64-character tile it is encoding.
struct Tile {
    uint32_t key;       // k-mer code of the window starting at this lane (ambiguous position = state 0)
    uint32_t cls;       // char_class of this lane's character
    uint64_t inv_mask;  // wave-uniform: lanes whose character is invalid
    uint64_t amb_mask;  // wave-uniform: lanes whose character is ambiguous
    bool in_range;      // this lane starts a window of the read (p < n_kmers, lane < tile stride)
};

// i2l::to_kmers<one_ambiguity_policy> for one tile (place.cpp:294): each lane
// classifies one character; the window code is gathered from the next k-1 lanes.
__device__ __forceinline__ Tile encode_tile(const uint8_t *__restrict__ seq, uint64_t len,
                                            uint64_t tile_pos, uint64_t n_kmers, uint32_t k,
                                            uint32_t sigma, uint32_t stride,
                                            const uint32_t *__restrict__ char_class)
{
    Tile t;
    const int lane = lane_id();
    const uint64_t pos = tile_pos + (uint64_t)lane;
    uint32_t cls = 0;
    if (pos < len) cls = char_class[seq[pos]];
    const bool multi = (cls & (cls - 1)) != 0;
    const uint32_t state = (cls && !multi) ? (uint32_t)(__ffs((int)cls) - 1) : 0u;
    t.cls = cls;
    t.inv_mask = __ballot(cls == 0 && pos < len);  // characters past the end belong to no window
    t.amb_mask = __ballot(multi);
    uint32_t key = state;
    for (uint32_t j = 1; j < k; ++j) {
        const uint32_t nxt = (uint32_t)__shfl((int)state, lane + (int)j);
        key = key * sigma + nxt;
    }
    t.key = key;
    t.in_range = ((uint32_t)lane < stride) && (pos < n_kmers);
    return t;
}

struct WaveLds {
    float *scores;     // [N]  _scores[thread]  (place.h:126)
    uint32_t *counts;  // [N]  _counts[thread]  (place.h:131); bit 31 = kAmbSeen
};

// One chunk of one posting list: lanes [0, cnt) each take one posting and add it
// to the wave's score/count vectors (place.cpp:358-367).
template <bool kLdsAtomic>
__device__ __forceinline__ void accumulate_chunk(const WaveLds &lds, uint2 e, bool active)
{
    if (active) {
        const uint32_t b = e.x;
        const float sc = __uint_as_float(e.y);
        if (kLdsAtomic) {
            // ds_add_f32 / ds_add_u32, no return: one RNE float add per cell, in wave issue order
            __hip_atomic_fetch_add(&lds.scores[b], sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(&lds.counts[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        } else {
            lds.scores[b] = __fadd_rn(lds.scores[b], sc);
            lds.counts[b] = lds.counts[b] + 1u;
        }
    }
}

}  // namespace

template <typename OffT, bool kLdsAtomic>
__global__ __launch_bounds__(256) void place_reads_kernel(PlaceParams p)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int lane = lane_id();
    const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t waves_per_block = blockDim.x >> 6;

    WaveLds lds;
    {
        unsigned char *base = lds_raw + (size_t)wave_in_block * p.lds_wave_bytes;
        lds.scores = reinterpret_cast<float *>(base);
        lds.counts = reinterpret_cast<uint32_t *>(base + (size_t)p.n_pad * 4);
    }
    const uint32_t N = p.num_branches;
    for (uint32_t i = lane; i < p.n_pad; i += kWave) {
        lds.scores[i] = 0.0f;
        lds.counts[i] = 0u;
    }

    const OffT *__restrict__ offsets = static_cast<const OffT *>(p.offsets);
    const uint2 *__restrict__ values = p.values;
    const uint32_t k = p.kmer_size;
    const uint32_t sigma = p.alphabet_size;
    const uint32_t stride = kWave - (k - 1);  // windows per 64-character tile
    const float k_f = (float)k;
    const float log_thr = p.log_threshold;

    const uint64_t wave_global = (uint64_t)blockIdx.x * waves_per_block + wave_in_block;
    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_block;

    for (uint64_t read = wave_global; read < p.n_reads; read += total_waves) {
        const uint64_t seq_begin = p.seq_offsets[read];
        const uint64_t len = p.seq_offsets[read + 1] - seq_begin;
        const uint8_t *__restrict__ seq = p.seqs + seq_begin;
        if (len < k) {  // place.cpp:322 underflows here; we report "no placement"
            if (lane == 0) p.n_rows[read] = 0;
            continue;
        }
        const uint64_t n_kmers = len - k + 1;  // :322
        bool any_amb = false;

        // ---- exact k-mers, read order (place.cpp:294-305, 349-371) -------------------
        for (uint64_t tile_pos = 0; tile_pos < n_kmers; tile_pos += stride) {
            const Tile t = encode_tile(seq, len, tile_pos, n_kmers, k, sigma, stride, p.char_class);
            uint64_t start = 0;
            uint32_t llen = 0;
            bool exact = t.in_range;
            if ((t.inv_mask | t.amb_mask) != 0) {  // wave-uniform, cold
                const uint64_t wmask = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
                const uint64_t inv_w = (t.inv_mask >> lane) & wmask;
                const uint64_t amb_w = (t.amb_mask >> lane) & wmask;
                const bool is_amb = t.in_range && inv_w == 0 && __popcll(amb_w) == 1;
                exact = t.in_range && inv_w == 0 && amb_w == 0;
                any_amb = any_amb || (__ballot(is_amb) != 0);
            }
            if (exact) load_range(offsets, t.key, start, llen);

            // lists of this tile, one k-mer at a time, 64 postings per step
            uint64_t found = __ballot(llen != 0);
            while (found) {
                const int m = __builtin_ctzll(found);
                found &= found - 1;
                const uint32_t s_lo = __builtin_amdgcn_readlane((uint32_t)start, m);
                const uint32_t s_hi = __builtin_amdgcn_readlane((uint32_t)(start >> 32), m);
                const uint32_t n = __builtin_amdgcn_readlane(llen, m);
                const uint2 *__restrict__ list = values + (((uint64_t)s_hi << 32) | s_lo);
                for (uint32_t off = 0; off < n; off += kWave) {
                    const bool active = off + (uint32_t)lane < n;
                    uint2 e = make_uint2(0u, 0u);
                    if (active) e = list[off + lane];
                    accumulate_chunk<kLdsAtomic>(lds, e, active);
                }
            }
        }

        // ---- ambiguous k-mers (place.cpp:306-313, 373-415), after all exact ones ------
        if (any_amb) {
            const float thr = p.threshold;
            for (uint64_t tile_pos = 0; tile_pos < n_kmers; tile_pos += stride) {
                const Tile t = encode_tile(seq, len, tile_pos, n_kmers, k, sigma, stride, p.char_class);
                if (t.amb_mask == 0) continue;
                const uint64_t wmask = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
                const uint64_t inv_w = (t.inv_mask >> lane) & wmask;
                const uint64_t amb_w = (t.amb_mask >> lane) & wmask;
                const bool is_amb = t.in_range && inv_w == 0 && __popcll(amb_w) == 1;
                uint64_t todo = __ballot(is_amb);
                while (todo) {
                    const int m = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    const uint64_t amb_w_m = (t.amb_mask >> m) & wmask;
                    const int j = __builtin_ctzll(amb_w_m);          // ambiguous position in the window
                    const uint32_t cls = __builtin_amdgcn_readlane(t.cls, m + j);
                    const uint32_t key0 = __builtin_amdgcn_readlane(t.key, m);
                    uint32_t weight = 1;
                    for (uint32_t q = (uint32_t)j + 1; q < k; ++q) weight *= sigma;
                    // every resolved key, ascending state order, is searched on its own (:308-312)
                    for (uint32_t st = 0; st < sigma; ++st) {
                        if (!((cls >> st) & 1u)) continue;
                        const uint32_t key = key0 + st * weight;
                        const uint64_t b0 = (uint64_t)offsets[key];
                        const uint32_t n = (uint32_t)((uint64_t)offsets[(uint64_t)key + 1] - b0);
                        const uint2 *__restrict__ list = values + b0;
                        for (uint32_t off = 0; off < n; off += kWave) {
                            if (off + (uint32_t)lane < n) {
                                const uint2 e = list[off + lane];
                                const uint32_t b = e.x;
                                const uint32_t c = lds.counts[b];
                                // Only the first ambiguous key that reaches a branch scores it:
                                // later ones find counts_amb[b] != 0 and stay out of l_amb (:385-388).
                                if (!(c & kAmbSeen)) {
                                    // counts_amb[b] == 1, scores_amb[b] == float(pow(10, score)) (:390-391)
                                    const float prob = (float)pow(10.0, (double)__uint_as_float(e.y));
                                    const float avg = __fdiv_rn(
                                        __fadd_rn(prob, __fmul_rn((float)(k - 1u), thr)), k_f);  // :400-402
                                    lds.counts[b] = (c | kAmbSeen) + 1u;                          // :409
                                    lds.scores[b] = __fadd_rn(lds.scores[b], avg);                // :410
                                }
                            }
                        }
                    }
                }
            }
        }

        // ---- score correction (:418-422) + sum_scores (:164-184), dense over N -----------
        const float nk_f = (float)n_kmers;
        double sum_placed = 0.0;
        uint32_t touched = 0;
        for (uint32_t i = lane; i < N; i += kWave) {
            const uint32_t c = lds.counts[i] & ~kAmbSeen;
            float s = -INFINITY;
            if (c != 0) {
                s = lds.scores[i];
                s = __fadd_rn(s, __fmul_rn((float)(n_kmers - (uint64_t)c), log_thr));  // :420
                s = __fdiv_rn(s, k_f);                                                 // :421
                sum_placed += pow10_f64((double)s);                                    // :181
                ++touched;
            }
            lds.scores[i] = s;  // -inf marks "not an edge"
            lds.counts[i] = c;
        }
        touched = wave_sum_u32(touched);
        sum_placed = wave_sum_f64(sum_placed);
        const float thr_score = __fdiv_rn(__fmul_rn(nk_f, log_thr), k_f);  // :175 / :146-147
        const double p_thr = pow10_f64((double)thr_score);
        const double sum_not_placed = (double)((float)N - (float)touched) * p_thr;  // :174-175
        const double score_sum = sum_not_placed + sum_placed;                       // :183
        const double keep_factor = (score_sum == 0.0) ? 0.0 : p.keep_factor;        // :247-251

        // ---- select_best_placements (:134-159): lane r ends up holding row r ------------
        const uint32_t keep = p.keep_at_most;
        uint32_t n_sel;
        float row_s = 0.0f;
        uint32_t row_b = 0, row_c = 0;
        if (touched == 0) {  // :141-152
            n_sel = keep;
            row_s = thr_score;
            row_b = (uint32_t)lane;
            row_c = 0;
        } else {
            n_sel = keep < touched ? keep : touched;  // :137
            uint64_t prev = ~0ull;                   // key of the previous winner
            float stop_below = -INFINITY;
            uint32_t r = 0;
            for (; r < n_sel; ++r) {
                uint64_t best = 0;  // (ord(score) << 32) | ~branch : max = higher score, then lower branch
                for (uint32_t i = lane; i < N; i += kWave) {
                    const float s = lds.scores[i];
                    const uint64_t key = ((uint64_t)ord_f32(s) << 32) | (uint64_t)(~i);
                    if (s != -INFINITY && key < prev && key > best) best = key;
                }
                best = wave_max_u64(best);
                const uint32_t bb = ~(uint32_t)best;
                const float bs = lds.scores[bb];
                // rows below best + log10(keep_factor) cannot pass filter_by_ratio (:188-199)
                if (r == 0 && keep_factor > 0.0) stop_below = bs + p.log10_keep_factor_margin;
                if (bs < stop_below) break;
                if ((uint32_t)lane == r) {
                    row_s = bs;
                    row_b = bb;
                    row_c = lds.counts[bb];
                }
                prev = best;
            }
            n_sel = r;
        }

        // ---- LWR (:241-264) and filter_by_ratio (:188-199), one row per lane ---------------
        const bool has_row = (uint32_t)lane < n_sel;
        double lwr = 0.0;
        if (has_row && score_sum != 0.0) {
            const double power = pow10_f64((double)row_s);           // :254
            lwr = (power == 0.0) ? 0.0 : power / score_sum;          // :255-262
        }
        const double best_ratio = readlane0_f64(lwr);              // :191, rows are sorted
        const double ratio_threshold = best_ratio * keep_factor;    // :192
        const bool kept = has_row && lwr >= ratio_threshold;        // :197
        const uint64_t kept_mask = __ballot(kept);
        if (kept) {
            const uint32_t slot = (uint32_t)__popcll(kept_mask & ((1ull << lane) - 1ull));
            epik_amd_placement out;
            out.branch = row_b;
            out.score = row_s;
            out.lwr = lwr;
            p.rows[read * keep + slot] = out;
            if (p.kmer_counts) p.kmer_counts[read * keep + slot] = row_c;
        }
        if (lane == 0) p.n_rows[read] = (uint32_t)__popcll(kept_mask);

        // ---- reset the wave's vectors for its next read (place.cpp:335-342) -------------
        for (uint32_t i = lane; i < N; i += kWave) {
            lds.scores[i] = 0.0f;
            lds.counts[i] = 0u;
        }
    }
}

// Algorithmic bytes of SURVEY.md 8(d): one thread per read, plain loops.
template <typename OffT>
__global__ void algorithmic_bytes_kernel(PlaceParams p, unsigned long long *total)
{
    const uint64_t read = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bytes = 0;
    if (read < p.n_reads) {
        const OffT *offsets = static_cast<const OffT *>(p.offsets);
        const uint64_t b = p.seq_offsets[read];
        const uint64_t len = p.seq_offsets[read + 1] - b;
        const uint8_t *seq = p.seqs + b;
        const uint32_t k = p.kmer_size;
        bytes = len;
        if (len >= k) {
            const uint64_t n_kmers = len - k + 1;
            bytes += 8ull * n_kmers;
            unsigned long long entries = 0;
            for (uint64_t pos = 0; pos < n_kmers; ++pos) {
                uint64_t key = 0, weight = 0;
                uint32_t n_amb = 0, amb_cls = 0, amb_pos = 0;
                bool ok = true;
                for (uint32_t j = 0; j < k; ++j) {
                    const uint32_t cls = p.char_class[seq[pos + j]];
                    if (cls == 0) { ok = false; break; }
                    uint32_t st = 0;
                    if (cls & (cls - 1)) { ++n_amb; amb_cls = cls; amb_pos = j; }
                    else st = (uint32_t)(__ffs((int)cls) - 1);
                    key = key * p.alphabet_size + st;
                }
                if (!ok || n_amb > 1) continue;
                if (n_amb == 0) {
                    entries += (unsigned long long)(offsets[key + 1] - offsets[key]);
                } else {
                    weight = 1;
                    for (uint32_t j = amb_pos + 1; j < k; ++j) weight *= p.alphabet_size;
                    for (uint32_t st = 0; st < p.alphabet_size; ++st)
                        if ((amb_cls >> st) & 1u) {
                            const uint64_t kk = key + (uint64_t)st * weight;
                            entries += (unsigned long long)(offsets[kk + 1] - offsets[kk]);
                        }
                }
            }
            bytes += 8ull * entries;
            if (p.n_rows) bytes += 16ull * p.n_rows[read];
        }
    }
    // block reduce then one atomic
    __shared__ unsigned long long partial[64];
    unsigned long long v = bytes;
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_u64(v, m);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) partial[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long s = 0;
        for (unsigned i = 0; i < (blockDim.x >> 6); ++i) s += partial[i];
        atomicAdd(total, s);
    }
}

hipError_t launch_place_reads(const PlaceParams &p, bool offsets64, bool lds_atomic, dim3 grid,
                              dim3 block, size_t lds_bytes, hipStream_t stream)
{
    if (offsets64) {
        if (lds_atomic)
            hipLaunchKernelGGL((place_reads_kernel<uint64_t, true>), grid, block, lds_bytes, stream, p);
        else
            hipLaunchKernelGGL((place_reads_kernel<uint64_t, false>), grid, block, lds_bytes, stream, p);
    } else {
        if (lds_atomic)
            hipLaunchKernelGGL((place_reads_kernel<uint32_t, true>), grid, block, lds_bytes, stream, p);
        else
            hipLaunchKernelGGL((place_reads_kernel<uint32_t, false>), grid, block, lds_bytes, stream, p);
    }
    return hipGetLastError();
}

hipError_t set_place_reads_lds_limit(size_t lds_bytes)
{
    hipError_t e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&place_reads_kernel<uint32_t, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&place_reads_kernel<uint32_t, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&place_reads_kernel<uint64_t, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&place_reads_kernel<uint64_t, false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
}

hipError_t place_reads_occupancy(bool offsets64, bool lds_atomic, int block_threads, size_t lds_bytes,
                                 int *blocks_per_cu)
{
    if (offsets64) {
        if (lds_atomic)
            return hipOccupancyMaxActiveBlocksPerMultiprocessor(
                blocks_per_cu, place_reads_kernel<uint64_t, true>, block_threads, lds_bytes);
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(
            blocks_per_cu, place_reads_kernel<uint64_t, false>, block_threads, lds_bytes);
    }
    if (lds_atomic)
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(
            blocks_per_cu, place_reads_kernel<uint32_t, true>, block_threads, lds_bytes);
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(
        blocks_per_cu, place_reads_kernel<uint32_t, false>, block_threads, lds_bytes);
}

hipError_t launch_algorithmic_bytes(const PlaceParams &p, bool offsets64, unsigned long long *d_total,
                                    hipStream_t stream)
{
    const dim3 block(256);
    const dim3 grid((unsigned)((p.n_reads + 255) / 256));
    if (offsets64)
        hipLaunchKernelGGL((algorithmic_bytes_kernel<uint64_t>), grid, block, 0, stream, p, d_total);
    else
        hipLaunchKernelGGL((algorithmic_bytes_kernel<uint32_t>), grid, block, 0, stream, p, d_total);
    return hipGetLastError();
}

}  // namespace epik_amd
