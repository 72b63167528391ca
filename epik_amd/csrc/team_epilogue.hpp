// team_epilogue.hpp -- the slice epilogue of team_stream_kernel and the emission of a shard's partial list at the
// cost of the rows a read really touched, for the (read, slice) items that touched few.
//
// Reference path: epik/src/epik/place.cpp:418-422 (score correction), :134-159 (select_best_placements), :164-184
// (sum_scores, this slice's share), :335-342 (the reset of the per-thread vectors).
//
// place_epilogue_body (place_device.hpp) sweeps a slice's ~2 500 rows three times through LDS -- correction, scan,
// reset -- whatever the read touched: 13 300 of an item's 27 800 cycles at N = 9 999
// (profiles/r03_team_stream_wave_timeline.txt).  On the benchmark database an item touches ~40 % of its rows and
// that sweep is as cheap as anything (round 4 tried it otherwise: DESIGN.md 3.2); but on a database built from
// reference sequences a read's lists fall into ONE slice of the four (bench.py --clades), and a shard of a
// k-mer-space-sharded database reaches a small part of every slice.  For those items:
//
//   1. nothing streamed / no entry in any shard's list: the slice is done -- an empty partial result, nothing to
//      reset (publish_empty_slice; the caller knows without looking at a row);
//   2. the slice's COUNTS are read once, four rows (a "quad": one 32-bit word of 8-bit counts, two of 16-bit ones)
//      per lane and trip; a ballot per trip says which quads received a k-mer, and their numbers are compacted
//      into the descriptor list (idle by now; 16 bits per quad, up to kSparseTrips * 64 of them).  More touched
//      quads than that: the dense epilogue runs, nothing has been changed;
//   3. a wave-trip then works on 64 TOUCHED quads instead of 64 consecutive ones: counts and scores gathered, the
//      rows reset by the store behind the read, correction in registers, where the corrected scores stay for tau
//      and for sum_scores' terms: one pass through LDS, straight-line code for 1 .. kSparseTrips trips.
//
// Results are bit for bit those of place_epilogue_body with a team context: the same float32 operations on the same
// operands (the partial sum of sum_scores adds the same float32 terms in another order: inside the 1e-5 bar on
// like_weight_ratio, as before).
#ifndef EPIK_AMD_TEAM_EPILOGUE_HPP
#define EPIK_AMD_TEAM_EPILOGUE_HPP
#include "team_device.hpp"

namespace epik_amd {

namespace {

typedef __attribute__((address_space(3))) v2u lds_v2u;
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4f lds_f32x4;

// (the compiler must not move an LDS access across a phase: the quad list is written over descriptors)
__device__ __forceinline__ void lds_phase() { asm volatile("" ::: "memory"); }
// The lane number as the compiler cannot trace it: the trips below are unrolled, and their addresses and bounds tests
// depend on nothing but the lane and the slice geometry -- invariant in the kernel's loop over reads, so the compiler
// computes them all before that loop and keeps them in registers through the whole kernel.  Behind a volatile asm
// they are computed where they are used.
__device__ __forceinline__ uint32_t here_lane()
{
    uint32_t lane = (uint32_t)lane_id();
    asm volatile("" : "+v"(lane));
    return lane;
}
__device__ __forceinline__ uint32_t here(uint32_t uniform)
{
    uniform = (uint32_t)__builtin_amdgcn_readfirstlane((int)uniform);
    asm volatile("" : "+s"(uniform));
    return uniform;
}

// a quad's counts: one word of 8-bit counts, two of 16-bit ones (the top bit of which is the ambiguous sweep's flag)
template <typename CountT>
struct QuadCounts {
    static_assert(sizeof(CountT) <= 2, "8- and 16-bit counts");
    static constexpr int kWords = (int)sizeof(CountT);
    static constexpr uint32_t kBytes = 4u * (uint32_t)sizeof(CountT);
    uint32_t w[kWords];
    __device__ __forceinline__ void load(uint32_t at)
    {
        if constexpr (kWords == 1) {
            w[0] = *(lds_u32 *)(uintptr_t)at;
        } else {
            const v2u v = *(lds_v2u *)(uintptr_t)at;
            w[0] = v.x & 0x7fff7fffu, w[1] = v.y & 0x7fff7fffu;
        }
    }
    __device__ static __forceinline__ void reset(uint32_t at)
    {
        if constexpr (kWords == 1)
            *(lds_u32 *)(uintptr_t)at = 0u;
        else
            *(lds_v2u *)(uintptr_t)at = v2u{0u, 0u};
    }
    __device__ __forceinline__ bool any() const { return (kWords == 1 ? w[0] : (w[0] | w[1])) != 0u; }
    __device__ __forceinline__ uint32_t count(int u) const
    {
        if constexpr (kWords == 1)
            return (w[0] >> (8 * u)) & 0xffu;
        else
            return (w[u >> 1] >> (16 * (u & 1))) & 0x7fffu;
    }
};

// Step 2: the numbers of the quads that hold a count, 16 bits each, into `list_addr` (the wave's descriptor list), when
// there are at most `cap` of them.  Returns how many there are (wave-uniform); more than `cap`: nothing was written.
// The slice's counts are read once, 16 bytes per lane and trip (16 rows of 8-bit counts: 4 quads; 8 of 16-bit ones:
// 2), all trips out together: ONE round trip to LDS -- with twelve waves of a CU streaming, a round trip is ~400
// cycles, and an epilogue is a chain of them -- then a comparison and a ballot per quad of a lane (what a slice that
// turns out to be dense has paid for nothing), and the listing from the same registers.
template <typename CountT>
__device__ __forceinline__ uint32_t list_touched_quads(uint32_t count_addr, uint32_t rows_pad, uint32_t list_addr, uint32_t cap)
{
    static_assert(sizeof(CountT) <= 2, "8- and 16-bit counts");
    constexpr int kPerLane = sizeof(CountT) == 1 ? 4 : 2;  // quads in the 16 bytes of a lane
    constexpr int kTrips = sizeof(CountT) == 1 ? 4 : 8;    // slices of up to 4 096 rows
    constexpr uint32_t kFlags = sizeof(CountT) == 1 ? 0xffffffffu : 0x7fff7fffu;  // (without the ambiguous sweep's flags)
    const uint32_t lane = here_lane();
    const uint32_t n_vec = rows_pad * (uint32_t)sizeof(CountT) / 16u;  // rows_pad is a multiple of 16
    if (n_vec > 64u * (uint32_t)kTrips) return cap + 1u;  // (a slice nobody gets unasked: the dense epilogue)
    // the quad j of a lane's 16 bytes holds a count
    auto touched = [&](const v4u &v, int j) {
        if constexpr (sizeof(CountT) == 1)
            return v[j] != 0u;
        else
            return ((v[2 * j] | v[2 * j + 1]) & kFlags) != 0u;
    };
    v4u v[kTrips];
#pragma unroll
    for (int t = 0; t < kTrips; ++t) {
        const uint32_t i = (uint32_t)t * 64u + lane;
        v[t] = v4u{0u, 0u, 0u, 0u};
        if (i < n_vec) v[t] = *(lds_u32x4 *)(uintptr_t)(count_addr + 16u * i);
    }
    uint32_t total = 0;
#pragma unroll
    for (int t = 0; t < kTrips; ++t) {
#pragma unroll
        for (int j = 0; j < kPerLane; ++j) total += (uint32_t)__popcll(__ballot(touched(v[t], j)));
    }
    if (total == 0 || total > cap) return total;
    uint32_t at = 0;
#pragma unroll
    for (int t = 0; t < kTrips; ++t) {
#pragma unroll
        for (int j = 0; j < kPerLane; ++j) {
            const bool mine = touched(v[t], j);
            const uint64_t m = __ballot(mine);
            if (m) {
                if (mine) *(lds_u16 *)(uintptr_t)(list_addr + 2u * (at + lanes_below(m))) = (uint16_t)(((uint32_t)t * 64u + lane) * (uint32_t)kPerLane + (uint32_t)j);
                at += (uint32_t)__popcll(m);
            }
        }
    }
    return total;
}

// sum of 10^score over the rows of LDS that hold an edge (not -inf), in double, this lane's share.  Out of line: the
// cold end of sum_scores, whose constants the compiler would otherwise keep in registers through the whole kernel.
__device__ __attribute__((noinline)) double sum_of_powers(uint32_t score_addr, uint32_t n_rows)
{
    double sum = 0.0;
    for (uint32_t i = (uint32_t)lane_id(); i < n_rows; i += kWave) {
        const float x = *(lds_f32 *)(uintptr_t)(score_addr + 4u * i);
        if (x != -INFINITY) sum += pow10_f64((double)x);
    }
    return sum;
}

// what a slice without a touched row hands to the merge (place_epilogue_body with touched == 0 and a team context)
template <typename Ctx>
__device__ __forceinline__ void publish_empty_slice(const Ctx &ctx, uint32_t n_kmers, uint32_t kmer_size, float log_thr, uint32_t keep)
{
    const uint32_t lane = (uint32_t)lane_id();
    const float thr_score = __fdiv_rn(__fmul_rn((float)n_kmers, log_thr), (float)kmer_size);  // :175 / :146-147
    for (uint32_t r = lane; r < keep; r += kWave) ctx.cand[r] = v4u{0u, 0u, 0u, 0u};
    if (lane == 0) {
        ctx.partial->touched = 0u;
        ctx.partial->relative = thr_score > -280.0f ? 1u : 0u;
        ctx.partial->ref_score = thr_score;
        ctx.partial->sum = 0.0;
    }
}

// tau of select_best_placements' candidates: a lower bound, within 2^kTauStop units in the last place, of the
// n_sel-th largest of the 64 lane maxima (1: fewer lanes than that hold an edge) -- see place_epilogue_body.
__device__ __forceinline__ uint32_t tau_of_lane_maxima(uint32_t lane_best, uint32_t top, uint32_t n_sel)
{
    constexpr int kTauStop = EPIK_AMD_TAU_STOP;
    uint32_t prefix = 0;
    int bit = 31;
#pragma unroll
    for (int shared = 20; shared <= 24; shared += 4) {
        const uint32_t trial = top & ~((1u << shared) - 1u);
        if (bit == 31 && trial != 0 && (uint32_t)__popcll(__ballot(lane_best >= trial)) >= n_sel) prefix = trial, bit = shared - 1;
    }
    auto reached = [&](uint32_t trial) { return (uint32_t)__popcll(__ballot(lane_best >= trial)) >= n_sel; };
    if (((bit - kTauStop + 1) & 1) != 0 && bit >= kTauStop) {
        if (reached(prefix | (1u << bit))) prefix |= 1u << bit;
        --bit;
    }
    for (; bit > kTauStop; bit -= 2) {
        const uint32_t t1 = prefix | (1u << (bit - 1)), t2 = prefix | (2u << (bit - 1)), t3 = prefix | (3u << (bit - 1));
        const bool r1 = reached(t1), r2 = reached(t2), r3 = reached(t3);
        prefix = r3 ? t3 : r2 ? t2 : r1 ? t1 : prefix;
    }
    return prefix ? prefix : 1u;
}

struct SliceArgs {
    uint32_t rows_pad, rows, base, kmer_size, keep;
    float log_threshold;
    uint32_t slice_at;  // read * slices + slice: where the slice's results go
    uint32_t trace_at;  // diagnostic builds
    uint32_t untouched; // nothing reached the slice's rows (the dense epilogue: no sweep over them)
};

// The slice epilogue over kTrips trips of touched quads (`n_quads` of them, their numbers in the wave's descriptor
// list; kTrips = ceil(n_quads / 64)).  The caller has reset the dummy row.
template <int W, typename CountT, int kTrips>
__device__ __forceinline__ void slice_epilogue_trips(const TeamParams *__restrict__ ktp, WaveLds<CountT> lds, uint32_t n_kmers,
                                                     SliceArgs a, uint32_t n_quads)
{
    [[maybe_unused]] const PlaceParams &p = ktp->base;  // (diagnostic builds)
    typedef QuadCounts<CountT> Counts;
    const uint32_t lane = here_lane();
    TeamCtx<W, true> ctx;
    ctx.base_ = a.base;
    ctx.cand = static_cast<v4u *>(ktp->slice_rows_out) + (uint64_t)a.slice_at * a.keep;
    ctx.partial = static_cast<TeamPartial *>(ktp->slice_sums_out) + a.slice_at;
    ctx.trace_at_ = a.trace_at;
#ifdef EPIK_AMD_ABLATION
    // (the wave whose timeline is recorded: entry `slot` of the ten this call owns, tools/trace_summary.py)
#define QEPI_STAMP(slot, code)                                                        \
    if (ctx.trace_at_ != 0xffffffffu && lane == 0) {                                  \
        const uint32_t i_ = ctx.trace_at_ + (uint32_t)(slot);                         \
        if (i_ < 100000u) {                                                           \
            p.dbg[64 + 2 * (size_t)i_] = (unsigned long long)(code);                  \
            p.dbg[65 + 2 * (size_t)i_] = __builtin_amdgcn_s_memtime();                \
        }                                                                             \
    }
#else
#define QEPI_STAMP(slot, code)
#endif
    const uint32_t score_addr = (uint32_t)(uintptr_t)lds.score, count_addr = (uint32_t)(uintptr_t)lds.count;
    const uint32_t list_addr = (uint32_t)(uintptr_t)lds.desc;
    const uint32_t keep = a.keep;
    const float k_f = (float)a.kmer_size;
    const float log_thr = a.log_threshold;
    // ---- the one pass through LDS: quad numbers; counts and scores in, zeros out --------------------------------
    uint32_t quad[kTrips];
    bool has[kTrips];
#pragma unroll
    for (int c = 0; c < kTrips; ++c) {
        has[c] = (uint32_t)c * 64u + lane < n_quads;  // (the last trip may be a partial one)
        quad[c] = 0u;
        if (has[c]) quad[c] = *(lds_u16 *)(uintptr_t)(list_addr + 2u * ((uint32_t)c * 64u + lane));
    }
    lds_phase();
    Counts cw[kTrips];
    float s[kTrips][4];
#pragma unroll
    for (int c = 0; c < kTrips; ++c) {
#pragma unroll
        for (int i = 0; i < Counts::kWords; ++i) cw[c].w[i] = 0u;
        s[c][0] = s[c][1] = s[c][2] = s[c][3] = 0.0f;
        if (has[c]) {
            const uint32_t c_at = count_addr + quad[c] * Counts::kBytes;
            lds_f32x4 *s_at = (lds_f32x4 *)(uintptr_t)(score_addr + 16u * quad[c]);
            cw[c].load(c_at);
            const v4f v = *s_at;
            Counts::reset(c_at);  // place.cpp:335-342 for the wave's next read
            *(lds_u32x4 *)s_at = v4u{0u, 0u, 0u, 0u};
            s[c][0] = v.x, s[c][1] = v.y, s[c][2] = v.z, s[c][3] = v.w;
        }
    }
    QEPI_STAMP(2, 21)  // quads, counts and scores in
    // ---- score correction (:418-422), as place_epilogue_body's correct_rows ------------------------------------
    const float nk_f = (float)n_kmers;
    const float inv_k = __fdiv_rn(1.0f, k_f);
    // x / k, correctly rounded for k <= 32 (epik_amd_placer_create refuses anything else) and |x| >= 2^-102
    // (tools/test_div.hip); a trip that meets a smaller |x| is redone with the division
    auto div_k = [&](float x) {
        const float q = __fmul_rn(x, inv_k);
        const float r = __fmaf_rn(-q, k_f, x);
        return __fmaf_rn(r, inv_k, q);
    };
    uint32_t touched = 0;
    float lane_best_f = -INFINITY;
#pragma unroll
    for (int c = 0; c < kTrips; ++c) {
        float pre[4], corrected[4];
        bool edge[4];
        float smallest = INFINITY;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // (the count straight to float32; nk - c is exact there, both below 2^16: float(nk - c) of :420)
            const float c_f = (float)cw[c].count(u);
            edge[u] = c_f != 0.0f;
            pre[u] = __fadd_rn(s[c][u], __fmul_rn(nk_f - c_f, log_thr));  // :420
            corrected[u] = div_k(pre[u]);                                  // :421
            smallest = fminf(smallest, fabsf(pre[u]));
        }
        if (__builtin_expect(__ballot(smallest < 0x1p-100f) != 0, 0)) {  // practically never
#pragma unroll
            for (int u = 0; u < 4; ++u) corrected[u] = __fdiv_rn(pre[u], k_f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s[c][u] = edge[u] ? corrected[u] : -INFINITY;  // -inf = "not an edge"
            touched += (uint32_t)__popcll(__ballot(edge[u]));
            lane_best_f = fmaxf(lane_best_f, s[c][u]);
        }
    }
    QEPI_STAMP(3, 22)  // correction
    const float thr_score = __fdiv_rn(__fmul_rn(nk_f, log_thr), k_f);  // :175 / :146-147
    // ---- select_best_placements (:134-159): tau, then the candidates --------------------------------------------
    // (touched != 0: a listed quad holds a count)
    const uint32_t n_sel = keep < touched ? keep : touched;  // :137
    const uint32_t lane_best = lane_best_f == -INFINITY ? 0u : ord_f32(lane_best_f);
    const uint32_t top = wave_max_u32(lane_best);
    const uint32_t tau = tau_of_lane_maxima(lane_best, top, n_sel);
    const float best_score = unord_f32(top);
    QEPI_STAMP(4, 23)  // tau
    const float ref_score = fmaxf(best_score, thr_score);
    const bool relative_sum = ref_score > -280.0f;
    const bool all_underflow = ref_score < -325.0f;
    constexpr float kLog2Of10 = 3.32192809488736f;
    constexpr uint32_t kCandCap = kTeamCandCap;
    auto *cand = reinterpret_cast<lds_v2u *>(lds.desc);  // {ord(score), row | count << 16} (the quad list is in registers by now)
    const float tau_f = tau <= 1u ? -FLT_MAX : unord_f32(tau);
    float rel_sum = 0.0f, rel_sum_b = 0.0f;
    uint32_t n_cand = 0;
#pragma unroll
    for (int c = 0; c < kTrips; ++c) {
        // sum_scores' terms relative to the largest, float32 (a reference point that could underflow: below)
        float term[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) term[u] = __builtin_amdgcn_exp2f(__fmul_rn(__fsub_rn(s[c][u], ref_score), kLog2Of10));
        rel_sum += term[0] + term[2];
        rel_sum_b += term[1] + term[3];
        const bool any_cand = fmaxf(fmaxf(s[c][0], s[c][1]), fmaxf(s[c][2], s[c][3])) >= tau_f;
        if (__ballot(any_cand) != 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool is_cand = s[c][u] >= tau_f;
                const uint64_t m = __ballot(is_cand);
                if (m) {
                    const uint32_t slot = n_cand + lanes_below(m);
                    if (is_cand && slot < kCandCap) cand[slot] = v2u{ord_f32(s[c][u]), (4u * quad[c] + (uint32_t)u) | (cw[c].count(u) << 16)};
                    n_cand += (uint32_t)__popcll(m);
                }
            }
        }
    }
    rel_sum += rel_sum_b;
    QEPI_STAMP(5, 24)  // terms and candidates
    // ---- rank: a candidate per lane, rank = number of candidates with a larger key (score desc, row asc) ------
    uint64_t my_key = 0;
    uint32_t my_count = 0, my_rank = 0;
    if (n_cand > kCandCap) {
        // Too many ties at tau for the candidate buffer: repeated selection over all edges (slow, rare).  Lane r
        // ends up with the row of rank r.
        uint64_t prev = ~0ull;
        for (uint32_t r = 0; r < n_sel; ++r) {
            uint64_t best = 0;
            uint32_t best_count = 0;
#pragma unroll
            for (int c = 0; c < kTrips; ++c) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint64_t key = ((uint64_t)ord_f32(s[c][u]) << 32) | (uint64_t)(~(4u * quad[c] + (uint32_t)u));
                    if (s[c][u] != -INFINITY && key < prev && key > best) best = key, best_count = cw[c].count(u);
                }
            }
            uint64_t wave_best = best;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const uint64_t o = shfl_xor_u64(wave_best, m);
                wave_best = o > wave_best ? o : wave_best;
            }
            const uint64_t owner = __ballot(best == wave_best);  // (keys are distinct: one lane)
            const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane(best_count, __builtin_ctzll(owner));
            if (lane == r) my_key = wave_best, my_count = cnt, my_rank = r;
            prev = wave_best;
        }
    } else {
        if (lane < n_cand) {
            const v2u e = cand[lane];
            my_key = ((uint64_t)e.x << 32) | (uint64_t)(~(e.y & 0xffffu));
            my_count = e.y >> 16;
        }
        uint32_t rank_b = 0;
        uint32_t j = 0;
        for (; j + 2 <= n_cand; j += 2) {  // two candidates per turn, out of the lanes
            const uint64_t ka = readlane_u64(my_key, (int)j), kb = readlane_u64(my_key, (int)j + 1);
            my_rank += ka > my_key ? 1u : 0u;
            rank_b += kb > my_key ? 1u : 0u;
        }
        if (j < n_cand) my_rank += readlane_u64(my_key, (int)j) > my_key ? 1u : 0u;
        my_rank += rank_b;
    }
    double abs_sum = 0.0;
    if (!relative_sum && !all_underflow) {
        // Everything in double, term by term, as place.cpp:178-182 (cold).  A loop cannot index registers: the quads'
        // corrected scores go back to LDS -- the quad of lane l of trip c to quad 64 c + l, below the slice's number
        // of quads and zero by now like every touched quad; -inf where there is no edge --, are summed from there and
        // reset again.
#pragma unroll
        for (int c = 0; c < kTrips; ++c)
            if (has[c]) *(lds_f32x4 *)(uintptr_t)(score_addr + 16u * ((uint32_t)c * 64u + lane)) = v4f{s[c][0], s[c][1], s[c][2], s[c][3]};
        lds_phase();
        abs_sum = sum_of_powers(score_addr, n_quads * 4u);
        lds_phase();
#pragma unroll
        for (int c = 0; c < kTrips; ++c)
            if (has[c]) *(lds_u32x4 *)(uintptr_t)(score_addr + 16u * ((uint32_t)c * 64u + lane)) = v4u{0u, 0u, 0u, 0u};
    }
    // ---- this slice's share of sum_scores (:164-184), ranked rows and partial sum to the merge ------------------
    const double sum = wave_sum_f64(relative_sum ? (double)rel_sum : abs_sum);
    if (my_key != 0 && my_rank < n_sel)
        ctx.cand[my_rank] = v4u{(uint32_t)(my_key >> 32), a.base + ~(uint32_t)my_key, my_count, 0u};
    for (uint32_t r = n_sel + lane; r < keep; r += kWave) ctx.cand[r] = v4u{0u, 0u, 0u, 0u};
    if (lane == 0) {
        ctx.partial->touched = touched;
        ctx.partial->relative = relative_sum ? 1u : 0u;
        ctx.partial->ref_score = ref_score;
        ctx.partial->sum = sum;
    }
    QEPI_STAMP(6, 25)  // rank, publish
#undef QEPI_STAMP
}

// kSparseTrips: up to so many trips of touched quads (their list: 16 bits each in the wave's descriptor list of
// kTeamDescCap + kTeamRing entries of 8 bytes)
constexpr int kSparseTrips = 4;
static_assert(kSparseTrips * 64 * 2 <= (int)((kTeamDescCap + kTeamRing) * 8u), "the quad list lies in the descriptor list");

// The slice epilogue for a slice of which few rows hold a count: at most `max_quads` (<= kSparseTrips * 64) quads.
// False: more do -- nothing has been changed, the dense epilogue is the caller's.  The caller has reset the dummy row.
template <int W, typename CountT>
__device__ __forceinline__ bool slice_epilogue_sparse(const TeamParams *__restrict__ ktp, WaveLds<CountT> lds, uint32_t n_kmers, SliceArgs a,
                                                      uint32_t max_quads)
{
    const uint32_t rows_pad = here(a.rows_pad);
    const uint32_t n_quads = list_touched_quads<CountT>((uint32_t)(uintptr_t)lds.count, rows_pad, (uint32_t)(uintptr_t)lds.desc, max_quads);
    if (n_quads > max_quads) return false;
    lds_phase();
    if (n_quads == 0) {  // (the caller usually knows before it comes here)
        TeamCtx<W, true> ctx;
        ctx.cand = static_cast<v4u *>(ktp->slice_rows_out) + (uint64_t)a.slice_at * a.keep;
        ctx.partial = static_cast<TeamPartial *>(ktp->slice_sums_out) + a.slice_at;
        publish_empty_slice(ctx, n_kmers, a.kmer_size, a.log_threshold, a.keep);
    } else if (n_quads <= 64u) {
        slice_epilogue_trips<W, CountT, 1>(ktp, lds, n_kmers, a, n_quads);
    } else if (n_quads <= 128u) {
        slice_epilogue_trips<W, CountT, 2>(ktp, lds, n_kmers, a, n_quads);
    } else if (n_quads <= 192u) {
        slice_epilogue_trips<W, CountT, 3>(ktp, lds, n_kmers, a, n_quads);
    } else {
        slice_epilogue_trips<W, CountT, 4>(ktp, lds, n_kmers, a, n_quads);
    }
    return true;
}

// The same for the partial list of a k-mer-space shard (emit_partial_list, team_device.hpp): the rows of the wave's
// slice that received a k-mer go to `out` (room for `cap` entries; never more are written) and are reset.
// *n_out = how many rows had received one.  False: too many quads for the list of them, nothing has been changed.
template <typename CountT, int kTrips>
__device__ __forceinline__ uint32_t emit_partial_list_trips(WaveLds<CountT> lds, uint32_t n_quads, uint8_t *__restrict__ out, uint32_t cap)
{
    typedef PartialEntry<CountT> Entry;
    typedef QuadCounts<CountT> Counts;
    const uint32_t lane = here_lane();
    const uint32_t score_addr = (uint32_t)(uintptr_t)lds.score, count_addr = (uint32_t)(uintptr_t)lds.count;
    const uint32_t list_addr = (uint32_t)(uintptr_t)lds.desc;
    auto *dst = reinterpret_cast<typename Entry::raw_t *>(out);
    uint32_t quad[kTrips];
    bool has[kTrips];
#pragma unroll
    for (int c = 0; c < kTrips; ++c) {
        has[c] = (uint32_t)c * 64u + lane < n_quads;
        quad[c] = 0u;
        if (has[c]) quad[c] = *(lds_u16 *)(uintptr_t)(list_addr + 2u * ((uint32_t)c * 64u + lane));
    }
    lds_phase();
    Counts cw[kTrips];
    float s[kTrips][4];
#pragma unroll
    for (int c = 0; c < kTrips; ++c) {
#pragma unroll
        for (int i = 0; i < Counts::kWords; ++i) cw[c].w[i] = 0u;
        s[c][0] = s[c][1] = s[c][2] = s[c][3] = 0.0f;
        if (has[c]) {
            const uint32_t c_at = count_addr + quad[c] * Counts::kBytes;
            lds_f32x4 *s_at = (lds_f32x4 *)(uintptr_t)(score_addr + 16u * quad[c]);
            cw[c].load(c_at);
            const v4f v = *s_at;
            Counts::reset(c_at);
            *(lds_u32x4 *)s_at = v4u{0u, 0u, 0u, 0u};
            s[c][0] = v.x, s[c][1] = v.y, s[c][2] = v.z, s[c][3] = v.w;
        }
    }
    uint32_t n = 0;
#pragma unroll
    for (int c = 0; c < kTrips; ++c) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t cnt = cw[c].count(u);
            const bool hit = cnt != 0u;
            const uint64_t m = __ballot(hit);
            if (m) {
                const uint32_t slot = n + lanes_below(m);
                if (hit && slot < cap) dst[slot] = Entry::make(__float_as_uint(s[c][u]), 4u * quad[c] + (uint32_t)u, cnt);
                n += (uint32_t)__popcll(m);
            }
        }
    }
    return n;
}
// The same for an item whose stream was ONE trip of the ring (a shard of many holds two or three of a (read, slice)
// item's lists: most of its items): the rows it touched are the rows of its postings, and the ring still holds their
// cells (stream_round's cells_out: slot i = chunk i, cell 0 behind a chunk's end and in padding slots).  No pass over
// the slice's counts, no list of quads: chunk by chunk -- two chunks may name the same row, and the first to come
// takes it and leaves a zero count behind -- the rows that hold a count go out and are reset.
template <typename CountT, int kDepth>
__device__ __forceinline__ uint32_t emit_partial_list_cells(WaveLds<CountT> lds, uint32_t rows_pad, const uint32_t (&cells)[kDepth],
                                                            uint32_t n_chunks, uint8_t *__restrict__ out, uint32_t cap)
{
    typedef WaveLds<CountT> Lds_t;
    typedef PartialEntry<CountT> Entry;
    auto *dst = reinterpret_cast<typename Entry::raw_t *>(out);
    uint32_t n = 0;
#pragma unroll
    for (int c = 0; c < kDepth; ++c) {
        if ((uint32_t)c < n_chunks) {  // wave-uniform
            const uint32_t row = rows_pad - 1u - cells[c];  // cell 0: the dummy row
            const uint2 cv = lds.load(row);
            const uint32_t count = cv.y & ~(uint32_t)Lds_t::kSeen;
            const bool hit = cells[c] != 0u && count != 0u;
            const uint64_t m = __ballot(hit);
            if (m) {
                const uint32_t slot = n + lanes_below(m);
                if (hit && slot < cap) dst[slot] = Entry::make(cv.x, row, count);
                if (hit) lds.store(row, 0u, 0u);
                n += (uint32_t)__popcll(m);
            }
        }
    }
    return n;
}

template <typename CountT>
__device__ __forceinline__ bool emit_partial_list_sparse(WaveLds<CountT> lds, uint32_t rows_pad, uint32_t max_quads, uint8_t *__restrict__ out,
                                                         uint32_t cap, uint32_t *n_out)
{
    rows_pad = here(rows_pad);
    const uint32_t n_quads = list_touched_quads<CountT>((uint32_t)(uintptr_t)lds.count, rows_pad, (uint32_t)(uintptr_t)lds.desc, max_quads);
    if (n_quads > max_quads) return false;
    lds_phase();
    if (n_quads == 0)
        *n_out = 0u;
    else if (n_quads <= 64u)
        *n_out = emit_partial_list_trips<CountT, 1>(lds, n_quads, out, cap);
    else if (n_quads <= 128u)
        *n_out = emit_partial_list_trips<CountT, 2>(lds, n_quads, out, cap);
    else if (n_quads <= 192u)
        *n_out = emit_partial_list_trips<CountT, 3>(lds, n_quads, out, cap);
    else
        *n_out = emit_partial_list_trips<CountT, 4>(lds, n_quads, out, cap);
    return true;
}

}  // namespace
}  // namespace epik_amd
#endif
