// team_device.hpp -- device-side pieces shared by the team kernels (team_kernel.hip: the whole
// placement of a read in one workgroup; team_stream.hip: the same with the front end taken out into
// a kernel of its own): the table entry of the sliced database, the slice context of a wave, and
// the merge of the slices' results.  Anonymous namespace: each translation unit gets its own copy.
#ifndef EPIK_AMD_TEAM_DEVICE_HPP
#define EPIK_AMD_TEAM_DEVICE_HPP
#include <type_traits>

#include "place_device.hpp"

namespace epik_amd {

namespace {

typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) v4u lds_u32x4;
typedef __attribute__((address_space(3))) TeamPartial lds_partial;

// Table entry of the team layout: {u32 line, u16 len[W]}, padded to 16 / 32 / 64 bytes.  The W
// sublists lie back to back from byte line * 128 on, each in chunks of <= 64 postings (f32 score[cnt]
// then u16 cell[cnt], cell local to the slice, 0 = the slice's dummy row) and padded to 4 bytes.
template <int W>
struct TeamEntry {
    static constexpr int kWords = team_entry_bytes(W) / 4;
    uint32_t line;
    uint32_t len[W];
    // the entry as it lies in HBM -- kQuads pieces of 16 bytes, or (two slices per pass) one of 8 -- and its fields from that
    typedef std::conditional_t<(kWords >= 4), uint4, uint2> raw_t;
    static constexpr int kRawWords = (int)(sizeof(raw_t) / 4);
    static constexpr int kQuads = kWords / kRawWords;
    __device__ static __forceinline__ raw_t zero_raw()
    {
        if constexpr (kRawWords == 4)
            return make_uint4(0u, 0u, 0u, 0u);
        else
            return make_uint2(0u, 0u);
    }
    // `position`: where in the read the k-mer starts (its parity picks the form a paired table holds it in;
    // a lookup by code alone passes 0)
    __device__ static __forceinline__ void fetch(const TeamParams &tp, uint32_t pass, uint32_t key, uint32_t position,
                                                 raw_t (&raw)[kQuads])
    {
        uint64_t index = (uint64_t)pass * tp.num_keys + key;
        if (tp.shard_count > 1u) {  // (wave-uniform) a k-mer-space shard: the codes of the other shards are not looked up
            const uint32_t q = key / tp.shard_count;
            if (key - q * tp.shard_count != tp.shard_index) {
#pragma unroll
                for (int i = 0; i < kQuads; ++i) raw[i] = zero_raw();
                return;
            }
            index = (uint64_t)pass * tp.num_keys + q;
        } else if (tp.team_paired) {  // (wave-uniform)
            const uint32_t shift = 2u * tp.base.kmer_size - 2u;  // X = k-1 letters of 2 bits
            const bool as_prefix = (position & 1u) != 0;
            const uint32_t block = as_prefix ? key >> 2 : key & ((1u << shift) - 1u);
            const uint32_t slot = as_prefix ? 4u + (key & 3u) : key >> shift;
            index = ((uint64_t)pass * (tp.num_keys / 4u) + block) * 8u + slot;
        }
        const raw_t *e = reinterpret_cast<const raw_t *>(tp.team_table + index * (kWords * 4u));
#pragma unroll
        for (int i = 0; i < kQuads; ++i) raw[i] = e[i];
    }
    __device__ __forceinline__ void unpack(const raw_t (&raw)[kQuads])
    {
        uint32_t w[kWords];
#pragma unroll
        for (int i = 0; i < kQuads; ++i) {
            w[kRawWords * i] = raw[i].x;
            w[kRawWords * i + 1] = raw[i].y;
            if constexpr (kRawWords == 4) {
                w[4 * i + 2] = raw[i].z;
                w[4 * i + 3] = raw[i].w;
            }
        }
        line = w[0];
#pragma unroll
        for (int s = 0; s < W; ++s) len[s] = (w[1 + s / 2] >> (16 * (s & 1))) & 0xffffu;
    }
    __device__ __forceinline__ void load(const TeamParams &tp, uint32_t pass, uint32_t key, uint32_t position = 0u)
    {
        raw_t raw[kQuads];
        fetch(tp, pass, key, position, raw);
        unpack(raw);
    }
    // byte offset of sublist s in the posting region
    __device__ __forceinline__ uint64_t start(int s) const
    {
        uint32_t off = 0;
#pragma unroll
        for (int q = 0; q < W; ++q)
            if (q < s) off += (len[q] * 6u + 3u) & ~3u;
        return (uint64_t)line * 128u + off;
    }
};

// The slice of the branch range one wave of a team accumulates (see WaveCtx in place_device.hpp).
// kGlobalOut: the slice's results go to HBM (team_stream_kernel: a kernel of its own merges them) instead of
// the workgroup's merge area in LDS.
template <int W, bool kGlobalOut = false>
struct TeamCtx {
    static constexpr bool kTeam = true;
    static constexpr uint32_t kCandCap = kTeamCandCap;  // top-k candidates of a slice: at most one per lane
    uint32_t rows_pad_, rows_, base_, slice_, pass_;
    uint32_t kmer_size_, keep_;  // placer constants, in registers: the slice epilogue loads nothing from the argument block
    float log_threshold_;
    __device__ __forceinline__ uint32_t kmer_size(const PlaceParams &) const { return kmer_size_; }
    __device__ __forceinline__ float log_threshold(const PlaceParams &) const { return log_threshold_; }
    __device__ __forceinline__ uint32_t keep_at_most(const PlaceParams &) const { return keep_; }
    std::conditional_t<kGlobalOut, v4u *, lds_u32x4 *> cand;             // [keep_at_most] ranked rows of this slice for the merge
    std::conditional_t<kGlobalOut, TeamPartial *, lds_partial *> partial;  // this slice's share of sum_scores
    uint32_t trace_at_ = 0xffffffffu;  // diagnostic builds: the wave whose timeline is recorded writes its next entries here
    bool untouched_ = false;  // no row of the slice holds a count (the caller knows: nothing streamed): no sweep over them
    __device__ __forceinline__ bool untouched() const { return untouched_; }
    __device__ __forceinline__ void before_publish() const {}
    template <typename Params>
    __device__ __forceinline__ uint32_t rows_pad(const Params &) const { return rows_pad_; }
    template <typename Params>
    __device__ __forceinline__ uint32_t rows(const Params &) const { return rows_; }
    __device__ __forceinline__ uint32_t branch_base() const { return base_; }
    template <typename Layout>
    __device__ __forceinline__ void lookup(const PlaceParams &p, uint32_t key, uint64_t &addr, uint32_t &len) const
    {
        const TeamParams &tp = reinterpret_cast<const TeamParams &>(p);  // PlaceParams is its first member
        TeamEntry<W> e;
        e.load(tp, pass_, key);
        addr = e.line * 128ull;
        len = 0;
#pragma unroll
        for (int s = 0; s < W; ++s) {
            if ((uint32_t)s < slice_) addr += (e.len[s] * 6u + 3u) & ~3u;
            if ((uint32_t)s == slice_) len = e.len[s];
        }
    }
};

typedef PackedLayout<kPlainTable> TeamChunks;  // the chunk format (and its loads) of the packed layout

// ---------------------------------------------------------------------------------
// The merge of a team placement, by one wave: the slices' ranked rows (S * keep_at_most slots of
// {ord(score), branch, count, -}, empty slots 0) ranked together, then exactly the tail of
// place_epilogue: sum_scores (:164-184) from the slices' partial sums, like-weight-ratios
// (:241-264), filter_by_ratio (:188-199), rows out.
// ---------------------------------------------------------------------------------
// the placer constants and output arrays of a launch, handed to team_merge in registers
struct MergeParams {
    uint32_t keep_at_most, kmer_size, num_branches;
    float log_threshold;
    double keep_factor;
    epik_amd_placement *rows;
    uint32_t *n_rows, *kmer_counts;
};

// What the merge of a read reads when its slots and slices fit the lanes of a wave (S * keep_at_most <= 64): a
// slot per lane, a slice's partial sum per lane.  team_merge_kernel asks for the next read's while it works on
// this one's.
struct MergeInputs {
    v4u mine;
    uint32_t touched, relative;
    float ref_score;
    double sum;
};
template <typename CandPtr, typename PartialPtr>
__device__ __forceinline__ MergeInputs load_merge_inputs(CandPtr cand, uint32_t cand_stride, PartialPtr partials,
                                                         uint32_t n_slices, uint32_t keep)
{
    const uint32_t lane = (uint32_t)lane_id();
    MergeInputs in;
    in.mine = v4u{0u, 0u, 0u, 0u};
    in.touched = in.relative = 0u;
    in.ref_score = 0.0f;
    in.sum = 0.0;
    if (lane < n_slices * keep) in.mine = cand[(lane / keep) * cand_stride + lane % keep];
    if (lane < n_slices) {
        in.touched = partials[lane].touched;
        in.relative = partials[lane].relative;
        in.ref_score = partials[lane].ref_score;
        in.sum = partials[lane].sum;
    }
    return in;
}

// kPreloaded: `in` holds the read's inputs (the caller made sure they fit the lanes)
template <bool kPreloaded, typename CandPtr, typename PartialPtr>
__device__ __forceinline__ void team_merge_body(MergeParams p, CandPtr cand, uint32_t cand_stride, PartialPtr partials,
                                                uint32_t n_slices, uint64_t read, uint64_t n_kmers, const MergeInputs &in)
{
    const int lane = lane_id();
    const uint32_t keep = p.keep_at_most;
    const uint32_t M = n_slices * keep;
    const float k_f = (float)p.kmer_size;
    const float thr_score = __fdiv_rn(__fmul_rn((float)n_kmers, p.log_threshold), k_f);  // :175 / :146-147
    constexpr float kLog2Of10 = 3.32192809488736f;
    // the slices' partial sums, one slice per lane (their LDS reads go out together; read one after the
    // other they were a chain of round trips on the path of the next read's first barrier)
    uint32_t my_touched = 0, my_relative = 0;
    float my_ref = 0.0f;
    double my_sum = 0.0;
    const bool sums_in_lanes = kPreloaded || n_slices <= (uint32_t)kWave;  // wave-uniform; else looped over below
    if constexpr (kPreloaded) {
        my_touched = in.touched, my_relative = in.relative, my_ref = in.ref_score, my_sum = in.sum;
    } else if (sums_in_lanes && (uint32_t)lane < n_slices) {
        my_touched = partials[lane].touched;
        my_relative = partials[lane].relative;
        my_ref = partials[lane].ref_score;
        my_sum = partials[lane].sum;
    }
    uint32_t touched = 0;
    if (sums_in_lanes)
        touched = wave_sum_u32(my_touched);
    else
        for (uint32_t s = 0; s < n_slices; ++s) touched += partials[s].touched;
    uint32_t n_sel;
    float best_score;
    // Up to 64 slots (4 or 8 slices of the default 7 rows): one slot per lane, everything in registers.
    const bool in_lanes = kPreloaded || M <= (uint32_t)kWave;  // wave-uniform
    v4u mine = v4u{0u, 0u, 0u, 0u};             // in_lanes: this lane's slot, .w = its rank
    int best_lane = 0;
    if (touched == 0) {  // :141-152: first keep_at_most branches at the threshold score
        n_sel = keep;
        best_score = thr_score;
        if (in_lanes) {
            if ((uint32_t)lane < keep) mine = v4u{ord_f32(thr_score), (uint32_t)lane, 0u, (uint32_t)lane};
        } else {
            for (uint32_t i = lane; i < M; i += kWave)
                cand[(i / keep) * cand_stride + i % keep] =
                    i < keep ? v4u{ord_f32(thr_score), i, 0u, i} : v4u{0u, 0u, 0u, 0u};
        }
    } else if (in_lanes) {
        n_sel = keep < touched ? keep : touched;  // :137
        if constexpr (kPreloaded)
            mine = in.mine;
        else if ((uint32_t)lane < M)
            mine = cand[((uint32_t)lane / keep) * cand_stride + (uint32_t)lane % keep];
        const uint64_t key = mine.x ? (((uint64_t)mine.x << 32) | (uint64_t)(~mine.y)) : 0ull;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < M; ++j) rank += readlane_u64(key, (int)j) > key ? 1u : 0u;
        mine.w = rank;
        const uint64_t first = __ballot(key != 0 && rank == 0);  // never empty: touched != 0
        best_lane = __builtin_ctzll(first);
        best_score = unord_f32(__builtin_amdgcn_readlane(mine.x, best_lane));
    } else {
        n_sel = keep < touched ? keep : touched;  // :137
        uint32_t best_ord = 0;
        for (uint32_t i0 = 0; i0 < M; i0 += kWave) {
            const uint32_t i = i0 + (uint32_t)lane;
            uint64_t key = 0;
            if (i < M) {
                const v4u c = cand[(i / keep) * cand_stride + i % keep];
                key = c.x ? (((uint64_t)c.x << 32) | (uint64_t)(~c.y)) : 0ull;
            }
            uint32_t rank = 0;
            for (uint32_t j = 0; j < M; ++j) {  // the same address in every lane: an LDS broadcast
                const v4u c = cand[(j / keep) * cand_stride + j % keep];
                const uint64_t kj = c.x ? (((uint64_t)c.x << 32) | (uint64_t)(~c.y)) : 0ull;
                rank += kj > key ? 1u : 0u;
            }
            if (i < M) cand[(i / keep) * cand_stride + i % keep].w = rank;
            const uint64_t first = __ballot(key != 0 && rank == 0);
            if (first) best_ord = __builtin_amdgcn_readlane((uint32_t)(key >> 32), __builtin_ctzll(first));
        }
        best_score = unord_f32(best_ord);
    }
    // 10^score of the rows that may be reported (:254): once per row, the lanes side by side
    const bool my_row = in_lanes && mine.x != 0 && mine.w < n_sel;
    double my_power = 0.0, best_power;
    if (in_lanes) {
        my_power = my_row ? pow10_f64((double)unord_f32(mine.x)) : 0.0;
        best_power = __longlong_as_double((long long)readlane_u64((uint64_t)__double_as_longlong(my_power), best_lane));
    } else {
        best_power = pow10_f64((double)best_score);
    }
    // ---- sum_scores (:164-184) ------------------------------------------------------------------
    const float ref_score = fmaxf(best_score, thr_score);
    double score_sum;
    {
        double rel = 0.0, absolute = 0.0;
        if (sums_in_lanes) {
            const bool counts = my_touched != 0;
            const double scaled = my_sum * (double)__builtin_amdgcn_exp2f(__fmul_rn(__fsub_rn(my_ref, ref_score), kLog2Of10));
            rel = wave_sum_f64(counts && my_relative ? scaled : 0.0);
            absolute = wave_sum_f64(counts && !my_relative ? my_sum : 0.0);
        } else {
            for (uint32_t s = 0; s < n_slices; ++s) {
                if (partials[s].touched == 0) continue;
                const double sum = partials[s].sum;
                if (partials[s].relative)
                    rel += sum * (double)__builtin_amdgcn_exp2f(__fmul_rn(__fsub_rn(partials[s].ref_score, ref_score), kLog2Of10));
                else
                    absolute += sum;
            }
        }
        const float not_placed = (float)p.num_branches - (float)touched;  // :174
        if (ref_score > -280.0f) {
            if (not_placed != 0.0f)
                rel += (double)(not_placed * __builtin_amdgcn_exp2f(__fmul_rn(__fsub_rn(thr_score, ref_score), kLog2Of10)));
            const double ref_power = (ref_score == best_score) ? best_power : pow10_f64((double)ref_score);
            score_sum = ref_power * rel + absolute;
        } else {
            score_sum = (double)not_placed * pow10_f64((double)thr_score) + absolute;  // :174-183, all double
        }
    }
    const double keep_factor = (score_sum == 0.0) ? 0.0 : p.keep_factor;  // :247-251
    const double best_ratio = (score_sum == 0.0 || best_power == 0.0) ? 0.0 : best_power / score_sum;  // :191
    const double ratio_threshold = best_ratio * keep_factor;                                           // :192
    // ---- LWR (:241-264), filter_by_ratio (:188-199): which ranks stay --------------------------
    uint64_t kept_ranks = 0;
    if (in_lanes) {
        const double lwr = (my_row && score_sum != 0.0 && my_power != 0.0) ? my_power / score_sum : 0.0;  // :255-262
        kept_ranks = wave_or_u64((my_row && lwr >= ratio_threshold) ? 1ull << mine.w : 0ull);            // :197
        if (my_row && ((kept_ranks >> mine.w) & 1ull)) {
            const uint32_t slot = (uint32_t)__popcll(kept_ranks & ((1ull << mine.w) - 1ull));
            epik_amd_placement out;
            out.branch = mine.y;
            out.score = unord_f32(mine.x);
            out.lwr = lwr;
            p.rows[read * keep + slot] = out;
            if (p.kmer_counts) p.kmer_counts[read * keep + slot] = mine.z;
        }
    } else {
        for (uint32_t i0 = 0; i0 < M; i0 += kWave) {
            const uint32_t i = i0 + (uint32_t)lane;
            if (i < M) {
                const v4u c = cand[(i / keep) * cand_stride + i % keep];
                if (c.x != 0 && c.w < n_sel) {
                    const double power = pow10_f64((double)unord_f32(c.x));
                    const double lwr = (score_sum != 0.0 && power != 0.0) ? power / score_sum : 0.0;  // :255-262
                    if (lwr >= ratio_threshold) kept_ranks |= 1ull << c.w;                              // :197
                }
            }
        }
        kept_ranks = wave_or_u64(kept_ranks);
        for (uint32_t i0 = 0; i0 < M; i0 += kWave) {
            const uint32_t i = i0 + (uint32_t)lane;
            if (i < M) {
                const v4u c = cand[(i / keep) * cand_stride + i % keep];
                if (c.x != 0 && c.w < n_sel && ((kept_ranks >> c.w) & 1ull)) {
                    const uint32_t slot = (uint32_t)__popcll(kept_ranks & ((1ull << c.w) - 1ull));
                    const double power = pow10_f64((double)unord_f32(c.x));
                    epik_amd_placement out;
                    out.branch = c.y;
                    out.score = unord_f32(c.x);
                    out.lwr = (score_sum != 0.0 && power != 0.0) ? power / score_sum : 0.0;
                    p.rows[read * keep + slot] = out;
                    if (p.kmer_counts) p.kmer_counts[read * keep + slot] = c.z;
                }
            }
        }
    }
    if (lane == 0) p.n_rows[read] = (uint32_t)__popcll(kept_ranks);
}
template <typename CandPtr, typename PartialPtr>
__device__ __attribute__((noinline)) void team_merge(MergeParams p, CandPtr cand, uint32_t cand_stride,
                                                     PartialPtr partials, uint32_t n_slices, uint64_t read,
                                                     uint64_t n_kmers)
{
    team_merge_body<false>(p, cand, cand_stride, partials, n_slices, read, n_kmers, MergeInputs{});
}



// ---------------------------------------------------------------------------------
// The same merge with SEVERAL reads to a wave (round 5).  A read's merge has S * keep_at_most slots -- 14 with two
// slices per pass, 28 with four -- and team_merge_body gives it a whole wave: 50 of 64 lanes idle through the one
// expensive thing in it, the double-precision 10^score of the rows that may be reported.  Here a read gets a GROUP of
// L = 16 or 32 lanes (4 or 2 reads to a wave); everything a read's lanes share is computed by every lane of the
// group from group reductions (DPP inside a row of 16 lanes; one ds_bpermute between the two rows of a group of 32),
// there is no scalar branch on a read's values, and the ranking rotates the keys around the row (row_ror) instead of
// pulling them out of the lanes one by one.  The arithmetic on a read's values is team_merge_body's, operation for
// operation (the sums over the slices add in the same tree): the same rows, the same bits.
// ---------------------------------------------------------------------------------
template <int kCtrl>
__device__ __forceinline__ double dpp_f64(double v)
{
    return __longlong_as_double((long long)dpp_u64<kCtrl>((uint64_t)__double_as_longlong(v)));
}
// reductions over a group of L lanes (L = 16: a DPP row; L = 32: two rows), the result in every lane of the group
template <int L, typename T, typename Op, typename Move>
__device__ __forceinline__ T group_reduce(T v, Op op, Move move)
{
    v = op(v, move(v, std::integral_constant<int, 0xB1>{}));   // quad_perm [1,0,3,2]
    v = op(v, move(v, std::integral_constant<int, 0x4E>{}));   // quad_perm [2,3,0,1]
    v = op(v, move(v, std::integral_constant<int, 0x141>{}));  // row_half_mirror
    v = op(v, move(v, std::integral_constant<int, 0x140>{}));  // row_mirror
    if constexpr (L == 32) {
        T other;
        if constexpr (sizeof(T) == 8) {
            const uint64_t bits = sizeof(T) == 8 ? *reinterpret_cast<const uint64_t *>(&v) : 0ull;
            const uint64_t o = shfl_xor_u64(bits, 16);
            other = *reinterpret_cast<const T *>(&o);
        } else {
            const uint32_t o = (uint32_t)__shfl_xor((int)*reinterpret_cast<const uint32_t *>(&v), 16);
            other = *reinterpret_cast<const T *>(&o);
        }
        v = op(v, other);
    }
    return v;
}
template <int L>
__device__ __forceinline__ uint32_t group_sum_u32(uint32_t v)
{
    return group_reduce<L>(v, [](uint32_t a, uint32_t b) { return a + b; }, [](uint32_t x, auto c) { return dpp_u32<decltype(c)::value>(x); });
}
template <int L>
__device__ __forceinline__ uint32_t group_max_u32(uint32_t v)
{
    return group_reduce<L>(v, [](uint32_t a, uint32_t b) { return a > b ? a : b; }, [](uint32_t x, auto c) { return dpp_u32<decltype(c)::value>(x); });
}
template <int L>
__device__ __forceinline__ uint32_t group_or_u32(uint32_t v)
{
    return group_reduce<L>(v, [](uint32_t a, uint32_t b) { return a | b; }, [](uint32_t x, auto c) { return dpp_u32<decltype(c)::value>(x); });
}
template <int L>
__device__ __forceinline__ double group_sum_f64(double v)
{
    return group_reduce<L>(v, [](double a, double b) { return a + b; }, [](double x, auto c) { return dpp_f64<decltype(c)::value>(x); });
}
template <int L>
__device__ __forceinline__ double group_max_f64(double v)
{
    return group_reduce<L>(v, [](double a, double b) { return a > b ? a : b; }, [](double x, auto c) { return dpp_f64<decltype(c)::value>(x); });
}
// how many keys of the lane's group are larger than its own: the keys of the row rotated past it (row_ror:1 .. 15),
// then (L = 32) the other row's
template <int L>
__device__ __forceinline__ uint32_t group_rank(uint64_t key)
{
    uint32_t rank = 0;
    auto pass = [&](uint64_t v, auto first) {
        if constexpr (decltype(first)::value) rank += v > key ? 1u : 0u;
        [&]<int... N>(std::integer_sequence<int, N...>) {
            ((rank += dpp_u64<0x121 + N>(v) > key ? 1u : 0u), ...);  // row_ror:1 .. row_ror:15
        }(std::make_integer_sequence<int, 15>{});
    };
    pass(key, std::false_type{});
    if constexpr (L == 32) pass(shfl_xor_u64(key, 16), std::true_type{});
    return rank;
}

// what a lane of a read's group holds of the read: its slot of the slices' ranked rows, and (the first S lanes) a slice's share
struct PackedMergeInputs {
    uint32_t flags, len;
    v4u mine;
    uint32_t touched, relative;
    float ref_score;
    double sum;
};
template <int L>
__device__ __forceinline__ PackedMergeInputs load_packed_merge_inputs(const TeamParams &tp, const v4u *rows_out, const TeamPartial *sums_out,
                                                                     uint64_t read, bool valid, uint32_t n_slices, uint32_t keep)
{
    const uint32_t slot = (uint32_t)lane_id() % (uint32_t)L;
    PackedMergeInputs in;
    in.flags = kFrontSlow, in.len = 0;  // (a group behind the batch's end: nothing to merge)
    in.mine = v4u{0u, 0u, 0u, 0u};
    in.touched = in.relative = 0u;
    in.ref_score = 0.0f;
    in.sum = 0.0;
    if (valid) {
        const uint32_t *hdr = reinterpret_cast<const uint32_t *>(tp.front_hdr + read * tp.front_hdr_stride);
        in.flags = hdr[1], in.len = hdr[2];
        if (slot < n_slices * keep) in.mine = rows_out[read * n_slices * keep + slot];
        if (slot < n_slices) {
            const TeamPartial &pt = sums_out[read * n_slices + slot];
            in.touched = pt.touched, in.relative = pt.relative, in.ref_score = pt.ref_score, in.sum = pt.sum;
        }
    }
    return in;
}

// one read per group of L lanes (S * keep_at_most <= L): team_merge_body<true>, lane for lane
template <int L>
__device__ __forceinline__ void team_merge_packed(const MergeParams &p, uint64_t read, bool valid, const PackedMergeInputs &in)
{
    const uint32_t slot = (uint32_t)lane_id() % (uint32_t)L;
    const uint32_t keep = p.keep_at_most;
    // reads without rows to merge (team_merge_kernel: settled)
    if (valid && (in.flags & (kFrontNoRows | kFrontTooNarrow)) && !(in.flags & kFrontSlow) && slot == 0)
        p.n_rows[read] = (in.flags & kFrontNoRows) ? 0u : kCountsTooNarrow;
    const bool live = valid && !(in.flags & (kFrontSlow | kFrontNoRows | kFrontTooNarrow));  // the same in a group's lanes
    const uint64_t n_kmers = (uint64_t)in.len - p.kmer_size + 1u;
    const float k_f = (float)p.kmer_size;
    const float thr_score = __fdiv_rn(__fmul_rn((float)n_kmers, p.log_threshold), k_f);  // :175 / :146-147
    constexpr float kLog2Of10 = 3.32192809488736f;
    const uint32_t touched = group_sum_u32<L>(in.touched);
    v4u mine = in.mine;
    uint32_t n_sel = keep < touched ? keep : touched;  // :137
    if (touched == 0) {  // :141-152: first keep_at_most branches at the threshold score
        n_sel = keep;
        mine = slot < keep ? v4u{ord_f32(thr_score), slot, 0u, 0u} : v4u{0u, 0u, 0u, 0u};
    }
    const uint64_t key = mine.x ? (((uint64_t)mine.x << 32) | (uint64_t)(~mine.y)) : 0ull;
    const uint32_t rank = group_rank<L>(key);
    const float best_score = touched == 0 ? thr_score : unord_f32(group_max_u32<L>(mine.x));  // the row of rank 0
    // 10^score of the rows that may be reported (:254): once per row, the lanes side by side
    const bool my_row = live && mine.x != 0 && rank < n_sel;
    const double my_power = my_row ? pow10_f64((double)unord_f32(mine.x)) : 0.0;
    const double best_power = group_max_f64<L>(my_power);  // (the largest score's)
    // ---- sum_scores (:164-184) ------------------------------------------------------------------
    const float ref_score = fmaxf(best_score, thr_score);
    const bool counts = in.touched != 0;
    const double scaled = in.sum * (double)__builtin_amdgcn_exp2f(__fmul_rn(__fsub_rn(in.ref_score, ref_score), kLog2Of10));
    double rel = group_sum_f64<L>(counts && in.relative ? scaled : 0.0);
    const double absolute = group_sum_f64<L>(counts && !in.relative ? in.sum : 0.0);
    const float not_placed = (float)p.num_branches - (float)touched;  // :174
    double score_sum;
    if (ref_score > -280.0f) {
        if (not_placed != 0.0f)
            rel += (double)(not_placed * __builtin_amdgcn_exp2f(__fmul_rn(__fsub_rn(thr_score, ref_score), kLog2Of10)));
        const double ref_power = (ref_score == best_score) ? best_power : pow10_f64((double)ref_score);
        score_sum = ref_power * rel + absolute;
    } else {
        score_sum = (double)not_placed * pow10_f64((double)thr_score) + absolute;  // :174-183, all double
    }
    const double keep_factor = (score_sum == 0.0) ? 0.0 : p.keep_factor;  // :247-251
    const double best_ratio = (score_sum == 0.0 || best_power == 0.0) ? 0.0 : best_power / score_sum;  // :191
    const double ratio_threshold = best_ratio * keep_factor;                                           // :192
    // ---- LWR (:241-264), filter_by_ratio (:188-199): which ranks stay --------------------------
    const double lwr = (my_row && score_sum != 0.0 && my_power != 0.0) ? my_power / score_sum : 0.0;  // :255-262
    const uint32_t kept_ranks = group_or_u32<L>((my_row && lwr >= ratio_threshold) ? 1u << rank : 0u);  // :197 (rank < n_sel <= 32)
    if (my_row && ((kept_ranks >> rank) & 1u)) {
        const uint32_t out_slot = (uint32_t)__popc(kept_ranks & ((1u << rank) - 1u));
        epik_amd_placement out;
        out.branch = mine.y;
        out.score = unord_f32(mine.x);
        out.lwr = lwr;
        p.rows[read * keep + out_slot] = out;
        if (p.kmer_counts) p.kmer_counts[read * keep + out_slot] = mine.z;
    }
    if (live && slot == 0) p.n_rows[read] = (uint32_t)__popc(kept_ranks);
}

// ---------------------------------------------------------------------------------
// Partial LISTS of a k-mer-space shard (include/epik_amd.h): what accumulate leaves per (read, slice) instead of
// a dense vector -- the rows that received a k-mer, as {f32 sum, u32 row | count << 16} (32-bit counts:
// {f32 sum, u32 row, u32 count, 0}), row = branch - first branch of the slice, in any order (a row is there once).
// ---------------------------------------------------------------------------------
template <typename CountT>
struct PartialEntry {
    static constexpr bool kWide = sizeof(CountT) == 4;
    static constexpr uint32_t kBytes = kWide ? 16u : 8u;
    typedef std::conditional_t<kWide, v4u, v2u> raw_t;
    __device__ static __forceinline__ raw_t make(uint32_t score_bits, uint32_t row, uint32_t count)
    {
        if constexpr (kWide)
            return v4u{score_bits, row, count, 0u};
        else
            return v2u{score_bits, row | (count << 16)};
    }
    __device__ static __forceinline__ uint32_t row(const raw_t &e)
    {
        if constexpr (kWide)
            return e.y;
        else
            return e.y & 0xffffu;
    }
    __device__ static __forceinline__ uint32_t count(const raw_t &e)
    {
        if constexpr (kWide)
            return e.z;
        else
            return e.y >> 16;
    }
};

__device__ __forceinline__ uint32_t lanes_below(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// The rows of the wave's slice that received a k-mer go to `out` (room for `cap` entries; never more are
// written), and every row is reset for the wave's next read (place.cpp:335-342).  Returns how many rows had
// received one (wave-uniform); the caller makes sure cap covers them (the front kernel's bound: the postings
// of the slice's sublists, at most the slice's rows).  Four consecutive rows per lane and trip.
template <typename CountT>
__device__ __forceinline__ uint32_t emit_partial_list(WaveLds<CountT> lds, uint32_t rows_pad, uint32_t rows,
                                                      uint8_t *__restrict__ out, uint32_t cap, uint32_t ablate = 0)
{
    (void)ablate;
    typedef WaveLds<CountT> Lds_t;
    typedef PartialEntry<CountT> Entry;
    typedef __attribute__((address_space(3))) v4u u32x4_t;
    const uint32_t lane = (uint32_t)lane_id();
    auto *dst = reinterpret_cast<typename Entry::raw_t *>(out);
    uint32_t n = 0;
    for (uint32_t base = 0; base < rows_pad; base += 4u * (uint32_t)kWave) {
        const uint32_t i0 = base + 4u * lane;
        const bool mine = i0 < rows_pad;  // rows_pad is a multiple of 16: the lane's four rows are inside or outside together
        float raw[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        uint32_t words[Lds_t::kCountWords];
#pragma unroll
        for (int q = 0; q < Lds_t::kCountWords; ++q) words[q] = 0u;
        if (mine) lds.load4_packed(i0, raw, words);
        // (a shard's lists reach a small part of the slice: a trip of 256 rows none of which has a count -- most of
        // them with many shards -- is done with this test; its rows are zero already)
        {
            uint32_t any = 0;
#pragma unroll
            for (int q = 0; q < Lds_t::kCountWords; ++q) any |= words[q];  // (counts AND the "seen" flags of the ambiguous sweep)
            if (__ballot(any != 0) == 0) {
                // the dummy row (the last one) may hold what out-of-range lanes added: it is reset below like any other
                if (base + 4u * (uint32_t)kWave < rows_pad) continue;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            uint32_t c;
            if constexpr (sizeof(CountT) == 1)
                c = (words[0] >> (8 * u)) & 0xffu;
            else if constexpr (sizeof(CountT) == 2)
                c = (words[u >> 1] >> (16 * (u & 1))) & (0xffffu & ~Lds_t::kSeen);
            else
                c = words[u] & ~Lds_t::kSeen;
            const bool hit = c != 0u && i0 + (uint32_t)u < rows;  // (the dummy row and the padding lie behind `rows`)
            const uint64_t m = __ballot(hit);
            if (m) {
                const uint32_t slot = n + lanes_below(m);
#ifdef EPIK_AMD_ABLATION
                if (ablate & 128u) cap = 0;  // (timing experiments: no entry leaves)
#endif
                if (hit && slot < cap) dst[slot] = Entry::make(__float_as_uint(raw[u]), i0 + (uint32_t)u, c);
                n += (uint32_t)__popcll(m);
            }
        }
        if (mine) {
            *reinterpret_cast<u32x4_t *>(lds.score + i0) = v4u{0u, 0u, 0u, 0u};
            if constexpr (sizeof(CountT) == 1) {
                *reinterpret_cast<typename Lds_t::u32_t *>(lds.count + i0) = 0u;
            } else {
                typename Lds_t::vcw zero;
#pragma unroll
                for (int q = 0; q < Lds_t::kCountWords; ++q) zero[q] = 0u;
                *reinterpret_cast<typename Lds_t::count_words_t *>(lds.count + i0) = zero;
            }
        }
    }
    return n;
}

// The other direction (finish): the lists every shard sent for a (read, slice) item, in shard order, added into
// the wave's rows -- the same float32 order as a dense sum over the shards in rank order, 0 + x being x
// (place.cpp:349-371 split over the shards' lists).  `item` = read * slices + slice, the read numbered inside
// the finisher's batch.  A trip is up to four chunks of 64 entries of ONE list: its rows are distinct, so the
// four LDS read-add-writes go out together (one round trip per trip, not per chunk); lists of different shards
// may name the same row and stay in order.  The next trip's entries are asked for before this one's are added,
// and the walk comes in three steps so that the caller can put an item's two dependent trips to memory (index,
// then entries) under the work on the item in front of it: request() asks for the item's index entries (lane g:
// shard g's), start() for its first trip, run() does the rest.
// where the shards' lists lie, one shard per lane (loaded from the argument block once per kernel: indexed by
// the lane there, every request would begin with a trip to memory for the pointer itself)
struct ListSources {
    uint64_t entries = 0, index = 0;  // lane g: shard g's
    uint32_t n_shards = 0;
    __device__ __forceinline__ void load(const SparseSources &src)
    {
        const uint32_t lane = (uint32_t)lane_id();
        n_shards = src.n_shards;
        if (lane < n_shards) {
            entries = (uint64_t)src.entries[lane];
            index = (uint64_t)src.index[lane];
        }
    }
};
template <typename CountT>
struct ListWalk {
    typedef PartialEntry<CountT> Entry;
    typedef typename Entry::raw_t raw_t;
    static constexpr int kTrip = 4;  // chunks of 64 entries
    struct Trip {
        raw_t e[kTrip];
    };
    uint64_t item = ~0ull;             // whose lists these are
    uint32_t first = 0, count = 0;     // lane g: where shard g's list lies in its part, and how long it is
                                       // (as loaded: nothing looks at them before settle(), or the request would wait)
    bool started = false, more = false;
    Trip trip0;                        // the first trip (started && more)
    uint32_t g = 0, at = 0, cnt = 0;   // the walk: shard, next entry of its list, the list's length (scalar)
    const uint8_t *base = nullptr;

    __device__ __forceinline__ void request(const ListSources &src, uint64_t it)
    {
        const uint32_t lane = (uint32_t)lane_id();
        item = it;
        first = count = 0;
        started = false;
        if (lane < src.n_shards) {
            typedef __attribute__((address_space(1))) const v2u global_v2u;
            const v2u ix = ((global_v2u *)(uintptr_t)src.index)[it];
            first = ix.x;
            count = ix.y;
        }
    }
    // on the next trip that exists; false at the end
    __device__ __forceinline__ bool settle(const ListSources &src)
    {
        while (g < src.n_shards) {
            cnt = (uint32_t)__builtin_amdgcn_readlane(count, (int)g);
            if (cnt == kSparseOverflow) cnt = 0;  // (the caller has checked: see epik_amd.h)
            if (at < cnt) {
                base = reinterpret_cast<const uint8_t *>(readlane_u64(src.entries, (int)g)) +
                       (uint64_t)(uint32_t)__builtin_amdgcn_readlane(first, (int)g) * Entry::kBytes;
                return true;
            }
            ++g, at = 0;
        }
        return false;
    }
    __device__ __forceinline__ Trip load_trip(uint32_t dummy)
    {
        const uint32_t lane = (uint32_t)lane_id();
        Trip t;
        // (a pointer that came through v_readlane is generic to the compiler: say that it is global memory, or the
        // loads become flat_load)
        typedef __attribute__((address_space(1))) const raw_t global_raw_t;
        global_raw_t *list = (global_raw_t *)(uintptr_t)base;
#pragma unroll
        for (int c = 0; c < kTrip; ++c) {
            t.e[c] = Entry::make(0u, dummy, 0u);  // lanes behind the list's end: +0 on the dummy row
            const uint32_t i = at + (uint32_t)c * (uint32_t)kWave + lane;
            if (i < cnt) t.e[c] = list[i];
        }
        at += (uint32_t)kTrip * (uint32_t)kWave;
        return t;
    }
    __device__ __forceinline__ void start(const ListSources &src, uint32_t dummy)
    {
        g = 0, at = 0;
        more = settle(src);
        if (more) trip0 = load_trip(dummy);
        started = true;
    }
    __device__ static __forceinline__ void add_trip(WaveLds<CountT> lds, const Trip &t)
    {
        float old_s[kTrip];
        uint32_t old_c[kTrip], row[kTrip];
#pragma unroll
        for (int c = 0; c < kTrip; ++c) {
            row[c] = Entry::row(t.e[c]);
            old_s[c] = lds.score[row[c]];
            old_c[c] = (uint32_t)lds.count[row[c]];
        }
#pragma unroll
        for (int c = 0; c < kTrip; ++c) {
            lds.score[row[c]] = __fadd_rn(old_s[c], __uint_as_float(t.e[c].x));
            lds.count[row[c]] = (CountT)(old_c[c] + Entry::count(t.e[c]));
        }
    }
    __device__ __forceinline__ void run(const ListSources &src, WaveLds<CountT> lds, uint32_t dummy)
    {
        if (!started) start(src, dummy);
        if (!more) return;
        Trip cur = trip0;
        for (;;) {
            const bool again = settle(src);
            Trip next;
            if (again) next = load_trip(dummy);
            add_trip(lds, cur);
            if (!again) break;
            cur = next;
        }
    }
};

}  // namespace
}  // namespace epik_amd
#endif
