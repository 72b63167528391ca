// place_device.hpp -- device-side building blocks shared by the placement kernels
// (place_kernel.hip: one wavefront per read; team_kernel.hip: one workgroup per read, the
// branch range split over its waves): cross-lane primitives, the database layouts in HBM,
// the k-mer encoder, the wave-private LDS vectors, the posting-chunk ring, the ambiguous
// k-mer sweep and the epilogue.  Everything lives in an anonymous namespace: each
// translation unit gets its own copy.
#ifndef EPIK_AMD_PLACE_DEVICE_HPP
#define EPIK_AMD_PLACE_DEVICE_HPP
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include <type_traits>

#include "place_kernel.h"

namespace epik_amd {

namespace {

constexpr int kWave = 64;
constexpr int kTilesPerPass = EPIK_AMD_TILES_PER_PASS;  // 64-character tiles encoded per pass
constexpr int kRing = EPIK_AMD_RING;  // posting-chunk loads kept in flight per wave
static_assert((kRing & (kRing - 1)) == 0 && kRing >= 4 && kRing <= 32, "kRing: power of two, one descriptor lane per stage");
constexpr uint32_t kChunkCap = (uint32_t)kTilesPerPass * 64u;  // chunk descriptors per round (LDS)

typedef unsigned int v2u __attribute__((ext_vector_type(2)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

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

// Cross-lane moves inside a row of 16 lanes (DPP, no LDS round trip):
// 0xB1 = quad_perm[1,0,3,2], 0x4E = quad_perm[2,3,0,1], 0x141 = row_half_mirror, 0x140 = row_mirror.
// Applying them in this order leaves every lane of a row with the row's reduction.
template <int kCtrl>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, kCtrl, 0xf, 0xf, false);
}
template <int kCtrl>
__device__ __forceinline__ uint64_t dpp_u64(uint64_t v)
{
    return ((uint64_t)dpp_u32<kCtrl>((uint32_t)(v >> 32)) << 32) | dpp_u32<kCtrl>((uint32_t)v);
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int l)
{
    // the builtin returns int: cast before widening, or the low half sign-extends
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((uint32_t)(v >> 32), l);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((uint32_t)v, l);
    return ((uint64_t)hi << 32) | lo;
}

// Wave reductions; the result is wave-uniform (combined from the four rows' lane 0/16/32/48).
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    // Six v_max with the cross-lane move as their DPP operand (the builtin route costs a copy, a DPP move and a
    // max per step, and four v_readlane at the end): quads, rows of 16, then row_bcast:15 / :31 carry the row
    // maxima forward so that lane 63 ends with the wave's.  s_nop 1: a VGPR written by the VALU is readable
    // as a DPP operand two wait states later.
    asm volatile("s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 0"
                 : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    v += dpp_u32<0xB1>(v);
    v += dpp_u32<0x4E>(v);
    v += dpp_u32<0x141>(v);
    v += dpp_u32<0x140>(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) +
           __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ uint64_t wave_or_u64(uint64_t v)
{
    v |= dpp_u64<0xB1>(v);
    v |= dpp_u64<0x4E>(v);
    v |= dpp_u64<0x141>(v);
    v |= dpp_u64<0x140>(v);
    return readlane_u64(v, 0) | readlane_u64(v, 16) | readlane_u64(v, 32) | readlane_u64(v, 48);
}
__device__ __forceinline__ double wave_sum_f64(double v)
{
    auto mv = [](double x, auto tag) {
        return __longlong_as_double((long long)dpp_u64<decltype(tag)::value>((uint64_t)__double_as_longlong(x)));
    };
    v += mv(v, std::integral_constant<int, 0xB1>{});
    v += mv(v, std::integral_constant<int, 0x4E>{});
    v += mv(v, std::integral_constant<int, 0x141>{});
    v += mv(v, std::integral_constant<int, 0x140>{});
    auto rl = [&](int l) { return __longlong_as_double((long long)readlane_u64((uint64_t)__double_as_longlong(v), l)); };
    return (rl(0) + rl(16)) + (rl(32) + rl(48));
}

// Inclusive prefix sum over the 64 lanes (row_shr DPP inside rows of 16, then the row totals).
template <int kCtrl>
__device__ __forceinline__ uint32_t dpp_zero_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, kCtrl, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
    v += dpp_zero_u32<0x111>(v);  // row_shr:1
    v += dpp_zero_u32<0x112>(v);  // row_shr:2
    v += dpp_zero_u32<0x114>(v);  // row_shr:4
    v += dpp_zero_u32<0x118>(v);  // row_shr:8
    const uint32_t r0 = __builtin_amdgcn_readlane(v, 15);
    const uint32_t r1 = __builtin_amdgcn_readlane(v, 31);
    const uint32_t r2 = __builtin_amdgcn_readlane(v, 47);
    const uint32_t row = (uint32_t)__lane_id() >> 4;
    return v + (row > 0 ? r0 : 0u) + (row > 1 ? r1 : 0u) + (row > 2 ? r2 : 0u);
}

// inverse of ord_f32
__device__ __forceinline__ float unord_f32(uint32_t o)
{
    return __uint_as_float(o ^ ((o >> 31) ? 0x80000000u : 0xffffffffu));
}

// 10^x in double.  The reference calls glibc pow(10.0, x) (place.cpp:46,181,254);
// the device routine differs from it by ulps, which is far inside the 1e-5 LWR bar.
__device__ __forceinline__ double pow10_f64(double x) { return exp10(x); }

// Everything a wave knows about the 64-character tile it is encoding.
struct Tile {
    uint32_t key;       // k-mer code of the window starting at this lane (ambiguous position = state 0)
    uint32_t prefix;    // code of the window's first k-1 letters ...
    uint32_t last;      // ... and the state of its last letter: key = prefix * sigma + last
    uint32_t first;     // state of its first letter (this lane's character)
    uint32_t cls;       // char_class of this lane's character
    uint64_t inv_mask;  // wave-uniform: lanes whose character is invalid
    uint64_t amb_mask;  // wave-uniform: lanes whose character is ambiguous
    bool in_range;      // this lane starts a window of the read (p < n_kmers, lane < tile stride)
};

// ---------------------------------------------------------------------------------
// Database layouts in HBM.  Both answer phylo_kmer_db::search (place.cpp:300,311): a
// k-mer code -> (byte offset of its posting list in p.postings, list length).
//
// A posting is i2l::pkdb_value with the branch id replaced by the LDS row that accumulates
// it, stored as cell = n_pad - 1 - branch.  The wave streams a list in chunks of <= 64
// postings through RANGE-CHECKED buffer loads: the chunk's buffer descriptor holds its
// byte length, lane l reads posting l, and a lane past the end reads 0 -- cell 0 = row
// n_pad - 1, a dummy row no branch owns.  So a stage needs no address clamp, no exec mask
// and no branch, and a padding chunk is a descriptor of zero bytes that touches no memory.
// (tools/probe_buffer.hip: the range check includes soffset; 6-byte {score, cell} structs
// read as 2-byte-aligned dwords work too but halve the load throughput.)
//
// A chunk descriptor {address (48 bits) | count << 48} is prepared by every lane at once,
// once per trip of the ring (vector work), as the kFields words a stage pulls out of the
// lanes with v_readlane (the packed layout without runs: the descriptor's two words as they
// are, the count split off by two scalar instructions per chunk).  `issue` puts one chunk's loads in flight from inline asm (hipcc
// must not count them, see the ring below); `load_posting` is the plain, compiler-counted
// access of the cold ambiguous path and returns {LDS row, score bits}.
// ---------------------------------------------------------------------------------
constexpr int kRawBufferFormat = 0x00020000;  // 4th descriptor word: untyped 32-bit raw buffer

// Compact CSR: offsets[code .. code+1] delimit the list in units of one posting; the lists
// lie back to back as 8-byte {f32 score, u32 cell} (4 or 8 bytes per code + 8 per posting;
// chosen when the table of the layout below would not fit).
template <typename OffT>
struct CompactLayout {
    static constexpr int kLoads = 1;  // vector-memory instructions per chunk
    static constexpr int kWaitLoads = 1;  // ... that the ring's waits may count on per chunk
    static constexpr uint32_t kChunkBytes = 64u * 8u;
    static constexpr int kFields = 3;  // base lo, base hi, bytes
    // what lookup() returns as `len` is the list's length and nothing else
    __device__ static __forceinline__ uint32_t length(uint32_t w) { return w; }
    // the descriptor of chunk c (`cnt` postings) of the list at byte offset `addr`
    __device__ static __forceinline__ uint64_t descriptor(const PlaceParams &p, uint64_t addr, uint32_t /*w*/, uint32_t c, uint64_t cnt)
    {
        return (uint64_t)(p.postings + addr + (uint64_t)c * kChunkBytes) | (cnt << 48);
    }
    __device__ static __forceinline__ uint64_t null_descriptor(const PlaceParams &p) { return (uint64_t)p.postings; }
    __device__ static __forceinline__ void lookup(const PlaceParams &p, uint32_t key, uint32_t /*position*/,
                                                  uint64_t &addr, uint32_t &len)
    {
        const OffT *__restrict__ offsets = static_cast<const OffT *>(p.table);
        const OffT b = offsets[key];
        const OffT e = offsets[(uint64_t)key + 1];
        addr = (uint64_t)b * 8u;
        len = (uint32_t)(e - b);
    }
    __device__ static __forceinline__ void lookup_window(const PlaceParams &p, const Tile &t, uint32_t position,
                                                         bool wanted, uint64_t &addr, uint32_t &len)
    {
        if (wanted) lookup(p, t.key, position, addr, len);
    }
    __device__ static __forceinline__ void prepare(const PlaceParams &, uint64_t d, uint32_t (&f)[kFields])
    {
        const uint32_t hi = (uint32_t)(d >> 32);
        f[0] = (uint32_t)d;
        f[1] = hi & 0xffffu;
        f[2] = (hi >> 16) * 8u;
    }
    // kSettled: the caller has put five instructions or more between the v_readlane that made `f` and this
    // statement (stream_round's stages; lint_ring_asm.py checks the distance in the ISA)
    template <bool kSettled = false>
    __device__ static __forceinline__ void issue(const uint32_t (&f)[kFields], uint32_t lane, uint32_t &cell,
                                                 uint32_t &score)
    {
        const v4i srd = {(int)f[0], (int)f[1], (int)f[2], kRawBufferFormat};
        v2u out;
        // s_nop 4: the descriptor comes out of v_readlane (VALU-written SGPR -> VMEM needs 5 wait states)
        if constexpr (kSettled)
            asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=&v"(out) : "v"(lane * 8u), "s"(srd) : "memory");
        else
            asm volatile("s_nop 4\n\tbuffer_load_dwordx2 %0, %1, %2, 0 offen"
                         : "=&v"(out)
                         : "v"(lane * 8u), "s"(srd)
                         : "memory");
        score = out.x;
        cell = out.y;
    }
    __device__ static __forceinline__ uint2 load_posting(const PlaceParams &p, uint32_t rows_pad, uint64_t addr,
                                                         uint32_t len, uint32_t j)
    {
        (void)len;
        const uint2 e = *reinterpret_cast<const uint2 *>(p.postings + addr + (uint64_t)j * 8u);
        return make_uint2(rows_pad - 1u - e.y, e.x);
    }
};

// Packed (default): a direct-index table of 8-byte entries {u32 len, u32 line} per k-mer
// code and every list on whole 128-byte lines of its own.  A list is stored chunk by chunk
// (<= 64 postings): f32 score[cnt] then u16 cell[cnt] -- 6 bytes per posting, a full chunk is
// exactly three lines, no line is shared between lists.
//
// A random 8-byte lookup costs a whole 128-byte line of fabric traffic (tools/probe_sector.hip),
// a third of what a read fetches.  kPaired (4-letter alphabets) halves the number of lines:
// the table is keyed by the (k-1)-mer X that two CONSECUTIVE k-mers of a read share -- a.X and
// X.b -- and block X holds the entries of all eight k-mers that have X as their suffix (slots
// 0..3, by first letter a) or as their prefix (slots 4..7, by last letter b).  Every k-mer is
// therefore stored twice; the k-mer at an even position of the read is looked up as a.X in
// the block of its suffix, the next one as X.b in the block of its prefix -- the same block,
// the same line, one fetch for the two lanes.  16 bytes of table per code instead of 8.
//
// kFiltered (other alphabets, sparse databases -- a protein database holds a small fraction of
// the 20^k codes): the same pairing applied to a presence filter.  filter[X] is one 64-bit
// word per (k-1)-mer X: bit a says whether a.X has a list, bit sigma + b whether X.b has one.
// Two consecutive k-mers read the same word; only the k-mers that are present go on to the
// table (keyed by code, as in the plain layout).  8 bytes of filter per (k-1)-mer.
//
// kRuns (the layouts of the one-wavefront kernels): a list whose branches are ONE ascending run b, b+1, b+2, ...
// -- a clade, what most lists of a phylo-k-mer database are (SURVEY.md 8d models all of them so) -- is stored
// without its cells: f32 score[cnt] per chunk, 4 bytes per posting instead of 6, and the cells come out of the
// table entry, whose first word is then len | first cell << 16 (0: the cells are stored, as above; cell 0 is the
// dummy row, never a posting's).  A list of 60 postings takes two 128-byte lines instead of three: a third fewer
// lines requested per read, which is what bounds these kernels.
enum : int { kPlainTable = 0, kPairedTable = 1, kFilteredTable = 2 };
template <int kTable, bool kRuns = false>
struct PackedLayout {
    static constexpr int kLoads = 2;
    // kRuns: a run chunk is ONE load (its scores), a chunk with explicit cells two -- the cells FIRST, so that the
    // scores' arrival says both are there (loads return in order).  The ring waits for "at most N younger loads in
    // flight": with a load per chunk counted it is exact for runs and waits longer than needed behind explicit
    // chunks -- never too short.  (The builder run-codes databases whose postings are nearly all in runs.)
    static constexpr int kWaitLoads = kRuns ? 1 : 2;
    static constexpr uint32_t kChunkBytes = 64u * 6u;
    // kRuns: base lo, base hi, postings in the chunk | the chunk's first cell << 7; else the descriptor's two words
    static constexpr int kFields = kRuns ? 3 : 2;
    __device__ static __forceinline__ uint32_t length(uint32_t w) { return kRuns ? (w & 0xffffu) : w; }
    // kRuns: byte offset / 2 from the start of the posting region (37 bits: 256 GiB) | cnt << 37 | first cell of the
    // chunk << 44 (0: explicit cells); else the chunk's address | cnt << 48
    __device__ static __forceinline__ uint64_t descriptor(const PlaceParams &p, uint64_t addr, uint32_t w, uint32_t c, uint64_t cnt)
    {
        if constexpr (kRuns) {
            const uint32_t first_cell = w >> 16;
            const uint64_t at = addr + (uint64_t)c * (first_cell ? 256u : kChunkBytes);
            const uint64_t cell = first_cell ? (uint64_t)(first_cell - (c << 6)) : 0ull;  // the run goes down the cells
            return (at >> 1) | (cnt << 37) | (cell << 44);
        } else {
            return (uint64_t)(p.postings + addr + (uint64_t)c * kChunkBytes) | (cnt << 48);
        }
    }
    __device__ static __forceinline__ uint64_t null_descriptor(const PlaceParams &p) { return kRuns ? 0ull : (uint64_t)p.postings; }
    // by code alone (the cold paths; the filter only saves traffic, the table is complete)
    __device__ static __forceinline__ void lookup(const PlaceParams &p, uint32_t key, uint32_t position,
                                                  uint64_t &addr, uint32_t &len)
    {
        uint64_t entry = key;
        if (kTable == kPairedTable) {
            const uint32_t shift = 2u * p.kmer_size - 2u;  // X = k-1 letters of 2 bits
            const bool as_prefix = (position & 1u) != 0;
            const uint32_t block = as_prefix ? key >> 2 : key & ((1u << shift) - 1u);
            const uint32_t slot = as_prefix ? 4u + (key & 3u) : key >> shift;
            entry = (uint64_t)block * 8u + slot;
        }
        const uint2 h = static_cast<const uint2 *>(p.table)[entry];
        len = h.x;
        addr = (uint64_t)h.y * 128u;
    }
    // the hot path: the window of lane `position` of the read, `wanted` = it is an exact k-mer
    __device__ static __forceinline__ void lookup_window(const PlaceParams &p, const Tile &t, uint32_t position,
                                                         bool wanted, uint64_t &addr, uint32_t &len)
    {
        if (kTable == kFilteredTable) {
            const bool as_prefix = (position & 1u) != 0;
            const uint32_t shared = as_prefix ? t.prefix : t.key - t.first * p.sigma_pow_km1;  // X
            const uint32_t bit = as_prefix ? p.alphabet_size + t.last : t.first;
            uint64_t word = 0ull;
            if (p.filter_rec_bytes == 8) {  // (wave-uniform)
                if (wanted) word = p.filter[shared];
            } else if (wanted) {
                // 40-bit records, 5 bytes apart: the two dwords that hold one, shifted down to its first byte
                const uint64_t at = (uint64_t)shared * 5u;
                const uint32_t *dwords = reinterpret_cast<const uint32_t *>(p.filter) + (at >> 2);
                word = (((uint64_t)dwords[1] << 32) | dwords[0]) >> (8u * ((uint32_t)at & 3u));
            }
            wanted = ((word >> bit) & 1ull) != 0;
        }
        if (wanted) lookup(p, t.key, position, addr, len);
    }
    __device__ static __forceinline__ void prepare(const PlaceParams &p, uint64_t d, uint32_t (&f)[kFields])
    {
        if constexpr (kRuns) {
            const uint64_t a = (uint64_t)p.postings + ((d & ((1ull << 37) - 1ull)) << 1);
            f[0] = (uint32_t)a;
            f[1] = (uint32_t)(a >> 32) & 0xffffu;
            f[2] = (uint32_t)(d >> 37);  // cnt (7 bits) | first cell << 7
        } else {
            // (the two words as they are: a stage takes them out of the lanes with two v_readlane and splits the
            // count off with scalar instructions -- a vector instruction less per chunk than three prepared fields)
            (void)p;
            f[0] = (uint32_t)d;
            f[1] = (uint32_t)(d >> 32);
        }
    }
    // One descriptor over the whole chunk; the cell load adds the scalar offset 4*cnt, which
    // takes part in the range check: lane l < cnt reads score l and cell l, every other lane
    // reads cell 0 (and, up to lane 1.5*cnt, a score made of cell bytes that lands on the dummy row).
    // (kSettled: see CompactLayout::issue)
    template <bool kSettled = false>
    __device__ static __forceinline__ void issue(const uint32_t (&f)[kFields], uint32_t lane, uint32_t &cell,
                                                 uint32_t &score)
    {
        // f[2] sits in a scalar register (v_readlane): the products are scalar instructions
        if constexpr (kRuns) {
            // A run (first cell != 0): the cells are first_cell, first_cell - 1, ... down the lanes -- nothing to
            // fetch for them; the lanes behind the chunk's end get cell 0, the dummy row, as a load past the end
            // would give them.
            // ONE asm statement with the branch inside: two statements in the arms of an `if` would let the
            // compiler give the slot different registers in each arm and copy them where the arms meet.
            const uint32_t cnt = f[2] & 127u, first_cell = f[2] >> 7;
            const v4i srd = {(int)f[0], (int)f[1], (int)(cnt * (first_cell ? 4u : 6u)), kRawBufferFormat};
            asm volatile(".if %9 == 0\n\ts_nop 4\n\t.endif\n\t"
                         "s_cmp_eq_u32 %6, 0\n\t"
                         "s_cbranch_scc1 .Lexplicit%=\n\t"
                         "buffer_load_dword %0, %2, %4, 0 offen\n\t"
                         "v_sub_u32 %1, %6, %7\n\t"
                         "v_cmp_gt_u32 vcc, %8, %7\n\t"
                         "v_cndmask_b32 %1, 0, %1, vcc\n\t"
                         "s_branch .Lissued%=\n"
                         ".Lexplicit%=:\n\t"
                         "buffer_load_ushort %1, %3, %4, %5 offen\n\t"
                         "buffer_load_dword %0, %2, %4, 0 offen\n"
                         ".Lissued%=:"
                         : "=&v"(score), "=&v"(cell)
                         : "v"(lane * 4u), "v"(lane * 2u), "s"(srd), "s"(cnt * 4u), "s"(first_cell), "v"(lane), "s"(cnt), "n"(kSettled ? 1 : 0)
                         : "memory", "vcc", "scc");
        } else {
            const uint32_t cnt = f[1] >> 16;
            const v4i srd = {(int)f[0], (int)(f[1] & 0xffffu), (int)(cnt * 6u), kRawBufferFormat};
            if constexpr (kSettled)
                asm volatile("buffer_load_dword %0, %2, %4, 0 offen\n\tbuffer_load_ushort %1, %3, %4, %5 offen"
                             : "=&v"(score), "=&v"(cell)
                             : "v"(lane * 4u), "v"(lane * 2u), "s"(srd), "s"(cnt * 4u)
                             : "memory");
            else
                asm volatile("s_nop 4\n\tbuffer_load_dword %0, %2, %4, 0 offen\n\tbuffer_load_ushort %1, %3, %4, %5 offen"
                             : "=&v"(score), "=&v"(cell)
                             : "v"(lane * 4u), "v"(lane * 2u), "s"(srd), "s"(cnt * 4u)
                             : "memory");
        }
    }
    __device__ static __forceinline__ uint2 load_posting(const PlaceParams &p, uint32_t rows_pad, uint64_t addr,
                                                         uint32_t w, uint32_t j)
    {
        const uint32_t len = length(w);
        if (kRuns && (w >> 16) != 0u)  // a run: scores back to back, the cells count down from the first
            return make_uint2(rows_pad - 1u - ((w >> 16) - j), *reinterpret_cast<const uint32_t *>(p.postings + addr + 4ull * j));
        const uint32_t chunk = j >> 6, r = j & 63u;
        const uint32_t rest = len - (chunk << 6);
        const uint32_t cnt = rest < 64u ? rest : 64u;
        const uint8_t *base = p.postings + addr + (uint64_t)chunk * kChunkBytes;
        const uint32_t cell = *reinterpret_cast<const uint16_t *>(base + 4u * cnt + 2u * r);
        return make_uint2(rows_pad - 1u - cell, *reinterpret_cast<const uint32_t *>(base + 4u * r));
    }
};

// absolute address of chunk c (64 postings each) of the list at byte offset `addr`
template <typename Layout>
__device__ __forceinline__ uint64_t chunk_address(const PlaceParams &p, uint64_t addr, uint32_t c)
{
    return (uint64_t)(p.postings + addr + (uint64_t)c * Layout::kChunkBytes);
}
// the padding chunk: zero bytes at a valid address
__device__ __forceinline__ uint64_t null_chunk(const PlaceParams &p) { return (uint64_t)p.postings; }

// i2l::to_kmers<one_ambiguity_policy> for one tile (place.cpp:294) in three steps, so that a
// pass can issue the loads of all its tiles together: the character of this lane (branch-free:
// positions past the end re-read the last character and are masked later), its class through
// the 256-entry table, and the window code gathered from the next k-1 lanes.
__device__ __forceinline__ uint32_t tile_char(const uint8_t *__restrict__ seq, uint64_t len, uint64_t tile_pos)
{
    const uint64_t pos = tile_pos + (uint64_t)lane_id();
    return seq[pos < len ? pos : len - 1];  // len >= k >= 1
}
__device__ __forceinline__ uint32_t tile_class(uint32_t ch, uint64_t len, uint64_t tile_pos,
                                               const uint32_t *__restrict__ char_class)
{
    const uint32_t cls = char_class[ch];
    return tile_pos + (uint64_t)lane_id() < len ? cls : 0u;  // past the end: no character
}
__device__ __forceinline__ Tile tile_from_class(uint32_t cls, uint64_t len, uint64_t tile_pos, uint64_t n_kmers,
                                                uint32_t k, uint32_t sigma, uint32_t stride)
{
    Tile t;
    const int lane = lane_id();
    const uint64_t pos = tile_pos + (uint64_t)lane;
    const bool multi = (cls & (cls - 1)) != 0;
    const uint32_t state = (cls && !multi) ? (uint32_t)(__ffs((int)cls) - 1) : 0u;
    t.cls = cls;
    t.inv_mask = __ballot(cls == 0 && pos < len);  // characters past the end belong to no window
    t.amb_mask = __ballot(multi);
    // Window code of the k characters from this lane on: the states of the next lanes come down
    // one lane per step with a whole-wave DPP shift (wave_shl:1, lane i <- lane i+1, 0 behind lane
    // 63) -- a few cycles per step, where an LDS shuffle per step costs a round trip each.
    uint32_t prefix = state, next = state;  // the first k-1 letters, then the last one separately
    const uint32_t steps = __builtin_amdgcn_readfirstlane(k) - 1u;
    if (sigma == 4u) {  // wave-uniform
        for (uint32_t j = 1; j < steps; ++j) {
            next = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)next, 0x130, 0xf, 0xf, true);
            prefix = (prefix << 2) | next;
        }
    } else {
        for (uint32_t j = 1; j < steps; ++j) {
            next = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)next, 0x130, 0xf, 0xf, true);
            prefix = prefix * sigma + next;
        }
    }
    if (steps == 0) {  // k == 1: no prefix
        t.prefix = 0;
        t.last = state;
        t.key = state;
    } else {
        t.prefix = prefix;
        t.last = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)next, 0x130, 0xf, 0xf, true);
        t.key = prefix * sigma + t.last;
    }
    t.first = state;
    t.in_range = ((uint32_t)lane < stride) && (pos < n_kmers);
    return t;
}
__device__ __forceinline__ Tile encode_tile(const uint8_t *__restrict__ seq, uint64_t len,
                                            uint64_t tile_pos, uint64_t n_kmers, uint32_t k,
                                            uint32_t sigma, uint32_t stride,
                                            const uint32_t *__restrict__ char_class)
{
    const uint32_t cls = tile_class(tile_char(seq, len, tile_pos), len, tile_pos, char_class);
    return tile_from_class(cls, len, tile_pos, n_kmers, k, sigma, stride);
}

// Wave-private LDS: score[b] = _scores[thread][b] (float32), count[b] = _counts[thread][b]
// (place.h:126-131) in two arrays -- counts are 16-bit by default (reads of up to 32767
// k-mers; 6 bytes per branch let 20 waves share a CU's 160 KiB), 32-bit in the "wide"
// kernels the host selects for longer reads.  The top bit of a count is the
// "already scored by an ambiguous key" flag.  `desc` holds the chunk descriptors of the
// current round and is reused by the epilogue for its top-k candidates.
template <typename CountT>
struct WaveLds {
    // 16- and 32-bit counts keep the ambiguous path's "seen" flag in their top bit; 8-bit counts
    // use all their bits (reads of up to 255 k-mers) and the flags live in a bitmap (place_ambiguous)
    static constexpr uint32_t kSeen = sizeof(CountT) == 1 ? 0u : 1u << (8 * sizeof(CountT) - 1);
    static constexpr uint64_t kMaxKmers = sizeof(CountT) == 1 ? 255u : (1ull << (8 * sizeof(CountT) - 1)) - 1u;
    // LDS-address-space pointers (32 bits): the out-of-line functions below get them as arguments
    // and must still compile their accesses to ds_read / ds_write -- through generic pointers they
    // become flat_load / flat_store, which take the slower path through the address check.
    typedef __attribute__((address_space(3))) float f32_t;
    typedef __attribute__((address_space(3))) CountT count_t;
    typedef __attribute__((address_space(3))) uint64_t u64_t;
    typedef __attribute__((address_space(3))) uint32_t u32_t;
    typedef __attribute__((address_space(3))) v2u u32x2_t;  // (a native vector: HIP's uint2 is a class and cannot live in LDS-typed memory)
    f32_t *score;    // [n_pad]
    count_t *count;  // [n_pad]
    u64_t *desc;     // [kTilesPerPass * 64 + kRing]
    // carves the three arrays out of `base` (16-byte aligned, generic pointer into the kernel's LDS)
    __device__ __forceinline__ void carve(unsigned char *base, uint32_t n_pad)
    {
        score = (f32_t *)reinterpret_cast<float *>(base);
        count = (count_t *)reinterpret_cast<CountT *>(base + (size_t)n_pad * 4);
        desc = (u64_t *)reinterpret_cast<uint64_t *>(base + (size_t)n_pad * (4 + sizeof(CountT)));
    }
    __device__ __forceinline__ uint2 load(uint32_t i) const
    {
        return make_uint2(__float_as_uint(score[i]), (uint32_t)count[i]);
    }
    __device__ __forceinline__ void store(uint32_t i, uint32_t score_bits, uint32_t c) const
    {
        score[i] = __uint_as_float(score_bits);
        count[i] = (CountT)c;
    }
    // Four consecutive rows from i0 (a multiple of 4) at once: one 16-byte LDS access for the scores,
    // one of 4 / 8 / 16 bytes for the counts.
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef CountT v4c __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4f f32x4_t;
    typedef __attribute__((address_space(3))) v4c countx4_t;
    __device__ __forceinline__ void load4(uint32_t i0, float (&s)[4], uint32_t (&c)[4]) const
    {
        const v4f sv = *reinterpret_cast<f32x4_t *>(score + i0);
        const v4c cv = *reinterpret_cast<countx4_t *>(count + i0);
        s[0] = sv.x, s[1] = sv.y, s[2] = sv.z, s[3] = sv.w;
        c[0] = (uint32_t)cv.x, c[1] = (uint32_t)cv.y, c[2] = (uint32_t)cv.z, c[3] = (uint32_t)cv.w;
    }
    // ... the four counts as they lie in LDS: one word (8-bit), two (16-bit) or four, rows in ascending order
    static constexpr int kCountWords = (int)sizeof(CountT);
    typedef uint32_t vcw __attribute__((ext_vector_type(sizeof(CountT) == 1 ? 1 : sizeof(CountT) == 2 ? 2 : 4)));
    typedef __attribute__((address_space(3))) vcw count_words_t;
    __device__ __forceinline__ void load4_packed(uint32_t i0, float (&s)[4], uint32_t (&w)[kCountWords]) const
    {
        const v4f sv = *reinterpret_cast<f32x4_t *>(score + i0);
        s[0] = sv.x, s[1] = sv.y, s[2] = sv.z, s[3] = sv.w;
        if constexpr (sizeof(CountT) == 1) {
            w[0] = *reinterpret_cast<u32_t *>(count + i0);
        } else {
            const vcw cv = *reinterpret_cast<count_words_t *>(count + i0);
#pragma unroll
            for (int q = 0; q < kCountWords; ++q) w[q] = cv[q];
        }
    }
    __device__ __forceinline__ void load_scores4(uint32_t i0, float (&s)[4]) const
    {
        const v4f sv = *reinterpret_cast<f32x4_t *>(score + i0);
        s[0] = sv.x, s[1] = sv.y, s[2] = sv.z, s[3] = sv.w;
    }
    __device__ __forceinline__ void store_scores4(uint32_t i0, const float (&s)[4]) const
    {
        *reinterpret_cast<f32x4_t *>(score + i0) = v4f{s[0], s[1], s[2], s[3]};
    }
    // place.cpp:335-342: all rows back to zero for the wave's next read, 16 bytes per lane and store
    __device__ __forceinline__ void clear(uint32_t n_pad) const
    {
        typedef __attribute__((address_space(3))) v4u u32x4_t;
        // (a scalar loop over whole trips of 64 stores, unrolled four times, was slower: N = 9 999, +0.8 %)
        const uint32_t lane = __lane_id();
        auto *s16 = reinterpret_cast<u32x4_t *>(score);
        for (uint32_t i = lane; i < n_pad / 4u; i += 64u) s16[i] = v4u{0u, 0u, 0u, 0u};
        auto *c16 = reinterpret_cast<u32x4_t *>(count);  // n_pad * sizeof(CountT) is a multiple of 64 bytes
        for (uint32_t i = lane; i < n_pad * (uint32_t)sizeof(CountT) / 16u; i += 64u) c16[i] = v4u{0u, 0u, 0u, 0u};
    }
};

// ---------------------------------------------------------------------------------
// The stream: n_padded (a multiple of kDepth) chunk descriptors, in read order, go through a
// ring of kDepth chunk loads in flight and are added into the wave's LDS vectors
// (place.cpp:349-371).  score_top / count_top: LDS byte addresses of the dummy row (cell 0)
// in the two vectors; chunks[] has one trip of spare entries behind n_padded.
//
// The loads are issued from inline asm (Layout::issue): hipcc must not count them, or it would
// drain the ring (vmcnt(0)) once per trip of the loop.  A stage waits for its slot with a
// counted s_waitcnt -- loads retire in issue order --, consumes the slot's two registers inside
// asm statements only, and refills the slot with the next chunk.  (Letting hipcc read a slot
// register itself, even behind a "+v" wait, is not safe: it is free to copy it into another
// register AHEAD of the wait; lint_ring_asm.py checks the generated ISA for that.)
// Every lane updates the row of its posting: branches are distinct inside a list, and the lanes
// past the chunk's end all hold the dummy row, whose content nobody reads.  Two consecutive
// chunks may hit the same row, so a chunk's LDS read-add-write has to be issued before the next
// chunk's read (LDS executes a wave's operations in order): what CAN go under the round trip of
// the reads is everything of the next stage that does not touch LDS.
//
// The stages are software-pipelined by hand: the wait for the NEXT slot and its two address
// computations are issued between this stage's LDS reads and the add that needs their data, i.e.
// under the LDS round trip the wave would otherwise just sit out (round 2: +4 % on the large
// trees, +1 % on the benchmark).  While stage i waits for slot i+1, slot i has been consumed and
// not yet refilled: seven slots are in flight and slot i+1 is the oldest, so the wait is
// vmcnt(kLoads * (kDepth - 2)).
// ---------------------------------------------------------------------------------
// The postings of a round's first kDepth chunks when they were fetched ahead (team_stream_kernel: the descriptors
// of a wave's next read are there a read ahead, so its first trip to memory can lie under the epilogue of the read in
// front): plain loads into plain registers -- the compiler counts them, and nothing of the ring is in flight
// while they are.  cell[i] / score[i]: what the ring's slot i would hold for chunk i.
template <int kDepth>
struct Preloaded {
    uint32_t cell[kDepth], score[kDepth];
};
// (`d`: the chunk's descriptor, address | count << 48, the same in all lanes)
template <typename Layout>
__device__ __forceinline__ void preload_chunk(uint64_t d, uint32_t &cell, uint32_t &score)
{
    static_assert(Layout::kChunkBytes == 64u * 6u, "the packed chunk format: f32 score[cnt], u16 cell[cnt]");
    typedef __attribute__((address_space(1))) const uint32_t global_u32;
    typedef __attribute__((address_space(1))) const uint16_t global_u16;
    const uint32_t lane = (uint32_t)lane_id();
    const uint64_t addr = d & 0xffffffffffffull;
    const uint32_t cnt = (uint32_t)(d >> 48);
    cell = score = 0;  // lanes past the end: cell 0, the dummy row
    if (lane < cnt) {
        score = ((global_u32 *)(uintptr_t)addr)[lane];
        cell = ((global_u16 *)(uintptr_t)(addr + 4u * cnt))[lane];
    }
}

// n_real (<= n_padded, 0: all of them): how many of the descriptors are chunks -- the rest is padding, and the slots of
// the last trip that hold padding are not retired.
// cells_out: where the cells of the ring's slots are left when the function returns -- for a round of ONE trip
// (n_padded == kDepth) slot i still holds chunk i's cells (0 behind its end and in padding slots), drained: the
// caller lists the rows the round touched from them (team_stream.hip: emit_partial_list_cells).
struct NoCells {};
template <typename Layout, typename CountT, int kDepth = kRing, bool kPreloaded = false, typename Cells = NoCells>
__device__ __forceinline__ void stream_round(const PlaceParams &p, const typename WaveLds<CountT>::u64_t *chunks,
                                                       uint32_t n_padded, uint32_t score_top, uint32_t count_top,
                                                       const Preloaded<kDepth> *pre = nullptr, uint32_t n_real = 0,
                                                       Cells &&cells_out = NoCells{})
{
    typedef __attribute__((address_space(3))) float lds_f32;
    typedef __attribute__((address_space(3))) CountT lds_count;
    const int lane = lane_id();
    uint32_t ring_c[kDepth], ring_s[kDepth];
#pragma unroll
    for (int i = 0; i < kDepth; ++i) {
        if constexpr (kPreloaded)
            ring_c[i] = pre->cell[i], ring_s[i] = pre->score[i];  // the round's first chunks are here already
        else
            ring_c[i] = ring_s[i] = 0;  // cell 0: the dummy row
    }
    // the two LDS addresses of a slot's posting, behind the wait for its loads
    auto addresses = [](uint32_t &slot_cell, auto wait_count, uint32_t &score_addr, uint32_t &count_addr, uint32_t s_top,
                        uint32_t c_top) {
        asm volatile("s_waitcnt vmcnt(%5)\n\t"
                     "v_mad_i32_i24 %0, %2, -4, %3\n\t"
                     "v_mad_i32_i24 %1, %2, %6, %4"
                     : "=&v"(score_addr), "=&v"(count_addr)
                     : "v"(slot_cell), "s"(s_top), "s"(c_top), "n"(decltype(wait_count)::value),
                       "n"(-(int)sizeof(CountT))
                     : "memory");
    };
    // one stage: LDS reads of this slot's rows; the refill's descriptor words out of the lanes and the NEXT
    // slot's wait and addresses meanwhile; add, LDS writes; refill
    auto stage = [&](uint32_t score_addr, uint32_t count_addr, uint32_t &slot_score, auto &&meanwhile, auto &&refill) {
        lds_f32 *score_cell = (lds_f32 *)(uintptr_t)score_addr;
        lds_count *count_cell = (lds_count *)(uintptr_t)count_addr;
#ifdef EPIK_AMD_ABLATION
        const bool skip_acc = (p.ablate & 1u) != 0;  // (timing experiments: no LDS update)
        if (skip_acc) asm volatile("" ::"v"(score_addr), "v"(count_addr));
#else
        constexpr bool skip_acc = false;
#endif
        float old_score = 0.0f;
        uint32_t old_count = 0;
        if (!skip_acc) {
            old_score = *score_cell;
            old_count = (uint32_t)*count_cell;
        }
        meanwhile();
        __builtin_amdgcn_sched_barrier(0);
        float new_score;
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(new_score) : "v"(old_score), "v"(slot_score) : "memory");
        if (!skip_acc) {
            *score_cell = new_score;
            *count_cell = (CountT)(old_count + 1u);
        }
        refill();
    };
    if (n_padded == 0) return;  // (no caller does; the first trip below would fetch whatever the list held before)
    uint32_t sa, ca;  // addresses of the slot the next stage works on
    // (preloaded: the ring starts out holding chunks 0 .. kDepth-1, and the first trip of the loop fetches the next kDepth)
    constexpr uint32_t kFirst = kPreloaded ? (uint32_t)kDepth : 0u;
    uint64_t d_next = chunks[kFirst + (lane & (kDepth - 1))];  // descriptors of the first trip, lane i <-> stage i
    if constexpr (kPreloaded) {
        addresses(ring_c[0], std::integral_constant<int, 0>{}, sa, ca, score_top, count_top);
    } else {
        // The ring is empty: the first trip only puts its chunks in flight (a stage here would read, add 0 to and
        // write the dummy row: kDepth LDS round trips for nothing -- a sixth of a short list's time in the ring).
        uint32_t field[Layout::kFields];
        Layout::prepare(p, d_next, field);
        d_next = chunks[kDepth + (lane & (kDepth - 1))];
#pragma unroll
        for (int i = 0; i < kDepth; ++i) {
            uint32_t f[Layout::kFields];
#pragma unroll
            for (int q = 0; q < Layout::kFields; ++q) f[q] = __builtin_amdgcn_readlane(field[q], i);
            Layout::issue(f, (uint32_t)lane, ring_c[i], ring_s[i]);
        }
        // slot 0 is the oldest of the kDepth in flight
        addresses(ring_c[0], std::integral_constant<int, Layout::kWaitLoads *(kDepth - 1)>{}, sa, ca, score_top, count_top);
    }
    for (uint32_t c0 = (uint32_t)kDepth; c0 < n_padded; c0 += kDepth) {
        uint32_t field[Layout::kFields];
        Layout::prepare(p, d_next, field);
        d_next = chunks[c0 + kDepth + (lane & (kDepth - 1))];  // next trip (spare entries behind the end)
#pragma unroll
        for (int i = 0; i < kDepth; ++i) {
            uint32_t f[Layout::kFields];
            uint32_t sa_next, ca_next;
            stage(
                sa, ca, ring_s[i],
                [&]() {
                    // (asm statements keep their order: the descriptor words leave the lanes HERE, and the wait and
                    // the two multiply-adds below, the add of this stage, the scalar instructions that split the
                    // words and the LDS writes lie between them and the refill that reads them as a buffer resource --
                    // the five wait states a VALU-written SGPR needs before a VMEM instruction may read it, without
                    // the s_nop 4 the refill would otherwise begin with)
#pragma unroll
                    for (int q = 0; q < Layout::kFields; ++q)
                        asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(f[q]) : "v"(field[q]), "i"(i));
                    addresses(ring_c[(i + 1) % kDepth], std::integral_constant<int, Layout::kWaitLoads *(kDepth - 2)>{}, sa_next, ca_next, score_top,
                              count_top);
                },
                [&]() { Layout::template issue<true>(f, (uint32_t)lane, ring_c[i], ring_s[i]); });
            sa = sa_next, ca = ca_next;
        }
    }
    // tail: nothing more to issue; retire the ring (slot 0 has been waited for; the rest after one drain -- the first
    // stage's, which always runs: nothing is in flight when the function returns).  Slots that hold padding are left.
    const uint32_t last_real = n_real ? n_real - (n_padded - (uint32_t)kDepth) : (uint32_t)kDepth;  // chunks in the last trip: 1 .. kDepth
#pragma unroll
    for (int i = 0; i < kDepth; ++i) {
        if (i == 0 || (uint32_t)i < last_real) {  // wave-uniform
            uint32_t sa_next = 0, ca_next = 0;
            stage(
                sa, ca, ring_s[i],
                [&]() {
                    if (i + 1 < kDepth) addresses(ring_c[i + 1], std::integral_constant<int, 0>{}, sa_next, ca_next, score_top, count_top);
                },
                []() {});
            sa = sa_next, ca = ca_next;
        }
    }
    if constexpr (!std::is_same_v<std::remove_cvref_t<Cells>, NoCells>) {
        // (the tail's first stage has drained the ring: every slot's load has returned.  Through an asm statement, so
        // that the copy stands HERE: a plain read of a slot register the compiler may place ahead of the drain)
#pragma unroll
        for (int i = 0; i < kDepth; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(cells_out[i]) : "v"(ring_c[i]));
    }
}

// ---------------------------------------------------------------------------------
// The two parts of a read's placement that need many registers -- the cold ambiguous-k-mer
// sweep (double-precision pow) and the epilogue (double-precision exp10, unrolled sweeps) --
// are real functions, not inlined: inlined, they set the register allocation of the whole
// kernel (~125 VGPRs) although the streaming loop itself needs ~70.  `kp` points at the
// kernel's own argument block.
// ---------------------------------------------------------------------------------
// Which rows of the tree a wave accumulates.  WaveCtx: all of them -- one wavefront places a
// read by itself (place_kernel.hip); everything comes from the kernel arguments, the context is
// empty and costs nothing.  team_kernel.hip has a context of its own: one slice of the branch
// range, the rest of the read's workgroup owning the other slices.
struct WaveCtx {
    static constexpr bool kTeam = false;
    static constexpr uint32_t kCandCap = kChunkCap;  // top-k candidates: as many as chunk descriptors fit
    __device__ __forceinline__ bool untouched() const { return false; }  // (a team context may know that no row holds a count)
    template <typename Params>
    __device__ __forceinline__ uint32_t rows_pad(const Params &p) const { return p.n_pad; }
    template <typename Params>
    __device__ __forceinline__ uint32_t rows(const Params &p) const { return p.num_branches; }
    __device__ __forceinline__ uint32_t branch_base() const { return 0u; }
    // the placer constants the epilogue needs (a team context carries them in registers instead)
    __device__ __forceinline__ uint32_t kmer_size(const PlaceParams &p) const { return p.kmer_size; }
    __device__ __forceinline__ float log_threshold(const PlaceParams &p) const { return p.log_threshold; }
    __device__ __forceinline__ uint32_t keep_at_most(const PlaceParams &p) const { return p.keep_at_most; }
    template <typename Layout>
    __device__ __forceinline__ void lookup(const PlaceParams &p, uint32_t key, uint64_t &addr, uint32_t &len) const
    {
        Layout::lookup(p, key, 0u, addr, len);
    }
};

// amb_slot >= 0 (accumulate-only launches of a k-mer-space shard): the read's ambiguous k-mers
// are not added to the sums but recorded, per branch, as {order of the first ambiguous key of
// THIS shard that reached it, its average probability} in row amb_slot of p.amb_order / p.amb_avg;
// the shards' records are combined by minimum order before finish (place.cpp:385-388 is global).
template <typename Layout, typename CountT, typename Ctx>
__device__ __attribute__((noinline)) void place_ambiguous(const PlaceParams *__restrict__ kp, WaveLds<CountT> lds,
                                                          const uint8_t *__restrict__ seq, uint64_t len,
                                                          uint64_t n_kmers, int64_t amb_slot, Ctx ctx)
{
    const PlaceParams &p = *kp;
    const int lane = lane_id();
    const uint32_t k = p.kmer_size;
    const uint32_t sigma = p.alphabet_size;
    const uint32_t stride = kWave - (k - 1);
    const float k_f = (float)k;
    const uint32_t rows_pad = ctx.rows_pad(p);
    // 8-bit counts: one "already scored by an ambiguous key" bit per row, in the chunk-descriptor
    // area (idle between the stream and the epilogue; the host offers this kernel only when it fits)
    auto *seen_bitmap = reinterpret_cast<typename WaveLds<CountT>::u32_t *>(lds.desc);
    if (sizeof(CountT) == 1)
        for (uint32_t i = lane; i < (rows_pad + 31u) / 32u; i += kWave) seen_bitmap[i] = 0u;
    uint32_t *order_row = nullptr;
    float *avg_row = nullptr;
    if (amb_slot >= 0) {
        order_row = p.amb_order + (uint64_t)amb_slot * p.num_branches + ctx.branch_base();
        avg_row = p.amb_avg + (uint64_t)amb_slot * p.num_branches + ctx.branch_base();
        for (uint32_t i = lane; i < ctx.rows(p); i += kWave) {
            order_row[i] = 0xffffffffu;
            avg_row[i] = 0.0f;
        }
    }
    // ---- ambiguous k-mers (place.cpp:306-313, 373-415), after all exact ones ------
    {
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
                    // position of this key among the read's ambiguous keys (k-mer, then state)
                    const uint32_t order = ((uint32_t)tile_pos + (uint32_t)m) * sigma + st;
                    uint64_t b0;
                    uint32_t w;  // the list's length (with the run layouts: | its first cell << 16)
                    ctx.template lookup<Layout>(p, key, b0, w);
                    const uint32_t n = Layout::length(w);
                    for (uint32_t off = 0; off < n; off += kWave) {
                        if (off + (uint32_t)lane < n) {
                            const uint2 e = Layout::load_posting(p, rows_pad, b0, w, off + (uint32_t)lane);  // {row, score bits}
                            uint2 cv = lds.load(e.x);
                            const uint32_t c = cv.y;
                            // Only the first ambiguous key that reaches a branch scores it:
                            // later ones find counts_amb[b] != 0 and stay out of l_amb (:385-388).
                            bool seen = (c & lds.kSeen) != 0;
                            if (sizeof(CountT) == 1) {  // no flag bit in the count: a bitmap in the descriptor area
                                const uint32_t bit = 1u << (e.x & 31u);
                                seen = (__hip_atomic_fetch_or(&seen_bitmap[e.x >> 5], bit, __ATOMIC_RELAXED,
                                                              __HIP_MEMORY_SCOPE_WAVEFRONT) & bit) != 0;
                            }
                            if (!seen) {
                                // counts_amb[b] == 1, scores_amb[b] == float(pow(10, score)) (:390-391)
                                const float prob = (float)pow(10.0, (double)__uint_as_float(e.y));
                                const float avg = __fdiv_rn(
                                    __fadd_rn(prob, __fmul_rn((float)(k - 1u), thr)), k_f);  // :400-402
                                if (order_row) {  // recorded for the cross-shard combination
                                    order_row[e.x] = order;
                                    avg_row[e.x] = avg;
                                    if (sizeof(CountT) != 1) lds.count[e.x] = (CountT)(c | lds.kSeen);
                                } else {
                                    cv.y = (c | lds.kSeen) + 1u;                                   // :409
                                    cv.x = __float_as_uint(__fadd_rn(__uint_as_float(cv.x), avg)); // :410
                                    lds.store(e.x, cv.x, cv.y);
                                }
                            }
                        }
                    }
                }
            }
        }
    }
}

// What one slice of the branch range hands to the merge of a team placement (team_kernel.hip):
// its best rows, ranked, as {ord(score), branch, k-mer count, -} and its share of sum_scores.
struct TeamPartial {
    uint32_t touched;    // branches of the slice that received a k-mer
    uint32_t relative;   // 1: sum is relative to 10^ref_score (float32 terms), 0: absolute (all double)
    float ref_score;     // max(best score of the slice, threshold score)
    uint32_t reserved;
    double sum;
};

// With Ctx = WaveCtx this is the whole epilogue of a read.  With a team context (Ctx::kTeam) it
// works on one slice of the branches: correction, the slice's own top rows and its share of
// sum_scores go to the merge area in LDS (ctx.cand / ctx.partial) and the function returns after
// clearing the slice; team_merge() then does the part from "10^score of every row" on.
#ifndef EPIK_AMD_TAU_STOP
#define EPIK_AMD_TAU_STOP 8
#endif
template <typename Layout, typename CountT, typename Ctx>
__device__ __forceinline__ void place_epilogue_body(const PlaceParams *__restrict__ kp, WaveLds<CountT> lds,
                                                    uint64_t read, uint64_t n_kmers, Ctx ctx)
{
    // (__builtin_amdgcn_kernarg_segment_ptr() is null inside an out-of-line function: the block comes as kp)
    const PlaceParams &p = *kp;
    const int lane = lane_id();
#ifdef EPIK_AMD_ABLATION
    // A timeline of ONE wave (the caller gives it ten entries from ctx.trace_at_ on; EPIK_AMD_STAMPS=1): {code,
    // s_memtime} pairs into dbg[64 ..], nothing from any other wave.
#define EPI_STAMP(k)                                                                   \
    if constexpr (Ctx::kTeam) {                                                        \
        if (ctx.trace_at_ != 0xffffffffu && lane == 0) {                               \
            const uint32_t i_ = ctx.trace_at_ + ((k) == 10 ? 0u : (uint32_t)(k) + 1u); \
            if (i_ < 100000u) {                                                        \
                p.dbg[64 + 2 * (size_t)i_] = (unsigned long long)(k);                  \
                p.dbg[65 + 2 * (size_t)i_] = __builtin_amdgcn_s_memtime();             \
            }                                                                          \
        }                                                                              \
    }
    // (timing experiments, EPIK_AMD_ABLATE bits 256 / 512 / 1024 / 2048: the epilogue stops behind its correction
    // sweep / tau / scan sweep / rank -- rows cleared, no result; SQ_INSTS_VALU of such runs tells the parts apart)
#define EPI_STOP(bit)                                                              \
    if (p.ablate & (bit)) {                                                        \
        lds.clear((uint32_t)__builtin_amdgcn_readfirstlane((int)ctx.rows_pad(p))); \
        return;                                                                    \
    }
#else
#define EPI_STAMP(k)
#define EPI_STOP(bit)
#endif
    typedef WaveLds<CountT> Lds_t;
    EPI_STAMP(10)  // entered
    // The arguments of an out-of-line function arrive in vector registers, and whatever is computed from
    // them counts as divergent: every loop over the rows and every test below would be compiled with exec
    // masks.  They are the same in all lanes; readfirstlane says so (scalar loops, scalar branches).
    auto uniform = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    n_kmers = ((uint64_t)uniform((uint32_t)(n_kmers >> 32)) << 32) | uniform((uint32_t)n_kmers);
    const uint32_t N = uniform(ctx.rows(p));
    const uint32_t kmer_size = uniform(ctx.kmer_size(p));
    const float k_f = (float)kmer_size;
    const float log_thr = __uint_as_float(uniform(__float_as_uint(ctx.log_threshold(p))));
    // ---- score correction (:418-422), dense over N --------------------------------------
    // score[i] becomes the corrected score (-inf = "not an edge"); count[i] keeps the count
    // (and the ambiguous path's flag bit).
    const float nk_f = (float)n_kmers;
    const uint32_t nk_u = (uint32_t)n_kmers;  // the host rejects reads of 2^32 characters or more
    const float inv_k = __fdiv_rn(1.0f, k_f);
    // x / k in three instructions: y = RN(1/k); q = RN(x*y); r = fma(-q, k, x) (exact);
    // q' = fma(r, y, q).  Bit-identical to the IEEE quotient for every k <= 32 and every
    // finite x with |x| >= 2^-102 (tools/test_div.hip sweeps all 2^32 floats); a trip that
    // meets a smaller |x| (or a database with k > 32) is redone with v_div.
    auto div_k = [&](float x) {
        const float q = __fmul_rn(x, inv_k);
        const float r = __fmaf_rn(-q, k_f, x);
        return __fmaf_rn(r, inv_k, q);
    };
    const bool fast_div = kmer_size <= 32u;
    uint32_t touched = 0;           // wave-uniform: counted with ballots
    float lane_best_f = -INFINITY;  // this lane's best score
    // The sweeps run over the padded rows [0, n_pad), n_pad a multiple of 16: cells behind N hold
    // no count (the dummy row of the out-of-range lanes was cleared by the caller), so rows need no
    // bounds test.  Four consecutive rows per lane (one 16-byte LDS access), 256 rows per trip; the
    // arithmetic is branch-free.
    constexpr int kUnroll = 4;
    const uint32_t n_rows_pad = uniform(ctx.rows_pad(p));
    // A trip in two halves -- the LDS reads, then everything else -- so that the loop can ask for the next
    // trip's rows before it works on this one's (the wave would otherwise sit out an LDS round trip per trip).
    struct Trip {
        float raw[kUnroll];
        uint32_t words[Lds_t::kCountWords];
    };
    auto load_trip = [&](auto whole, uint32_t base) {
        constexpr bool kWhole = decltype(whole)::value;
        Trip t;
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) t.raw[u] = 0.0f;
#pragma unroll
        for (int q = 0; q < Lds_t::kCountWords; ++q) t.words[q] = 0u;
        const uint32_t i0 = base + 4u * (uint32_t)lane;
        // the last, partial trip of 256 rows (kWhole == false) leaves the lanes behind the end without rows
        if (kWhole || i0 < n_rows_pad) lds.load4_packed(i0, t.raw, t.words);
        return t;
    };
    auto correct_rows = [&](auto whole, auto fast, uint32_t base, const Trip &t) {
        constexpr bool kWhole = decltype(whole)::value;
        constexpr bool kFastDiv = decltype(fast)::value;  // k <= 32: the three-instruction division
        constexpr int kRows = kUnroll;
        const uint32_t i0 = base + 4u * (uint32_t)lane;
        const bool mine = kWhole || i0 < n_rows_pad;
        float missing[kRows], pre[kRows], s[kRows];
        bool edge[kRows];
#pragma unroll
        for (int u = 0; u < kRows; ++u) {
            if constexpr (sizeof(CountT) <= 2) {
                // the count goes from its byte / half-word straight to float32 (one conversion instruction),
                // and nk - c is exact in float32 (both below 2^16): the same value as float(nk - c) of :420
                const uint32_t w = sizeof(CountT) == 1 ? (t.words[0] >> (8 * u)) & 0xffu
                                                       : (t.words[u >> 1] >> (16 * (u & 1))) & (0xffffu & ~lds.kSeen);
                const float c_f = (float)w;
                missing[u] = nk_f - c_f;
                edge[u] = c_f != 0.0f;
            } else {
                const uint32_t c = t.words[u] & ~lds.kSeen;
                missing[u] = (float)(nk_u - c);
                edge[u] = c != 0u;
            }
        }
        float smallest = INFINITY;
#pragma unroll
        for (int u = 0; u < kRows; ++u) {
            pre[u] = __fadd_rn(t.raw[u], __fmul_rn(missing[u], log_thr));  // :420
            s[u] = kFastDiv ? div_k(pre[u]) : __fdiv_rn(pre[u], k_f);       // :421
            // (rows without a k-mer take part too: theirs is nk * log_thr, tiny only if log_thr is 0 --
            // then the exact division below runs, which is as right and only slower)
            smallest = fminf(smallest, fabsf(pre[u]));
        }
        if (kFastDiv && __builtin_expect(__ballot(smallest < 0x1p-100f) != 0, 0)) {  // wave-uniform, practically never
#pragma unroll
            for (int u = 0; u < kRows; ++u) s[u] = __fdiv_rn(pre[u], k_f);
        }
#pragma unroll
        for (int u = 0; u < kRows; ++u) {
            s[u] = edge[u] ? s[u] : -INFINITY;  // -inf = "not an edge"
            touched += (uint32_t)__popcll(__ballot(edge[u]));
            lane_best_f = fmaxf(lane_best_f, s[u]);
        }
        if (mine) lds.store_scores4(i0, s);  // the count cells stay as they are
    };
    auto correction_sweep = [&](auto fast) {
        constexpr uint32_t kTripRows = kUnroll * kWave;
        const uint32_t whole_trips = n_rows_pad / kTripRows;
        uint32_t base = 0;
        if (whole_trips) {
            // two trips per turn, each in registers of its own: the rows of one are on their way while the
            // other is worked on, and nothing is copied
            Trip a = load_trip(std::true_type{}, 0u);
            uint32_t trip = 0;
            for (; trip + 2 <= whole_trips; trip += 2, base += 2 * kTripRows) {
                const Trip b = load_trip(std::true_type{}, base + kTripRows);
                correct_rows(std::true_type{}, fast, base, a);
                if (trip + 2 < whole_trips) a = load_trip(std::true_type{}, base + 2 * kTripRows);
                correct_rows(std::true_type{}, fast, base + kTripRows, b);
            }
            if (trip < whole_trips) {
                correct_rows(std::true_type{}, fast, base, a);
                base += kTripRows;
            }
        }
        if (base < n_rows_pad) correct_rows(std::false_type{}, fast, base, load_trip(std::false_type{}, base));
    };
    // (a slice nothing was streamed into: every row is zero as the last reset left it -- touched stays 0)
    const bool untouched = __builtin_amdgcn_readfirstlane((int)ctx.untouched()) != 0;
    if (untouched) {
    } else if (fast_div) {
        correction_sweep(std::true_type{});
    } else {
        correction_sweep(std::false_type{});
    }
    EPI_STAMP(0)  // correction sweep
    EPI_STOP(256u)
    const uint32_t lane_best = lane_best_f == -INFINITY ? 0u : ord_f32(lane_best_f);  // 0 = none
    const float thr_score = __fdiv_rn(__fmul_rn(nk_f, log_thr), k_f);  // :175 / :146-147

    // ---- select_best_placements (:134-159) + sum_scores (:164-184) --------------------
    // Candidates = every edge whose score reaches tau, the n_sel-th largest of the 64
    // per-lane maxima: at least n_sel edges qualify, usually only a few more.  They are
    // compacted into LDS and ranked by counting, rank = final row (score desc, branch asc).
    // The same sweep accumulates sum_scores relative to the largest term, 10^ref_score:
    //   score_sum = 10^ref_score * (sum_i 10^(score_i - ref_score) + (N - n) * 10^(thr - ref_score))
    // with the relative terms in float32 (v_exp_f32): ~1e-7 relative on score_sum, i.e. on
    // every like_weight_ratio (bar: 1e-5).  Row scores and 10^ref_score stay in double, and
    // so does everything when 10^ref_score could underflow (the score_sum == 0 rule, :243-251).
    const uint32_t keep = uniform(ctx.keep_at_most(p));
    auto *cand = reinterpret_cast<typename WaveLds<CountT>::u32x2_t *>(lds.desc);  // {ord(score), branch}
    constexpr uint32_t kCandCap = Ctx::kCandCap;  // top-k candidates the wave's descriptor list holds
    constexpr int kQ = (int)((kCandCap + kWave - 1) / kWave);  // ... per lane
    constexpr float kLog2Of10 = 3.32192809488736f;
    uint32_t n_sel, n_cand;
    float best_score;
    float rel_sum = 0.0f, rel_sum_b = 0.0f;  // this lane's share of sum_i 10^(score_i - ref_score), in two halves
    bool ranked_in_place = false;  // cand[] already sorted: rank == index
    uint32_t tau = 1;
    if (touched == 0) {  // :141-152: first keep_at_most branches at the threshold score
        n_sel = n_cand = Ctx::kTeam ? 0u : keep;  // (a slice without edges offers no row; the merge decides)
        best_score = thr_score;
        if (!Ctx::kTeam && (uint32_t)lane < keep) cand[lane] = v2u{ord_f32(thr_score), (uint32_t)lane};
        ranked_in_place = true;
    } else {
        n_sel = keep < touched ? keep : touched;  // :137
        // tau = the n_sel-th largest of the 64 lane maxima (1 if fewer lanes than that hold an edge: every
        // edge is a candidate), found bit by bit from the top: one comparison and a ballot per bit, the rest
        // scalar.  Scores of one read lie close together, so the search usually starts below the bits they
        // share with the largest one.  (Seven rounds of wave maximum + knock-out took three times as long.)
        // The search stops kTauStop bits above the end: what it has then is a lower bound of that value, within
        // 2^kTauStop units in the last place (a hundredth in a score of a few hundred), and any tau that at
        // least n_sel edges reach serves -- a lower one only lets the few edges in between into the candidates,
        // which the ranking below drops again.  A step is a chain of compare, scalar count, scalar select: a
        // hundred cycles of latency that nothing else fills.
        constexpr int kTauStop = EPIK_AMD_TAU_STOP;
        const uint32_t top = wave_max_u32(lane_best);  // != 0: touched != 0
        uint32_t prefix = 0;
        int bit = 31;
#pragma unroll
        for (int shared = 20; shared <= 24; shared += 4) {  // try: everything above the low 20 (24) bits as in `top`
            const uint32_t trial = top & ~((1u << shared) - 1u);
            if (bit == 31 && trial != 0 && (uint32_t)__popcll(__ballot(lane_best >= trial)) >= n_sel) prefix = trial, bit = shared - 1;
        }
        auto reached = [&](uint32_t trial) { return (uint32_t)__popcll(__ballot(lane_best >= trial)) >= n_sel; };
        if (((bit - kTauStop + 1) & 1) != 0 && bit >= kTauStop) {  // an odd number of bits: one by itself
            if (reached(prefix | (1u << bit))) prefix |= 1u << bit;
            --bit;
        }
        // two bits per step: the three comparisons do not depend on one another, the chain is half as long
        for (; bit > kTauStop; bit -= 2) {
            const uint32_t t1 = prefix | (1u << (bit - 1)), t2 = prefix | (2u << (bit - 1)), t3 = prefix | (3u << (bit - 1));
            const bool r1 = reached(t1), r2 = reached(t2), r3 = reached(t3);
            prefix = r3 ? t3 : r2 ? t2 : r1 ? t1 : prefix;
        }
        tau = prefix ? prefix : 1u;
        best_score = unord_f32(top);
    }
    EPI_STAMP(1)  // tau
    EPI_STOP(512u)
    const float ref_score = fmaxf(best_score, thr_score);
    const bool relative_sum = ref_score > -280.0f;  // wave-uniform
    // Below 10^-325 a power is zero in double whatever computes it (a fiftieth of the smallest denormal): with the
    // largest term there, sum_scores is 0 term by term and every reported row's power is 0 -- nothing to evaluate.
    // (A protein read of 300 residues lives here: its threshold score alone is 294 * log10(threshold) / 7 = -374;
    // the double-precision loop over every touched row was a quarter of such a read's instructions.)
    const bool all_underflow = ref_score < -325.0f;  // wave-uniform
    if (touched != 0) {
        n_cand = 0;
        // every edge's score is finite, every other row holds -inf: one comparison tells a candidate
        const float tau_f = tau <= 1u ? -FLT_MAX : unord_f32(tau);
        struct Rows {
            float row[kUnroll];
        };
        auto load_rows = [&](auto whole, uint32_t base) {
            constexpr bool kWhole = decltype(whole)::value;  // as in the correction sweep
            Rows r;
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) r.row[u] = -INFINITY;
            const uint32_t i0 = base + 4u * (uint32_t)lane;
            if (kWhole || i0 < n_rows_pad) lds.load_scores4(i0, r.row);
            return r;
        };
        auto scan_rows = [&](uint32_t base, const Rows &r) {
            constexpr int kRows = kUnroll;
            const uint32_t i0 = base + 4u * (uint32_t)lane;
            // -inf where there is no edge (a sum of finite log10 scores never is); exp2(-inf) = 0: rows without
            // an edge add nothing.  Two partial sums side by side: the four rows' arithmetic pairs up.
            float term[kRows];
#pragma unroll
            for (int u = 0; u < kRows; ++u)
                term[u] = __builtin_amdgcn_exp2f(__fmul_rn(__fsub_rn(r.row[u], ref_score), kLog2Of10));
            rel_sum += term[0] + term[2];
            rel_sum_b += term[1] + term[3];
            // a candidate among the trip's 256 rows is rare: one test for all four
            const bool any_cand = r.row[0] >= tau_f || r.row[1] >= tau_f || r.row[2] >= tau_f || r.row[3] >= tau_f;
            if (__ballot(any_cand) != 0) {
#pragma unroll
                for (int u = 0; u < kRows; ++u) {
                    const bool is_cand = r.row[u] >= tau_f;
                    const uint64_t m = __ballot(is_cand);
                    if (m) {
                        const uint32_t slot = n_cand + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                        if (is_cand && slot < kCandCap) cand[slot] = v2u{ord_f32(r.row[u]), i0 + (uint32_t)u};
                        n_cand += (uint32_t)__popcll(m);
                    }
                }
            }
        };
        {
            constexpr uint32_t kTripRows = kUnroll * kWave;
            const uint32_t whole_trips = n_rows_pad / kTripRows;
            uint32_t base = 0;
            if (whole_trips) {
                Rows a = load_rows(std::true_type{}, 0u);  // two trips per turn, as in the correction sweep
                uint32_t trip = 0;
                for (; trip + 2 <= whole_trips; trip += 2, base += 2 * kTripRows) {
                    const Rows b = load_rows(std::true_type{}, base + kTripRows);
                    scan_rows(base, a);
                    if (trip + 2 < whole_trips) a = load_rows(std::true_type{}, base + 2 * kTripRows);
                    scan_rows(base + kTripRows, b);
                }
                if (trip < whole_trips) {
                    scan_rows(base, a);
                    base += kTripRows;
                }
            }
            if (base < n_rows_pad) scan_rows(base, load_rows(std::false_type{}, base));
        }
        rel_sum += rel_sum_b;
        if (n_cand > kCandCap) {
            // Too many ties at tau for the candidate buffer: repeated selection over all
            // edges instead (slow, rare).  Leaves cand[0..n_sel) sorted.
            uint64_t prev = ~0ull;
            for (uint32_t r = 0; r < n_sel; ++r) {
                uint64_t best = 0;
                for (uint32_t i = lane; i < N; i += kWave) {
                    const uint2 cv = lds.load(i);
                    const uint64_t key = ((uint64_t)ord_f32(__uint_as_float(cv.x)) << 32) | (uint64_t)(~i);
                    if (cv.y != 0 && key < prev && key > best) best = key;
                }
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) {
                    const uint64_t o = shfl_xor_u64(best, m);
                    best = o > best ? o : best;
                }
                if (lane == 0) cand[r] = v2u{(uint32_t)(best >> 32), ~(uint32_t)best};
                prev = best;
            }
            n_cand = n_sel;
            ranked_in_place = true;
        }
    }
    EPI_STAMP(2)  // scan sweep
    EPI_STOP(1024u)
    // ---- rank: <= 3 candidates per lane, rank = number of candidates with a larger key --------
    const uint32_t n_q = (n_cand + kWave - 1) / kWave;
    uint64_t my_key[kQ];
    uint32_t my_rank[kQ];
#pragma unroll
    for (int q = 0; q < kQ; ++q) {
        const uint32_t idx = (uint32_t)q * kWave + (uint32_t)lane;
        my_key[q] = 0;
        my_rank[q] = ranked_in_place ? idx : 0u;
        if ((uint32_t)q < n_q && idx < n_cand) {
            const v2u c = cand[idx];
            my_key[q] = ((uint64_t)c.x << 32) | (uint64_t)(~c.y);
        }
    }
    if (!ranked_in_place && n_q == 1) {
        // The usual case, a candidate per lane at most (a dozen of them): candidate j's key comes out of lane j with
        // two v_readlane and is compared in all lanes at once -- no trip to the LDS per candidate.  Lanes without a
        // candidate hold key 0, which outranks nothing; two candidates per turn.
        uint32_t rank_b = 0;
        uint32_t j = 0;
        for (; j + 2 <= n_cand; j += 2) {
            const uint64_t ka = readlane_u64(my_key[0], (int)j), kb = readlane_u64(my_key[0], (int)j + 1);
            my_rank[0] += ka > my_key[0] ? 1u : 0u;
            rank_b += kb > my_key[0] ? 1u : 0u;
        }
        if (j < n_cand) my_rank[0] += readlane_u64(my_key[0], (int)j) > my_key[0] ? 1u : 0u;
        my_rank[0] += rank_b;
    } else if (!ranked_in_place) {
        // cand[j] is read at the same address by every lane (LDS broadcast); four per trip.
        // Entries past n_cand are stale: their key is forced to 0, which outranks nothing.
        for (uint32_t j0 = 0; j0 < n_cand; j0 += kUnroll) {
            v2u c[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) c[u] = cand[j0 + (uint32_t)u];  // < kCandCap + kUnroll: spare entries exist
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const uint64_t kj =
                    (j0 + (uint32_t)u < n_cand) ? (((uint64_t)c[u].x << 32) | (uint64_t)(~c[u].y)) : 0ull;
                my_rank[0] += kj > my_key[0] ? 1u : 0u;
                if (n_q > 1) {
#pragma unroll
                    for (int q = 1; q < kQ; ++q) my_rank[q] += kj > my_key[q] ? 1u : 0u;
                }
            }
        }
    }
    EPI_STAMP(3)  // rank
    EPI_STOP(2048u)
    if constexpr (Ctx::kTeam) {
        // ---- one slice of a team placement: ranked rows and partial sum to the merge area ----------
        double sum;
        if (relative_sum) {
            sum = wave_sum_f64((double)rel_sum);
        } else {  // everything in double, term by term, as place.cpp:178-182
            sum = 0.0;
            for (uint32_t i = lane; i < N && !all_underflow; i += kWave) {
                const uint2 cv = lds.load(i);
                if (cv.y != 0) sum += pow10_f64((double)__uint_as_float(cv.x));
            }
            sum = wave_sum_f64(sum);
        }
        EPI_STAMP(4)  // partial sum
        ctx.before_publish();
        EPI_STAMP(5)  // waiting for the previous merge
#pragma unroll
        for (int q = 0; q < kQ; ++q) {
            if ((uint32_t)q < n_q && my_key[q] != 0 && my_rank[q] < n_sel) {
                const uint32_t row = ~(uint32_t)my_key[q];
                ctx.cand[my_rank[q]] = v4u{(uint32_t)(my_key[q] >> 32), ctx.branch_base() + row,
                                           (uint32_t)lds.count[row] & ~(uint32_t)lds.kSeen, 0u};
            }
        }
        for (uint32_t r = n_sel + (uint32_t)lane; r < keep; r += kWave) ctx.cand[r] = v4u{0u, 0u, 0u, 0u};
        if (lane == 0) {
            ctx.partial->touched = touched;
            ctx.partial->relative = relative_sum ? 1u : 0u;
            ctx.partial->ref_score = ref_score;
            ctx.partial->sum = sum;
        }
        EPI_STAMP(6)  // publish
        if (!untouched) lds.clear(n_rows_pad);
        EPI_STAMP(7)  // clear
        (void)read;
        return;
    }
    // ---- 10^score of every row that will be reported (:254), the lanes side by side.  The row of
    // rank 0 carries best_score: its power is rows[0]'s (:191) and, whenever best_score is the
    // reference point of the relative sum, 10^ref_score as well -- one exp10 instead of three.
    double my_power[kQ];
    double best_power = 0.0;
#pragma unroll
    for (int q = 0; q < kQ; ++q) {
        my_power[q] = 0.0;
        if ((uint32_t)q < n_q) {
            const bool has_row = my_key[q] != 0 && my_rank[q] < n_sel;
            if (has_row && !all_underflow) my_power[q] = pow10_f64((double)unord_f32((uint32_t)(my_key[q] >> 32)));
            const uint64_t first = __ballot(has_row && my_rank[q] == 0);
            if (first)
                best_power = __longlong_as_double(
                    (long long)readlane_u64((uint64_t)__double_as_longlong(my_power[q]), __builtin_ctzll(first)));
        }
    }
    double score_sum;
    {
        const float not_placed = (float)N - (float)touched;  // :174
        if (relative_sum) {
            double rel = wave_sum_f64((double)rel_sum);
            if (not_placed != 0.0f)
                rel += (double)(not_placed *
                                __builtin_amdgcn_exp2f(__fmul_rn(__fsub_rn(thr_score, ref_score), kLog2Of10)));
            const double ref_power = (ref_score == best_score) ? best_power : pow10_f64((double)ref_score);
            score_sum = ref_power * rel;
        } else {
            // everything in double, term by term, as place.cpp:174-183
            double sum_placed = 0.0;
            if (!all_underflow) {
                for (uint32_t i = lane; i < N; i += kWave) {
                    const uint2 cv = lds.load(i);
                    if (cv.y != 0) sum_placed += pow10_f64((double)__uint_as_float(cv.x));
                }
                sum_placed = wave_sum_f64(sum_placed);
            }
            score_sum = all_underflow ? 0.0 : (double)not_placed * pow10_f64((double)thr_score) + sum_placed;
        }
    }
    const double keep_factor = (score_sum == 0.0) ? 0.0 : p.keep_factor;  // :247-251

    // ---- LWR (:241-264), filter_by_ratio (:188-199) ---------------------------------------------
    const double best_ratio =
        (score_sum == 0.0 || best_power == 0.0) ? 0.0 : best_power / score_sum;  // :191, rows[0]
    const double ratio_threshold = best_ratio * keep_factor;                      // :192
    double my_lwr[kQ];
    uint64_t kept_ranks = 0;  // bit r set <=> the row of rank r passes the filter
#pragma unroll
    for (int q = 0; q < kQ; ++q) {
        my_lwr[q] = 0.0;
        if ((uint32_t)q < n_q) {
            const bool has_row = my_key[q] != 0 && my_rank[q] < n_sel;
            if (has_row && score_sum != 0.0 && my_power[q] != 0.0) my_lwr[q] = my_power[q] / score_sum;  // :255-262
            if (has_row && my_lwr[q] >= ratio_threshold) kept_ranks |= 1ull << my_rank[q];            // :197
        }
    }
    kept_ranks = wave_or_u64(kept_ranks);
#pragma unroll
    for (int q = 0; q < kQ; ++q) {
        if ((uint32_t)q < n_q) {
            const bool has_row = my_key[q] != 0 && my_rank[q] < n_sel;
            if (has_row && ((kept_ranks >> my_rank[q]) & 1ull)) {
                const uint32_t slot = (uint32_t)__popcll(kept_ranks & ((1ull << my_rank[q]) - 1ull));
                const uint32_t branch = ~(uint32_t)my_key[q];
                epik_amd_placement out;
                out.branch = branch;
                out.score = unord_f32((uint32_t)(my_key[q] >> 32));
                out.lwr = my_lwr[q];
                p.rows[read * keep + slot] = out;
                if (p.kmer_counts)
                    p.kmer_counts[read * keep + slot] = (touched && branch < N) ? ((uint32_t)lds.count[branch] & ~(uint32_t)lds.kSeen) : 0u;
            }
        }
    }
    if (lane == 0) p.n_rows[read] = (uint32_t)__popcll(kept_ranks);

    // ---- reset the wave's vectors for its next read (place.cpp:335-342) -------------
    lds.clear(n_rows_pad);
}
// out of line (see above); team_stream_kernel, which has the registers, takes the body inline
template <typename Layout, typename CountT, typename Ctx>
__device__ __attribute__((noinline)) void place_epilogue(const PlaceParams *__restrict__ kp, WaveLds<CountT> lds,
                                                         uint64_t read, uint64_t n_kmers, Ctx ctx)
{
    place_epilogue_body<Layout, CountT, Ctx>(kp, lds, read, n_kmers, ctx);
}


// Algorithmic bytes of SURVEY.md 8(d), sum over the reads of a batch of
//   L + 8 * n_kmers + 8 * sum |posting list| + 16 * rows_out:
// one thread per read, plain loops; list_length(key) = number of postings of a k-mer code.
template <typename ListLength>
__device__ __forceinline__ void algorithmic_bytes_block(const PlaceParams &p, unsigned long long *total,
                                                        ListLength &&list_length)
{
    const uint64_t read = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bytes = 0;
    if (read < p.n_reads) {
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
                    entries += list_length((uint32_t)key);
                } else {
                    weight = 1;
                    for (uint32_t j = amb_pos + 1; j < k; ++j) weight *= p.alphabet_size;
                    for (uint32_t st = 0; st < p.alphabet_size; ++st)
                        if ((amb_cls >> st) & 1u) entries += list_length((uint32_t)(key + (uint64_t)st * weight));
                }
            }
            bytes += 8ull * entries;
            if (p.n_rows && p.n_rows[read] != kCountsTooNarrow) bytes += 16ull * p.n_rows[read];
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

}  // namespace
}  // namespace epik_amd
#endif
