// db_layout.h -- how the phylo-k-mer database lies in HBM, and the LDS arithmetic that decides it.
// Plain C++ (no HIP): shared by the kernels (place_kernel.h), the host-side image builder
// (db_image.cpp) and its CPU tests.  Internal; the public boundary is include/epik_amd.h.
#ifndef EPIK_AMD_DB_LAYOUT_H
#define EPIK_AMD_DB_LAYOUT_H

#include <stddef.h>
#include <stdint.h>

#ifndef EPIK_AMD_TILES_PER_PASS
#define EPIK_AMD_TILES_PER_PASS 3  // 64-character tiles of a read encoded, looked up and laid out per pass (experiments: -D)
#endif
#ifndef EPIK_AMD_RING
#define EPIK_AMD_RING 8  // posting-chunk loads kept in flight per wave (power of two)
#endif

namespace epik_amd {

// Layouts of the one-wavefront-per-read kernels (place_kernel.hip documents them).
enum class DbLayout : int {
    kCompact32 = 0,  // CSR: 32-bit offsets[num_keys + 1], 8-byte {f32 score, u32 cell} postings back to back
    kCompact64 = 1,  // same with 64-bit offsets
    kPacked = 2,     // 8-byte {len, first 128-byte line} entry per k-mer code, lists on whole lines,
                     // 6 bytes per posting: f32 score[cnt] then u16 cell[cnt] per chunk of <= 64
    kPaired = 3,     // the same lists; the table keyed by the (k-1)-mer two consecutive k-mers share
                     // (4-letter alphabets): one table line per two lookups, 16 bytes per code
    kFiltered = 4,   // kPacked behind a presence filter keyed the same way (other alphabets, sparse
                     // databases): one filter word per two lookups, the table only for present codes
    kTeam = 5,       // the team kernel's layout (team_kernel.hip): every list pre-split into one
                     // sublist per slice of the branch range, a {line, len[W]} entry per code and pass
};

// Width of the per-branch k-mer counts in LDS: 16 bits by default (reads of up to 32767 k-mers),
// 32 for longer reads, 8 (reads of up to 255 k-mers) when that lets more waves share a CU.
enum CountBits : int { kCounts8 = 0, kCounts16 = 1, kCounts32 = 2 };

// ---- gfx950 numbers the geometry is computed from -------------------------------------------
constexpr uint32_t kLdsPerCu = 160u * 1024u;  // bytes
constexpr uint32_t kLdsGranule = 1280u;       // LDS is handed out in 128ths of a CU's 160 KiB (measured)
constexpr uint32_t kWaveKernelWavesPerCu = 20u;  // place_reads_kernel: __launch_bounds__(256, 5)
// team_place_kernel: __launch_bounds__(256, 3) with 4 waves (no spills; LDS leaves 3 workgroups at most where
// this kernel is chosen), (512, 4) with 8
#ifndef EPIK_AMD_STREAM_OCC
#define EPIK_AMD_STREAM_OCC 0  // experiments: waves per SIMD the team kernels are compiled for (0: 3 with 4-wave teams, 4 with 8)
#endif
constexpr uint32_t team_waves_per_simd(int waves) { return EPIK_AMD_STREAM_OCC ? (uint32_t)EPIK_AMD_STREAM_OCC : waves <= 4 ? 3u : 4u; }
constexpr uint32_t team_kernel_waves_per_cu(int waves) { return 4u * team_waves_per_simd(waves); }

constexpr uint32_t kWaveDescBytes = (EPIK_AMD_TILES_PER_PASS * 64u + EPIK_AMD_RING) * 8u;
// LDS bytes of one wave of the one-wavefront-per-read kernels: scores + counts + chunk descriptors
constexpr uint32_t wave_lds_bytes(uint32_t n_pad, int counts)
{
    return (n_pad * (4u + (1u << counts)) + kWaveDescBytes + 15u) & ~15u;
}
// Resident waves per CU of those kernels: workgroups of 4, 2 or 1 independent waves, whichever keeps most
constexpr uint32_t wave_kernel_resident_waves(uint32_t n_pad, int counts)
{
    uint32_t best = 0;
    for (uint32_t wpb = 4; wpb >= 1; wpb >>= 1) {
        const uint32_t block = wpb * wave_lds_bytes(n_pad, counts);
        if (block > kLdsPerCu) continue;
        const uint32_t units = (block + kLdsGranule - 1) / kLdsGranule;
        uint32_t blocks = 128u / (units ? units : 1u);
        if (blocks * wpb > kWaveKernelWavesPerCu) blocks = kWaveKernelWavesPerCu / wpb;
        if (blocks * wpb > best) best = blocks * wpb;
    }
    return best;
}

// ---- team kernel ---------------------------------------------------------------------------------
#ifndef EPIK_AMD_TEAM_RING
#define EPIK_AMD_TEAM_RING 8
#endif
constexpr uint32_t kTeamRing = EPIK_AMD_TEAM_RING;  // posting-chunk loads in flight per wave (16 measured slower: r02 notes in DESIGN.md)
constexpr uint32_t kTeamDescCap = 64u - kTeamRing;  // chunk descriptors per slice and round (a multiple of the ring; + one trip of spare entries: 64)
constexpr uint32_t kTeamCandCap = 60;  // top-k candidates of a slice (+ 4 spare entries = the list's 64)
// bytes of one table entry {u32 line, u16 len[W]}: 8 with two slices per pass (round 5: the front kernel is bound by the
// table lines it fetches, and a block of the paired table is then 64 bytes -- two blocks to a line, the table of a
// k = 10 database 16 MB instead of 32), else padded to 16 / 32 / 64
constexpr int team_entry_bytes(int waves) { return waves <= 2 ? 8 : waves <= 6 ? 16 : waves <= 14 ? 32 : 64; }
// LDS rows of a slice: its branches + the dummy row, to a multiple of 16 (the epilogue's sweeps mask their last trip)
constexpr uint32_t team_rows_pad(uint32_t slice_rows) { return (slice_rows + 1u + 15u) & ~15u; }
constexpr uint32_t team_slice_bytes(uint32_t rows_pad, int counts) { return (rows_pad * (4u + (1u << counts)) + 15u) & ~15u; }
// a slice's descriptor list (one round + one trip of spare entries) also holds its top-k candidates
constexpr uint32_t team_desc_bytes(uint32_t /*keep*/)
{
    static_assert((kTeamCandCap + 4u) * 8u <= (kTeamDescCap + kTeamRing) * 8u, "the candidates live in the descriptor list");
    return ((kTeamDescCap + kTeamRing) * 8u + 15u) & ~15u;
}
// LDS bytes of a workgroup: slices, descriptor lists, tile totals + flags, partial sums, the slices' ranked rows
constexpr size_t team_lds_bytes(int waves, uint32_t passes, uint32_t slice_bytes, uint32_t desc_bytes, uint32_t keep)
{
    return (size_t)waves * slice_bytes + (size_t)waves * desc_bytes + ((size_t)waves * waves + 4) * 4 +
           (size_t)waves * passes * 24 + (size_t)waves * passes * keep * 16;
}
// team_stream_kernel (team_stream.hip): workgroups of kStreamWaves independent waves, each with one slice's rows
// and descriptor list; with 8 slices per pass two workgroups share a read.  Compiled for 4 waves per SIMD with
// 4 slices (128 vector registers: the LDS of a large tree leaves room for three of its waves, and the fourth
// wave's registers go to the front and merge kernels, which run beside it) and 5 with 8 (the slice epilogue
// needs 70 VGPRs, the streaming loop fewer; only the cold ambiguous sweep spills).
// Workgroups of FOUR or (round 5) TWO waves -- `bw`, a template parameter of the kernel, chosen per geometry by
// stream_block_waves(): the waves of a workgroup share nothing but the LDS allocation, which the hardware hands out in
// granules of 1 280 bytes per workgroup, and halves of a four-wave workgroup sometimes fit where a whole one does not
// (waves per CU by LDS, four / two: N = 2 999 with two slices per pass 16 / 18, 3 999 12 / 14, 5 999 with four 16 / 18).
// Measured, M reads/s, four / two: N = 2 999 121.3 / 128.2, 3 999 (two slices) 96.0 / 107.5, 5 999 86.2 / 90.0 -- and where
// the waves stay the same two-wave workgroups LOSE 2-3 % (3 499: 120.5 / 118.4, 7 499: 81.8 / 79.3): two only where they
// put more waves on a CU.
constexpr int kStreamWaves = 4;  // the larger of the two
// ... each wave with one slice of some read: with W >= bw slices per pass W / bw consecutive workgroups share a read, with
// fewer a workgroup holds the slices of bw / W reads
constexpr uint32_t stream_parts(int slices_per_pass, int bw) { return slices_per_pass >= bw ? (uint32_t)(slices_per_pass / bw) : 1u; }
constexpr uint32_t stream_reads_per_block(int slices_per_pass, int bw) { return slices_per_pass >= bw ? 1u : (uint32_t)(bw / slices_per_pass); }
constexpr size_t stream_lds_bytes(uint32_t slice_bytes, uint32_t desc_bytes, int bw) { return (size_t)bw * (slice_bytes + desc_bytes); }
constexpr uint32_t stream_blocks_by_lds(size_t lds_bytes)
{
    if (lds_bytes > kLdsPerCu) return 0;
    const uint32_t units = (uint32_t)((lds_bytes + kLdsGranule - 1) / kLdsGranule);
    return 128u / (units ? units : 1u);
}
// The streaming kernel comes in two builds.  WIDE (2 or 4 slices per pass, slices so large that LDS holds twelve waves of it
// on a CU at most -- three per SIMD, whatever the registers): 168 vector registers, which the slice
// epilogue over the touched quads (team_epilogue.hpp) holds its rows in.  LEAN (everything else): the 96 registers of
// five waves per SIMD -- the hardware fills a CU with as many workgroups as registers and LDS allow, whatever the
// kernel was compiled for, and on the small slices of a mid-size tree that is twenty waves (measured, round 4: the
// wide build on slices of 1 000 rows places 74 M reads/s, the lean one 100 M).
constexpr bool stream_wide(int slices_per_pass, size_t lds_bytes, int bw)
{
    return slices_per_pass <= 4 && stream_blocks_by_lds(lds_bytes) != 0 && stream_blocks_by_lds(lds_bytes) * (uint32_t)bw <= 13u;
}
// (what the builds are COMPILED for -- the lean build of 4 slices for three waves per SIMD like the wide one: held to the
// 96 registers of five it spills seven of them, left alone it takes 96 and no scratch -- and what a CU then holds)
constexpr uint32_t stream_waves_per_simd(int slices_per_pass)
{
    return EPIK_AMD_STREAM_OCC ? (uint32_t)EPIK_AMD_STREAM_OCC : slices_per_pass <= 4 ? 3u : 5u;
}
constexpr uint32_t stream_resident_blocks(int slices_per_pass, size_t lds_bytes, int bw)
{
    const uint32_t by_lds = stream_blocks_by_lds(lds_bytes);
    const uint32_t by_regs = (stream_wide(slices_per_pass, lds_bytes, bw) ? 12u : 20u) / (uint32_t)bw;
    return by_lds < by_regs ? by_lds : by_regs;
}
constexpr uint32_t stream_resident_waves(int slices_per_pass, uint32_t slice_bytes, uint32_t desc_bytes, int bw)
{
    return stream_resident_blocks(slices_per_pass, stream_lds_bytes(slice_bytes, desc_bytes, bw), bw) * (uint32_t)bw;
}
// the workgroup that leaves most waves on a CU (a tie: four)
constexpr int stream_block_waves(int slices_per_pass, uint32_t slice_bytes, uint32_t desc_bytes)
{
    return stream_resident_waves(slices_per_pass, slice_bytes, desc_bytes, 2) > stream_resident_waves(slices_per_pass, slice_bytes, desc_bytes, 4) ? 2 : 4;
}
constexpr uint32_t team_resident_blocks(int waves, size_t lds_bytes)
{
    if (lds_bytes > kLdsPerCu) return 0;
    const uint32_t units = (uint32_t)((lds_bytes + kLdsGranule - 1) / kLdsGranule);
    const uint32_t by_lds = 128u / (units ? units : 1u), by_regs = team_kernel_waves_per_cu(waves) / (uint32_t)waves;
    return by_lds < by_regs ? by_lds : by_regs;
}

}  // namespace epik_amd
#endif
