/*
 * epik_amd.h -- C ABI of the MI355X placement engine (libepik_amd.so).
 *
 * This is the drop-in boundary for EPIK's one hot path, the per-read loop of
 * `epik::placer` (reference: epik/include/epik/place.h:81-140,
 * epik/src/epik/place.cpp:201-440).  The reference has no FFI layer; the
 * narrowest seam the path sits behind is the C++ class `epik::placer`:
 *
 *     placer(const i2l::phylo_kmer_db&, const i2l::phylo_tree&,
 *            size_t keep_at_most, double keep_factor, size_t max_threads);   place.h:94-95
 *     placed_collection place(const std::vector<i2l::seq_record>&, size_t);  place.h:103
 *
 * Each entry point below names the reference interface it replaces.  Plain
 * pointers and sizes only; no C++ or torch types cross the boundary; no
 * exception crosses it either (the reference throws std::runtime_error,
 * place.cpp:104-108,430-433; here every call returns a status code and
 * epik_amd_last_error() holds the message).
 *
 * There is no CPU fallback: every entry point that computes fails with
 * EPIK_AMD_ERR_NO_DEVICE when no HIP device is usable.
 */
#ifndef EPIK_AMD_H
#define EPIK_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EPIK_AMD_ABI_VERSION 3

enum epik_amd_status {
    EPIK_AMD_OK = 0,
    EPIK_AMD_ERR_INVALID = 1,   /* bad argument / inconsistent database */
    EPIK_AMD_ERR_NO_DEVICE = 2, /* no usable HIP device (no CPU fallback exists) */
    EPIK_AMD_ERR_HIP = 3,       /* HIP runtime error, see epik_amd_last_error() */
    EPIK_AMD_ERR_UNSUPPORTED = 4 /* configuration outside what the kernels cover */
};

/* i2l::pkdb_value {branch, score}: one phylo-k-mer posting (main.cpp:257,
 * place.cpp:358).  branch = post-order node id, score = log10 probability. */
typedef struct {
    uint32_t branch;
    float score;
} epik_amd_pkdb_value;

/* One reported placement: the fields of epik::impl::placement computed on the
 * hot path (place.h:45-56: branch_id, score, weight_ratio).  distal_length and
 * pendant_length are per-branch constants the host joins (place.cpp:435-437). */
typedef struct {
    uint32_t branch;
    float score;
    double lwr;
} epik_amd_placement;

/*
 * What `epik::placer`'s constructor receives through `db` and `tree`
 * (place.cpp:83-96), flattened: the phylo-k-mer database as a CSR-like layout
 * (k-mer code -> posting list) replacing the i2l hash map behind
 * `phylo_kmer_db::search` (place.cpp:300,311).
 *
 * The code of a k-mer is the base-`alphabet_size` number of its k state codes, first
 * character most significant; num_keys must equal alphabet_size^kmer_size.  Two forms:
 *   dense  (keys == NULL): offsets[code] .. offsets[code+1] delimit the postings of `code` in
 *          values[]; offsets has num_keys + 1 entries -- 8 bytes per POSSIBLE k-mer, 10 GB for
 *          amino k = 7 whatever the database holds;
 *   sparse (keys != NULL, ABI 3): keys[num_present] are the codes that have a list, strictly
 *          ascending, and offsets[i] .. offsets[i+1] delimit the postings of keys[i]; offsets has
 *          num_present + 1 entries -- memory per PRESENT key, as the hash map behind
 *          phylo_kmer_db::search (place.cpp:300,311) has it.
 * All pointers are HOST pointers; create() streams them to the device (it keeps no copy and
 * needs no array of its own per code), the caller may free them afterwards.
 */
typedef struct {
    uint32_t abi_version;    /* EPIK_AMD_ABI_VERSION */
    uint32_t kmer_size;      /* db.kmer_size()                       place.cpp:87 */
    uint32_t alphabet_size;  /* 4 = nucl (epik-dna), 20 = amino (epik-aa); epik/CMakeLists.txt:72,124 */
    uint32_t num_branches;   /* tree.get_node_count()                place.cpp:92 */
    uint32_t keep_at_most;   /* --keep-at-most, default 7            main.cpp:219 */
    uint32_t offset_bits;    /* width of offsets[]: 32 or 64 */
    double keep_factor;      /* --keep-factor, default 0.01          main.cpp:220 */
    float threshold;         /* i2l::score_threshold(omega, k)       place.cpp:87 */
    float log_threshold;     /* std::log10(threshold), as float      place.cpp:88 */
    uint64_t num_keys;       /* alphabet_size ^ kmer_size */
    uint64_t num_entries;    /* the last offset = db.get_num_entries_loaded() */
    const void *offsets;     /* uint32_t/uint64_t [num_keys + 1] (dense) or [num_present + 1] (sparse) */
    const epik_amd_pkdb_value *values; /* [num_entries]; branches distinct within one list */
    const uint32_t *char_class; /* [256]: bit s set <=> the character may be state s;
                                   popcount 1 plain, >1 ambiguous, 0 invalid
                                   (i2l::to_kmers<one_ambiguity_policy>, place.cpp:294) */
    int32_t device;          /* HIP device ordinal */
    uint32_t shard;          /* 0: a whole database.  g | G << 16 (G >= 2, g < G): the descriptor holds shard g of G of one
                                ALREADY -- the lists of the codes with code % G == g and no others (k-mer-space shard,
                                below): create() then sizes its table for those codes alone, as create_sharded() does */
    const uint32_t *keys;    /* NULL: dense form; else [num_present] ascending codes (sparse form) */
    uint64_t num_present;    /* sparse form: number of codes that have a list */
} epik_amd_placer_desc;

typedef struct epik_amd_placer epik_amd_placer;

/* Number of visible HIP devices (0 when there is none / no driver). */
int epik_amd_device_count(void);

/* Message of the last failing call on this thread. */
const char *epik_amd_last_error(void);

/* Replaces epik::placer::placer (place.cpp:83-126): uploads the database to
 * the device's HBM once and precomputes what the kernel needs. */
int epik_amd_placer_create(const epik_amd_placer_desc *desc, epik_amd_placer **out);

/*
 * What create() would decide for this database on a device with `free_bytes` of free memory, and how
 * large the device image is -- without a device (capacity planning against 288 GB; no reference
 * counterpart: the reference keeps its hash map in host RAM, main.cpp:277).  `kernel` 0: one wavefront
 * places a read (trees whose score vector leaves enough waves on a CU); 1: the branch range is split into
 * team_waves * team_passes slices of `slice_rows` branches and a wavefront places one slice of a read
 * (large trees; place.cpp:92-96 bounds the tree by nothing, and neither does this).
 * The environment overrides of create() (EPIK_AMD_LAYOUT, EPIK_AMD_KERNEL) apply here too.
 */
typedef struct {
    uint32_t kernel;         /* 0 = one wavefront per read, 1 = one workgroup per read */
    uint32_t layout;         /* 0/1 compact CSR (32/64-bit offsets), 2 packed, 3 paired, 4 filtered, 5 sliced */
    uint32_t team_waves;
    uint32_t team_passes;
    uint32_t slice_rows;
    uint32_t resident_waves[3]; /* one-wavefront kernel: waves per CU with 8/16/32-bit counts (0: does not fit) */
    uint64_t table_bytes;
    uint64_t filter_bytes;
    uint64_t posting_bytes;
    uint64_t kept_entries;   /* postings this placer keeps (its shard) */
    uint32_t run_coded;      /* 1: lists that are one ascending run of branches are stored without their cells
                                (4 bytes per posting; databases well beyond the Infinity Cache) */
    uint32_t posting_bytes_is_bound; /* epik_amd_placer_plan_sizes, sliced layout: posting_bytes is an upper bound */
} epik_amd_plan;
int epik_amd_placer_plan(const epik_amd_placer_desc *desc, uint32_t shard_index, uint32_t shard_count,
                         uint64_t free_bytes, epik_amd_plan *plan);
/*
 * The same plan from the SIZES of a database alone -- before it exists, or is at hand: the tree, the key space, and a
 * histogram of the posting lists the placer (shard shard_index of shard_count) will keep.  The reference's only
 * capacity control is --max-ram on the host (main.cpp:252-266, README.md:121-128); this answers "how many GPUs'
 * HBM does a database of this shape take, table included" without touching a posting.  kernel, layout, geometry,
 * resident_waves, table_bytes, filter_bytes, kept_entries and run_coded are what epik_amd_placer_plan() gives on
 * a database with these lists; posting_bytes too for the layouts of the one-wavefront kernel, and an upper bound
 * (a few percent: how a list falls over the slices of the branch range is not in a histogram) for the sliced
 * layout -- posting_bytes_is_bound says which.  No device needed.
 */
typedef struct {
    uint64_t length;         /* postings of a list */
    uint64_t lists;          /* lists of that length this placer keeps */
    uint64_t lists_in_runs;  /* ... of which are one ascending run of branches b, b + 1, ... (<= lists; 0 if unknown) */
} epik_amd_list_bin;
int epik_amd_placer_plan_sizes(uint32_t kmer_size, uint32_t alphabet_size, uint32_t num_branches, uint32_t keep_at_most,
                               const epik_amd_list_bin *bins, uint64_t n_bins, uint32_t shard_index, uint32_t shard_count,
                               uint64_t free_bytes, epik_amd_plan *plan);
/* The image create() uploads for that plan, written front to back into host buffers of
 * plan->table_bytes / filter_bytes / posting_bytes (NULL = that part is produced and dropped).
 * Host only; create() streams the same bytes to the device without holding them. */
int epik_amd_placer_build_image(const epik_amd_placer_desc *desc, uint32_t shard_index, uint32_t shard_count,
                                uint64_t free_bytes, void *table, void *filter, void *postings);

/* Replaces ~placer (place.h:100). */
void epik_amd_placer_destroy(epik_amd_placer *p);

/*
 * Replaces the OpenMP loop of epik::placer::place (place.cpp:218-268): places n
 * reads that the host has already de-duplicated (place.cpp:207-212).
 *   seqs / seq_offsets[n+1]: concatenated read bytes (HOST).
 *   rows[n * keep_at_most], n_rows[n]: placements in final order (sorted by
 *     score descending -- ties by branch ascending --, LWR-filtered,
 *     place.cpp:240-267); n_rows[i] == 0 iff read i is shorter than k.
 *   kmer_counts[n * keep_at_most] (nullable): placement::count (place.h:53).
 * Synchronous: copies in, runs the kernel, copies out.
 */
int epik_amd_placer_place(epik_amd_placer *p, const char *seqs, const uint64_t *seq_offsets,
                          uint64_t n, epik_amd_placement *rows, uint32_t *n_rows,
                          uint32_t *kmer_counts);

/*
 * Same computation with every buffer already resident in device memory, enqueued
 * on `stream` (a hipStream_t passed as void*; NULL = the default stream) without
 * synchronising.  d_kmer_counts may be NULL.  The launches of one handle share its scratch
 * memory (large trees): enqueue them on one stream, or order them.
 */
int epik_amd_placer_place_device(epik_amd_placer *p, const void *d_seqs, const void *d_seq_offsets,
                                 uint64_t n, void *d_rows, void *d_n_rows, void *d_kmer_counts,
                                 void *stream);

/*
 * Measurement helper: algorithmic bytes of a device-resident
 * batch, sum over reads of L + 8*n_kmers + 8*sum|posting list| + 16*rows_out,
 * with rows_out taken from d_n_rows (NULL counts no output rows).  Synchronous.
 */
int epik_amd_placer_algorithmic_bytes(epik_amd_placer *p, const void *d_seqs,
                                      const void *d_seq_offsets, uint64_t n, const void *d_n_rows,
                                      void *stream, uint64_t *bytes_out);

/*
 * The kernel keeps one k-mer count per branch in LDS (the reference counts in size_t, place.h:86):
 * 16 bits by default (reads of up to 32767 k-mers), 32 bits for longer reads, 8 bits (reads of up to
 * 255 k-mers) when that lets more wavefronts share a CU -- large trees.  epik_amd_placer_place()
 * looks at the batch and chooses by itself; for the device-pointer entry points the caller chooses
 * here: enabled = 0 back to the default (16 bits, and place() chooses again), 1 = 32 bits, 2 = 8 bits
 * (EPIK_AMD_ERR_UNSUPPORTED for trees too large for that kernel).  A read with more k-mers than the
 * counts of a device-pointer launch hold is not placed and says so: n_rows == EPIK_AMD_ROWS_COUNTS_TOO_NARROW
 * (a read shorter than k has n_rows == 0).
 */
#define EPIK_AMD_ROWS_COUNTS_TOO_NARROW 0xffffffffu
int epik_amd_placer_set_wide_counts(epik_amd_placer *p, int enabled);
/* The same choice epik_amd_placer_place() makes, for the device-pointer entry points: the counts
 * that fit a batch whose longest read has `longest_read` characters. */
int epik_amd_placer_choose_counts(epik_amd_placer *p, uint64_t longest_read);

/*
 * Database larger than one GPU's memory: k-mer-space shard (BASELINE configs[4]).
 * No reference counterpart -- the reference keeps one database in host RAM (main.cpp:277).
 *
 * Shard g of G holds the posting lists of the k-mer codes with code % G == g.  Either hand create()
 * a descriptor that holds only those lists (every other code an empty list: no process then ever
 * holds the whole database -- create() streams what it is given to the device and keeps no copy), or
 * let epik_amd_placer_create_sharded() pick them out of a whole database
 * (create() == create_sharded(desc, 0, 1, out)).  Every shard then sees ALL reads of a batch:
 *
 *   accumulate_device : per read, the raw float32 score sums and the k-mer counts of this shard's
 *                       lists, d_scores float32 / d_counts uint16 = [n][num_branches]
 *                       (place.cpp:349-371 without 418-422); reads of up to 65535 k-mers (a longer one
 *                       leaves an all-zero vector and finish marks it EPIK_AMD_ROWS_COUNTS_TOO_NARROW:
 *                       the partial lists below have no such limit);
 *   (caller)          : adds d_scores and d_counts over the shards -- one all-to-all + a sum in rank
 *                       order, each GPU keeping the totals of its own reads (epik_amd/dist.py);
 *   finish_device     : correction, top-k, like-weight-ratio and filter on the totals
 *                       (place.cpp:418-422, 134-199, 241-267); rows as epik_amd_placer_place_device.
 *
 * Ambiguous k-mers (place.cpp:373-415): only the first ambiguous key that reaches a branch scores
 * it, first over the WHOLE database (:385-388).  A read that may hold an ambiguous character gets a
 * slot: d_amb_slot[i] >= 0 (int32, -1 = none) is its row in d_amb_order (uint32) and d_amb_avg
 * (float32), both [slots][num_branches].  accumulate fills that row, per branch, with the order
 * (k-mer position * alphabet_size + state) of the first ambiguous key of THIS shard that reached the
 * branch (0xffffffff: none) and its average probability (:400-402); the caller keeps, per branch, the
 * average of the smallest order over the shards (0 where there is none); finish adds it after the
 * exact scores and counts one k-mer, as the one-pass loop does (:409-410).  With d_amb_slot == NULL a
 * shard scores its ambiguous k-mers by itself -- the one-pass result with one shard only.
 *
 * With one shard the two calls give exactly the rows of place_device.  With several, a branch's
 * float32 adds happen in a different order (per shard, then over shards): scores agree to float32
 * rounding, like-weight-ratios within the 1e-5 bar.
 */
int epik_amd_placer_create_sharded(const epik_amd_placer_desc *desc, uint32_t shard_index,
                                   uint32_t shard_count, epik_amd_placer **out);
int epik_amd_placer_accumulate_device(epik_amd_placer *p, const void *d_seqs, const void *d_seq_offsets,
                                      uint64_t n, void *d_scores, void *d_counts, const void *d_amb_slot,
                                      void *d_amb_order, void *d_amb_avg, void *stream);
int epik_amd_placer_finish_device(epik_amd_placer *p, const void *d_seq_offsets, uint64_t n,
                                  const void *d_scores, const void *d_counts, const void *d_amb_slot,
                                  const void *d_amb_avg, void *d_rows, void *d_n_rows, void *d_kmer_counts,
                                  void *stream);

/*
 * The same two halves with partial LISTS instead of dense vectors (large trees: the handles whose
 * epik_amd_placer_partial_info() says lists == 1).  A shard's lists reach a small part of a large tree per read
 * (N = 9 999, 8 shards: ~5 % of the branches), so accumulate leaves, per read and SLICE of the branch range
 * (slices * slice_rows >= num_branches; the same on every shard: the geometry depends on the tree alone), only the
 * rows that received a k-mer:
 *
 *   entry  {f32 sum, u32 row | count << 16}      8 bytes (counts of 8 or 16 bits: reads of up to 32767 k-mers)
 *          {f32 sum, u32 row, u32 count, u32 0}  16 bytes (32-bit counts, chosen for longer reads)
 *          row = branch - slice * slice_rows; any order inside a list, every row at most once;
 *   index  [n][slices] {u32 first, u32 count}: the list of (read, slice) = entries first .. first + count - 1 of
 *          the read's PART.
 *
 * The n reads form n_parts equal runs of ceil(n / n_parts) reads -- part r is what finisher r needs, and its
 * entries lie together: d_part_entries[r] (uint64) says how many entries part r takes in d_entries, the parts one
 * after the other from the start (lists are laid out before they are filled, from an upper bound -- the postings
 * of the slice's sublists --: a part is that bound added up, a few percent more than the entries its index names).
 * If the parts together exceed entries_cap the lists that found no room are marked count == 0xffffffff and the
 * call must be repeated with a larger buffer (read d_part_entries after the stream has finished; it is always
 * complete).  entries_cap < 2^32.
 *
 * finish_lists takes, for its n reads (one part), what each of the n_shards shards left for them: d_entries[g] =
 * the start of the part's entries as shard g wrote them, d_index[g] = the part's [n][slices] index of shard g
 * (HOST arrays of device pointers), and adds the lists of a slice into LDS in shard order -- the float32 sums
 * are those of the dense exchange's rank-order sum (0 + x is x) -- then finishes as finish_device does.  All
 * shards and the finisher must run with the same count width (epik_amd_placer_choose_counts with the batch's
 * longest read on each).  The ambiguous records cross as with the dense calls.
 * accumulate_lists and finish_lists of ONE handle may run side by side on two streams (the finish of a batch beside
 * the accumulate of the next: they keep separate scratch); two launches of the same kind may not.
 */
#define EPIK_AMD_MAX_SHARDS 16
typedef struct {
    uint32_t lists;        /* 1: accumulate_lists / finish_lists are available on this handle */
    uint32_t slices;       /* lists per read */
    uint32_t slice_rows;   /* branches per slice */
    uint32_t entry_bytes;  /* 8 or 16, for the count width chosen last */
    uint32_t num_branches;
    uint32_t reserved;
    double postings_per_kmer; /* mean postings of this shard's lists per k-mer code (to size d_entries:
                                 reads * k-mers per read * this, and some margin) */
} epik_amd_partial_info;
int epik_amd_placer_partial_info(const epik_amd_placer *p, epik_amd_partial_info *out);
int epik_amd_placer_accumulate_lists_device(epik_amd_placer *p, const void *d_seqs, const void *d_seq_offsets,
                                            uint64_t n, uint32_t n_parts, void *d_entries, uint64_t entries_cap,
                                            void *d_index, void *d_part_entries, const void *d_amb_slot,
                                            void *d_amb_order, void *d_amb_avg, void *stream);
int epik_amd_placer_finish_lists_device(epik_amd_placer *p, const void *d_seq_offsets, uint64_t n, uint32_t n_shards,
                                        const void *const *d_entries, const void *const *d_index,
                                        const void *d_amb_slot, const void *d_amb_avg, void *d_rows, void *d_n_rows,
                                        void *d_kmer_counts, void *stream);

/*
 * The whole of it for a caller that drives all the devices from one process (epik-dna --db-shard): places n
 * HOST reads on `n_shards` handles that together hold one database -- handle g created with shard g of n_shards
 * (create() on a descriptor of its codes, or create_sharded()), on any devices, several on one if need be.
 * Replaces the OpenMP loop of epik::placer::place (place.cpp:218-268) as epik_amd_placer_place does, same
 * arguments and results.  Inside: chunks of the batch; every handle accumulates the partial lists of a chunk on
 * its device, part r of each goes to the device of handle r with one peer copy per pair (its own xGMI link),
 * handle r finishes its reads; the copies of a chunk run under the kernels of the next.  Reads with ambiguous
 * characters follow the first-key rule over all shards.  Large trees only (partial lists): a database that needs
 * several devices has one.  Synchronous; the handles must not be used by another thread meanwhile.
 */
int epik_amd_placer_place_sharded(epik_amd_placer *const *shards, uint32_t n_shards, const char *seqs,
                                  const uint64_t *seq_offsets, uint64_t n, epik_amd_placement *rows,
                                  uint32_t *n_rows, uint32_t *kmer_counts);

/* Which kernels the last launch of this handle ran (reports; a large-tree handle falls back from the three-kernel
 * placement to the one-kernel one when the device has no room for the scratch of a launch). */
#define EPIK_AMD_PATH_WAVE 0u            /* place_reads_kernel: one wavefront per read */
#define EPIK_AMD_PATH_TEAM_ONE_KERNEL 1u /* team_place_kernel: one workgroup per read */
#define EPIK_AMD_PATH_TEAM_STREAMED 2u   /* team_front_kernel + team_stream_kernel + team_merge_kernel (+ the other for the rest) */
int epik_amd_placer_last_path(const epik_amd_placer *p, uint32_t *path);
/* What the streaming kernel of a large-tree placer is, with the handle's current count width (diagnostics, tests):
 * *wide = 1: the build for slices so large that LDS keeps a CU to twelve waves, which holds the slice epilogue over
 * the touched quads; *sparse_quads = how many touched quads (4 rows each) an item may have to take that epilogue
 * (0: never).  Both 0 for a placer of the one-wavefront kernel. */
int epik_amd_placer_stream_build(const epik_amd_placer *p, uint32_t *wide, uint32_t *sparse_quads);

/* Gives back what the handle's launches have grown and kept: the scratch of large-tree launches (descriptor pool,
 * headers, slice results: up to ~1 GB after batches of a million reads), the staging buffers of the host entry
 * points, the buffers of place_sharded.  The database stays; the next launch allocates again what it needs.
 * Synchronises the handle's device. */
int epik_amd_placer_release_scratch(epik_amd_placer *p);

/* Launch geometry actually used (for reports): waves per workgroup (one read per wave with the
 * one-wavefront kernel, one slice of a read per wave on large trees), workgroups of the last launch,
 * dynamic LDS bytes per workgroup. */
int epik_amd_placer_launch_info(const epik_amd_placer *p, uint32_t *waves_per_block,
                                uint32_t *blocks, uint32_t *lds_bytes);

/* Times the last place_device launch on its own stream with HIP events recorded
 * around the kernel inside the library (milliseconds; <0 if none was recorded).
 * Event recording is enabled with epik_amd_placer_set_timing(p, 1). */
int epik_amd_placer_set_timing(epik_amd_placer *p, int enabled);
int epik_amd_placer_last_kernel_ms(epik_amd_placer *p, float *ms_out);

#ifdef __cplusplus
}
#endif
#endif
