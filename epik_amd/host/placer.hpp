// placer.hpp -- host-side mirror of `epik::placer` over the C ABI of libepik_amd.so.
//
// Same names, arguments and error behaviour as the reference class
// (epik/include/epik/place.h:39-140): construct with (db, tree, keep_at_most, keep_factor,
// max_threads), call place(seq_records, num_threads) per FASTA batch, get a
// placed_collection whose string_views point into the caller's batch.
// Extra, MI355X-specific: a list of HIP devices (database replicated on each, no collective) and
// place_batches(): several FASTA batches in one launch on one of them -- the driver hands whole
// groups of batches to the devices in turn.
#ifndef EPIK_AMD_HOST_PLACER_HPP
#define EPIK_AMD_HOST_PLACER_HPP

#include <functional>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "epik_amd.h"
#include "phylo_kmer_db.hpp"
#include "phylo_tree.hpp"
#include "seq_record.hpp"

namespace epik_amd::impl {

/// "sequence content -> list of headers" (place.h:42)
using sequence_map_t = std::unordered_map<std::string_view, std::vector<std::string_view>>;

/// A placement of one sequence (place.h:45-56)
struct placement {
    using weight_ratio_type = double;
    uint32_t branch_id;
    float score;
    weight_ratio_type weight_ratio;
    size_t count;
    phylo_node::branch_length_type distal_length;
    phylo_node::branch_length_type pendant_length;
};

/// place.h:59-68
struct placed_sequence {
    std::string_view sequence;
    std::vector<placement> placements;
};

/// place.h:72-75.  placed_seqs follow the first occurrence of each sequence in the batch
/// (the reference: std::unordered_map iteration order, place.cpp:57-61).
struct placed_collection {
    sequence_map_t sequence_map;
    std::vector<placed_sequence> placed_seqs;
};

/// The driver's own form of a placed batch: what a placed_collection holds, in five flat arrays (a
/// placed_collection is a hash-map node, a vector of headers and a vector of placements PER READ -- three
/// allocations each, made by the placing thread and freed by the writing one; at a million reads per second and
/// device that is what the driver would spend its time on).  Unique sequences in first-occurrence order.
struct placed_batch {
    std::vector<std::string_view> sequences;  // [n_unique]
    std::vector<uint32_t> row_begin;          // [n_unique + 1]: the placements of sequence u are rows[row_begin[u] .. row_begin[u + 1])
    std::vector<placement> rows;
    std::vector<uint32_t> name_begin;         // [n_unique + 1]: ... its headers names[name_begin[u] .. name_begin[u + 1]), input order
    std::vector<std::string_view> names;
    size_t size() const noexcept { return sequences.size(); }
};

}  // namespace epik_amd::impl

namespace epik_amd {

class placer {
public:
    using placed_collection = impl::placed_collection;

    /// Shard g of a k-mer-space-sharded database (--db-shard), loaded when its turn comes and dropped as soon as
    /// its lists are on the device: the process never holds more than one shard on the host.
    using shard_loader = std::function<phylo_kmer_db(uint32_t shard_index)>;

    /// WARNING (as place.h:91-93): db and tree are kept by reference.
    /// db_shards == 1: the database replicated on every device of `devices`, whole groups of batches go to them in
    /// turn (no collective).  db_shards == G > 1: `db` holds shard 0 of G (phylo_kmer_db::shard_count() says so),
    /// load_shard(g) the others; handle g is created on devices[g % devices.size()] and every batch is placed by
    /// all of them together (epik_amd_placer_place_sharded) -- a database larger than one device's memory.
    placer(const phylo_kmer_db& db, const phylo_tree& original_tree, size_t keep_at_most, double keep_factor,
           size_t max_threads, std::vector<int> devices = {0}, uint32_t db_shards = 1,
           const shard_loader& load_shard = {});
    placer(const placer&) = delete;
    placer& operator=(const placer&) = delete;
    ~placer() noexcept;

    /// The reference's call: one batch, on the first device.
    placed_collection place(const std::vector<seq_record>& seq_records, size_t num_threads);

    /// Several batches in ONE launch on device `device_index` (of the list given to the constructor).
    /// Every batch is de-duplicated on its own, exactly as `place` does it (place.cpp:207-212: dedup is
    /// per batch), the unique reads of all of them cross the boundary together, and every batch gets
    /// its own placed_collection back.  Thread-safe across different devices.
    /// The host work on either side of the launch -- dedup, joining the reads, the placements with their branch
    /// lengths -- is per batch: `num_threads` threads share the batches.
    std::vector<placed_collection> place_batches(const std::vector<const std::vector<seq_record>*>& batches,
                                                 size_t device_index, size_t num_threads = 1);
    /// The same placement in the driver's flat form (impl::placed_batch): what epik-dna / epik-aa call.
    std::vector<impl::placed_batch> place_flat(const std::vector<const std::vector<seq_record>*>& batches,
                                               size_t device_index, size_t num_threads = 1);

    /// How many callers may place at the same time (place_batches' device_index): the devices of a replicated
    /// database, ONE for a sharded one (all its handles work on every batch).
    size_t device_count() const noexcept { return _sharded ? 1 : _handles.size(); }
    size_t handle_count() const noexcept { return _handles.size(); }
    /// distal_length / pendant_length of a placement on branch b (place.cpp:110-123, 435-437)
    std::vector<double> distal_lengths() const;
    const std::vector<double>& pendant_lengths() const noexcept { return _pendant_lengths; }

private:
    const phylo_kmer_db& _db;
    const phylo_tree& _original_tree;
    const float _threshold;
    const float _log_threshold;
    const size_t _keep_at_most;
    const double _keep_factor;
    std::vector<double> _pendant_lengths;
    std::vector<epik_amd_placer*> _handles;  // one per device (replicated) or per shard (sharded)
    bool _sharded = false;
};

}  // namespace epik_amd
#endif
