#include "placer.hpp"

#include <cmath>
#include <stdexcept>
#include <string>

namespace epik_amd {

using impl::placed_collection;
using impl::placed_sequence;
using impl::placement;
using impl::sequence_map_t;

placer::placer(const phylo_kmer_db& db, const phylo_tree& original_tree, size_t keep_at_most, double keep_factor,
               size_t /*max_threads*/, std::vector<int> devices, uint32_t db_shards, const shard_loader& load_shard)
    : _db{db}
    , _original_tree{original_tree}
    , _threshold{score_threshold(db.omega(), db.kmer_size(), alphabet_size(db.sequence_type()))}  // place.cpp:87
    , _log_threshold{std::log10(_threshold)}                                                      // place.cpp:88
    , _keep_at_most{keep_at_most}
    , _keep_factor{keep_factor}
{
    // pendant lengths (place.cpp:99-125)
    const auto& index = _db.tree_index();
    for (uint32_t i = 0; i < original_tree.get_node_count(); ++i) {
        const auto node = _original_tree.get_by_postorder_id(i);
        if (!node || i >= index.size())
            throw std::runtime_error("Could not find node by post-order id: " + std::to_string(i));
        const auto distal_length = (*node)->get_branch_length() / 2;
        auto mean_subtree_branch_length = 0.0;
        if (index[i].subtree_num_nodes > 1)
            mean_subtree_branch_length = index[i].subtree_total_length / (double)index[i].subtree_num_nodes;
        _pendant_lengths.push_back(mean_subtree_branch_length + distal_length);
    }

    const auto char_class = char_class_table(db.sequence_type());
    if (devices.empty()) devices.push_back(0);
    _sharded = db_shards > 1;
    if (_sharded && (db.shard_count() != db_shards || db.shard_index() != 0 || !load_shard))
        throw std::runtime_error("GPU placer: a sharded placer takes shard 0 of the database and a loader for the others");
    // one handle per device (the database replicated) or per shard (shard g on devices[g % devices.size()])
    const size_t n_handles = _sharded ? db_shards : devices.size();
    for (size_t g = 0; g < n_handles; ++g) {
        phylo_kmer_db loaded;  // shard g > 0: here until its lists are on the device
        if (_sharded && g > 0) {
            loaded = load_shard((uint32_t)g);
            if (loaded.shard_index() != g || loaded.shard_count() != db_shards || loaded.kmer_size() != db.kmer_size() ||
                loaded.num_keys() != db.num_keys() || loaded.omega() != db.omega())
                throw std::runtime_error("GPU placer: shard " + std::to_string(g) + " does not belong to this database");
        }
        const phylo_kmer_db& part = (_sharded && g > 0) ? loaded : db;
        epik_amd_placer_desc desc{};
        desc.abi_version = EPIK_AMD_ABI_VERSION;
        desc.kmer_size = (uint32_t)db.kmer_size();
        desc.alphabet_size = alphabet_size(db.sequence_type());
        desc.num_branches = (uint32_t)original_tree.get_node_count();
        desc.keep_at_most = (uint32_t)keep_at_most;
        desc.offset_bits = 64;
        desc.keep_factor = keep_factor;
        desc.threshold = _threshold;
        desc.log_threshold = _log_threshold;
        desc.num_keys = db.num_keys();
        desc.num_entries = part.values().size();
        desc.offsets = part.offsets().data();  // the sparse form: memory per present k-mer (ABI 3)
        desc.keys = part.keys().data();
        desc.num_present = part.keys().size();
        desc.values = part.values().data();
        desc.char_class = char_class.data();
        desc.device = devices[g % devices.size()];
        static const uint32_t no_key = 0;  // (an empty shard: keys must still be non-null to say "sparse form")
        if (desc.num_present == 0) desc.keys = &no_key;
        epik_amd_placer* handle = nullptr;
        const int rc = _sharded ? epik_amd_placer_create_sharded(&desc, (uint32_t)g, db_shards, &handle)
                                : epik_amd_placer_create(&desc, &handle);
        if (rc != EPIK_AMD_OK) {
            const std::string message = epik_amd_last_error();
            for (auto* h : _handles) epik_amd_placer_destroy(h);
            throw std::runtime_error("GPU placer: " + message);
        }
        _handles.push_back(handle);
    }
}

placer::~placer() noexcept
{
    for (auto* h : _handles) epik_amd_placer_destroy(h);
}

placed_collection placer::place(const std::vector<seq_record>& seq_records, size_t /*num_threads*/)
{
    auto placed = place_batches({&seq_records}, 0);
    return std::move(placed[0]);
}

std::vector<placed_collection> placer::place_batches(const std::vector<const std::vector<seq_record>*>& batches,
                                                     size_t device_index)
{
    if (device_index >= device_count()) throw std::runtime_error("GPU placer: no such device index");
    std::vector<placed_collection> out(batches.size());
    // identical sequences of a batch are placed once (place.cpp:73-81, 207-212); the unique reads of
    // all batches go through the boundary in one call
    std::string bytes;
    std::vector<uint64_t> offsets{0};
    std::vector<size_t> first_unique(batches.size() + 1, 0);
    for (size_t b = 0; b < batches.size(); ++b) {
        auto& sequence_map = out[b].sequence_map;
        auto& placed_seqs = out[b].placed_seqs;
        for (const auto& rec : *batches[b]) {
            auto [it, inserted] = sequence_map.try_emplace(rec.sequence());
            if (inserted) {
                placed_seqs.push_back({rec.sequence(), {}});
                bytes.append(rec.sequence());
                offsets.push_back(bytes.size());
            }
            it->second.push_back(rec.header());
        }
        first_unique[b + 1] = first_unique[b] + placed_seqs.size();
    }
    const size_t n = offsets.size() - 1;
    if (n == 0) return out;
    std::vector<epik_amd_placement> rows(n * _keep_at_most);
    std::vector<uint32_t> n_rows(n), counts(n * _keep_at_most);
    const int rc = _sharded ? epik_amd_placer_place_sharded(_handles.data(), (uint32_t)_handles.size(), bytes.data(),
                                                            offsets.data(), n, rows.data(), n_rows.data(), counts.data())
                            : epik_amd_placer_place(_handles[device_index], bytes.data(), offsets.data(), n, rows.data(),
                                                    n_rows.data(), counts.data());
    if (rc != EPIK_AMD_OK) throw std::runtime_error(std::string("GPU placer: ") + epik_amd_last_error());
    for (size_t b = 0; b < batches.size(); ++b) {
        for (size_t u = 0; u < out[b].placed_seqs.size(); ++u) {
            const size_t i = first_unique[b] + u;
            auto& placements = out[b].placed_seqs[u].placements;
            // (n_rows is a row count here: epik_amd_placer_place widens the counts by itself, so the
            // EPIK_AMD_ROWS_COUNTS_TOO_NARROW mark of the device entry points must never arrive)
            if (n_rows[i] > _keep_at_most)
                throw std::runtime_error("GPU placer: read " + std::to_string(i) + " came back with " +
                                         std::to_string(n_rows[i]) + " rows (keep_at_most " +
                                         std::to_string(_keep_at_most) + ")");
            placements.reserve(n_rows[i]);
            for (uint32_t r = 0; r < n_rows[i]; ++r) {
                const auto& row = rows[i * _keep_at_most + r];
                const size_t count = counts[i * _keep_at_most + r];
                // rows fabricated for a read without hits carry 0.0 lengths (place.cpp:150)
                double distal = 0.0, pendant = 0.0;
                if (count != 0) {
                    const auto node = _original_tree.get_by_postorder_id(row.branch);
                    if (!node)
                        throw std::runtime_error("Could not find node by post-order id: " + std::to_string(row.branch));
                    distal = (*node)->get_branch_length() / 2;  // place.cpp:435
                    pendant = _pendant_lengths[row.branch];
                }
                placements.push_back({row.branch, row.score, row.lwr, count, distal, pendant});
            }
        }
    }
    return out;
}

}  // namespace epik_amd
