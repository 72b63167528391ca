#include "placer.hpp"

#include <cmath>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>

#include "parallel.hpp"

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
        if (_sharded && g == 0) {
            // A tree of the one-wavefront kernels leaves dense partial vectors, which place_sharded does not take:
            // said now, before the other shards are loaded and uploaded, not at the first batch.
            epik_amd_partial_info info{};
            if (epik_amd_placer_partial_info(handle, &info) == EPIK_AMD_OK && !info.lists) {
                for (auto* h : _handles) epik_amd_placer_destroy(h);
                _handles.clear();
                throw std::runtime_error("GPU placer: --db-shard needs the kernels of a large tree (this one has " +
                                         std::to_string(desc.num_branches) +
                                         " branches and fits one wavefront per read): replicate the database with --devices / --gpus instead");
            }
        }
    }
}

std::vector<double> placer::distal_lengths() const
{
    std::vector<double> out(_pendant_lengths.size(), 0.0);
    for (uint32_t i = 0; i < out.size(); ++i)
        if (const auto node = _original_tree.get_by_postorder_id(i)) out[i] = (*node)->get_branch_length() / 2;  // place.cpp:435
    return out;
}

placer::~placer() noexcept
{
    for (auto* h : _handles) epik_amd_placer_destroy(h);
}

placed_collection placer::place(const std::vector<seq_record>& seq_records, size_t /*num_threads*/)
{
    auto placed = place_batches({&seq_records}, 0, 1);
    return std::move(placed[0]);
}

namespace {

// 64-bit hash of a sequence, eight bytes at a time (dedup of a batch: place.cpp:73-81 groups by content)
inline uint64_t hash_bytes(std::string_view s)
{
    uint64_t h = 0x9e3779b97f4a7c15ull ^ (uint64_t)s.size();
    const char* p = s.data();
    size_t n = s.size();
    while (n >= 8) {
        uint64_t w;
        std::memcpy(&w, p, 8);
        h = (h ^ w) * 0xff51afd7ed558ccdull;
        h ^= h >> 32;
        p += 8, n -= 8;
    }
    uint64_t w = 0;
    std::memcpy(&w, p, n);
    h = (h ^ w) * 0xc4ceb9fe1a85ec53ull;
    return h ^ (h >> 29);
}

}  // namespace

std::vector<impl::placed_batch> placer::place_flat(const std::vector<const std::vector<seq_record>*>& batches,
                                                   size_t device_index, size_t num_threads)
{
    if (device_index >= device_count()) throw std::runtime_error("GPU placer: no such device index");
    std::vector<impl::placed_batch> out(batches.size());
    // identical sequences of a batch are placed once (place.cpp:73-81, 207-212); the unique reads of
    // all batches go through the boundary in one call.  The batches are independent of each other up to
    // that call and after it: `num_threads` threads take them one at a time.
    std::vector<size_t> first_unique(batches.size() + 1, 0), first_byte(batches.size() + 1, 0);
    parallel_for(batches.size(), num_threads, [&](size_t b) {
        const auto& batch = *batches[b];
        impl::placed_batch pb;  // (the thread's own while it grows: neighbours in out[] share cache lines)
        if (batch.size() >= 0xffffffffull) throw std::runtime_error("GPU placer: a batch of 2^32 reads or more");
        // open addressing over the positions of the batch: slot -> index of a unique sequence + 1
        size_t cap = 16;
        while (cap < 2 * batch.size()) cap *= 2;
        std::vector<uint32_t> table(cap, 0), unique_of(batch.size());
        pb.sequences.reserve(batch.size());
        std::vector<uint32_t> n_names;
        n_names.reserve(batch.size());
        size_t bytes = 0;
        for (size_t i = 0; i < batch.size(); ++i) {
            const std::string_view seq = batch[i].sequence();
            size_t slot = (size_t)hash_bytes(seq) & (cap - 1);
            for (;;) {
                const uint32_t u = table[slot];
                if (u == 0) {
                    table[slot] = (uint32_t)pb.sequences.size() + 1;
                    unique_of[i] = (uint32_t)pb.sequences.size();
                    pb.sequences.push_back(seq);
                    n_names.push_back(1);
                    bytes += seq.size();
                    break;
                }
                if (pb.sequences[u - 1] == seq) {
                    unique_of[i] = u - 1;
                    ++n_names[u - 1];
                    break;
                }
                slot = (slot + 1) & (cap - 1);
            }
        }
        // the headers of every unique sequence, in input order (jplace "nm", jplace.cpp:141-158)
        const size_t n_unique = pb.sequences.size();
        pb.name_begin.assign(n_unique + 1, 0);
        for (size_t u = 0; u < n_unique; ++u) pb.name_begin[u + 1] = pb.name_begin[u] + n_names[u];
        pb.names.resize(batch.size());
        std::vector<uint32_t> at(pb.name_begin.begin(), pb.name_begin.end() - 1);
        for (size_t i = 0; i < batch.size(); ++i) pb.names[at[unique_of[i]]++] = batch[i].header();
        first_unique[b + 1] = n_unique;
        first_byte[b + 1] = bytes;
        out[b] = std::move(pb);
    });
    for (size_t b = 0; b < batches.size(); ++b) {
        first_unique[b + 1] += first_unique[b];
        first_byte[b + 1] += first_byte[b];
    }
    const size_t n = first_unique.back();
    if (n == 0) {
        for (auto& pb : out) pb.row_begin.assign(1, 0);
        return out;
    }
    std::unique_ptr<char[]> bytes(new char[first_byte.back() + 1]);
    std::unique_ptr<uint64_t[]> offsets(new uint64_t[n + 1]);
    offsets[n] = first_byte.back();
    parallel_for(batches.size(), num_threads, [&](size_t b) {
        size_t at = first_byte[b], i = first_unique[b];
        for (const auto seq : out[b].sequences) {
            offsets[i++] = at;
            std::memcpy(bytes.get() + at, seq.data(), seq.size());
            at += seq.size();
        }
    });
    std::unique_ptr<epik_amd_placement[]> rows(new epik_amd_placement[n * _keep_at_most]);
    std::unique_ptr<uint32_t[]> n_rows(new uint32_t[n]), counts(new uint32_t[n * _keep_at_most]);
    const int rc = _sharded ? epik_amd_placer_place_sharded(_handles.data(), (uint32_t)_handles.size(), bytes.get(),
                                                            offsets.get(), n, rows.get(), n_rows.get(), counts.get())
                            : epik_amd_placer_place(_handles[device_index], bytes.get(), offsets.get(), n, rows.get(),
                                                    n_rows.get(), counts.get());
    if (rc != EPIK_AMD_OK) throw std::runtime_error(std::string("GPU placer: ") + epik_amd_last_error());
    parallel_for(batches.size(), num_threads, [&](size_t b) {
        auto& pb = out[b];
        const size_t n_unique = pb.sequences.size();
        pb.row_begin.assign(n_unique + 1, 0);
        for (size_t u = 0; u < n_unique; ++u) {
            const size_t i = first_unique[b] + u;
            // (n_rows is a row count here: epik_amd_placer_place widens the counts by itself, so the
            // EPIK_AMD_ROWS_COUNTS_TOO_NARROW mark of the device entry points must never arrive)
            if (n_rows[i] > _keep_at_most)
                throw std::runtime_error("GPU placer: read " + std::to_string(i) + " came back with " +
                                         std::to_string(n_rows[i]) + " rows (keep_at_most " +
                                         std::to_string(_keep_at_most) + ")");
            pb.row_begin[u + 1] = pb.row_begin[u] + n_rows[i];
        }
        pb.rows.resize(pb.row_begin[n_unique]);
        for (size_t u = 0; u < n_unique; ++u) {
            const size_t i = first_unique[b] + u;
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
                pb.rows[pb.row_begin[u] + r] = {row.branch, row.score, row.lwr, count, distal, pendant};
            }
        }
    });
    return out;
}

std::vector<placed_collection> placer::place_batches(const std::vector<const std::vector<seq_record>*>& batches,
                                                     size_t device_index, size_t num_threads)
{
    // the reference's containers (place.h:59-75), filled from the flat form
    auto flat = place_flat(batches, device_index, num_threads);
    std::vector<placed_collection> out(batches.size());
    parallel_for(batches.size(), num_threads, [&](size_t b) {
        const auto& pb = flat[b];
        out[b].sequence_map.reserve(pb.size());
        out[b].placed_seqs.reserve(pb.size());
        for (size_t u = 0; u < pb.size(); ++u) {
            out[b].sequence_map.emplace(pb.sequences[u], std::vector<std::string_view>(pb.names.begin() + pb.name_begin[u],
                                                                                       pb.names.begin() + pb.name_begin[u + 1]));
            out[b].placed_seqs.push_back({pb.sequences[u], std::vector<placement>(pb.rows.begin() + pb.row_begin[u],
                                                                                  pb.rows.begin() + pb.row_begin[u + 1])});
        }
    });
    return out;
}

}  // namespace epik_amd
