// phylo_kmer_db.hpp -- the phylo-k-mer database as the placer's host side sees it.
//
// Mirrors the i2l surface EPIK uses (absent submodule; contract from the call sites):
//   i2l::load(file, mu, omega, max_entries)                       main.cpp:277
//   db.kmer_size() omega() tree() tree_index() sequence_type() version()
//   positions_loaded() get_num_entries_loaded() get_num_entries_total()   main.cpp:278-294, place.cpp:87,113
//   i2l::score_threshold(omega, k)                                 place.cpp:87
//   i2l::pkdb_value {branch, score}                                main.cpp:257
//
// ON-DISK FORMAT.  EPIK reads `.ipk` files (Boost.Serialization + zlib inside i2l).
// Neither i2l nor a sample file is available here, so that format cannot be restated or
// validated; load() recognises it only to refuse it with a clear message.  What load()
// reads is this repository's own flat container "EPIKAMD1" (written by
// epik_amd/dbfile.py), which keeps the semantics the call sites need:
//   * k-mer records are stored most informative first, so that a prefix is the best
//     `mu` fraction (README.md:126) and `--max-ram` cuts the same order (main.cpp:252-257);
//   * postings below the score threshold of the user's omega are dropped (README.md:125).
#ifndef EPIK_AMD_HOST_PHYLO_KMER_DB_HPP
#define EPIK_AMD_HOST_PHYLO_KMER_DB_HPP

#include <cstddef>
#include <cstdint>
#include <limits>
#include <string>
#include <utility>
#include <vector>

#include "epik_amd.h"
#include "phylo_tree.hpp"

namespace epik_amd {

using pkdb_value = epik_amd_pkdb_value;  // {uint32_t branch; float score;}  (main.cpp:257)

namespace protocol {
constexpr unsigned int EARLIEST_INDEX = 1;  // oldest container version this build accepts (main.cpp:278-283)
constexpr unsigned int CURRENT = 1;
}

/// (omega / sigma)^k as float (call site place.cpp:87).  ASSUMPTION: evaluated in double, rounded once.
float score_threshold(float omega, size_t kmer_size, unsigned int alphabet_size);

unsigned int alphabet_size(const std::string& sequence_type);  // "DNA" -> 4, "Proteins" -> 20

/// 256-entry class table of the k-mer encoder (bit s set <=> the character may be state s).
/// ASSUMPTION register of SURVEY.md 8c: nucl A0 C1 G2 T3 (U = T), IUPAC codes ambiguous;
/// amino R H K D E S T N Q C G P A I L M F W Y V, B/Z/J/X ambiguous; everything else invalid.
std::vector<uint32_t> char_class_table(const std::string& sequence_type);

class phylo_kmer_db {
public:
    size_t kmer_size() const noexcept { return _kmer_size; }
    float omega() const noexcept { return _omega; }
    const std::string& tree() const noexcept { return _tree; }
    const std::string& sequence_type() const noexcept { return _sequence_type; }
    unsigned int version() const noexcept { return _version; }
    bool positions_loaded() const noexcept { return false; }
    size_t get_num_entries_loaded() const noexcept { return _values.size(); }
    size_t get_num_entries_total() const noexcept { return _num_entries_total; }
    const std::vector<tree_index_entry>& tree_index() const noexcept { return _tree_index; }

    // The lists as handed to the C ABI, in its sparse form (memory per PRESENT k-mer, as the reference's hash
    // map): keys() = the codes that have a list, ascending; the list of keys()[i] =
    // values[offsets()[i] .. offsets()[i + 1]).
    uint64_t num_keys() const noexcept { return _num_keys; }  // alphabet_size ^ k: the size of the key space
    const std::vector<uint32_t>& keys() const noexcept { return _keys; }
    const std::vector<uint64_t>& offsets() const noexcept { return _offsets; }
    const std::vector<pkdb_value>& values() const noexcept { return _values; }
    /// phylo_kmer_db::search (place.cpp:300): the list of a k-mer code, {nullptr, 0} when it has none
    std::pair<const pkdb_value*, size_t> search(uint32_t key) const noexcept;
    // which part of the database this object holds (k-mer-space shard: the codes with code % count == index)
    uint32_t shard_index() const noexcept { return _shard_index; }
    uint32_t shard_count() const noexcept { return _shard_count; }
    /// Frees the lists (the placer has them on the device: place.cpp:300 is answered there); what the jplace
    /// output needs -- tree, k, omega -- stays.
    void drop_lists() noexcept
    {
        std::vector<uint32_t>().swap(_keys);
        std::vector<uint64_t>().swap(_offsets);
        std::vector<pkdb_value>().swap(_values);
    }

    size_t _kmer_size = 0;
    float _omega = 0.0f;
    std::string _tree;
    std::string _sequence_type;
    unsigned int _version = 0;
    size_t _num_entries_total = 0;
    std::vector<tree_index_entry> _tree_index;
    uint64_t _num_keys = 0;
    uint32_t _shard_index = 0, _shard_count = 1;
    std::vector<uint32_t> _keys;
    std::vector<uint64_t> _offsets;
    std::vector<pkdb_value> _values;
};

/// Loads at most `max_entries` postings of the best `mu` fraction of k-mers, dropping
/// postings under the threshold of `omega`.  Throws std::runtime_error (caught main.cpp:384).
/// shard_index / shard_count (--db-shard): only the k-mers with code % shard_count == shard_index are kept --
/// a process (or device thread) of a k-mer-space-sharded run never holds the others; `mu` and `max_entries`
/// cut the file's order as before, `max_entries` counting what THIS shard keeps.
/// The file is mapped and walked twice (count, then place): every kept posting is written once, straight to
/// its final place; nothing but the postings, 12 bytes per present k-mer and a transient 24-byte record
/// per k-mer of the shard is allocated.
phylo_kmer_db load(const std::string& filename, float mu = 1.0f, float omega = 1.5f,
                   size_t max_entries = std::numeric_limits<size_t>::max(), uint32_t shard_index = 0,
                   uint32_t shard_count = 1);

}  // namespace epik_amd
#endif
