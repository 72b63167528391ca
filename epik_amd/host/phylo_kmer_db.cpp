#include "phylo_kmer_db.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace epik_amd {

float score_threshold(float omega, size_t kmer_size, unsigned int sigma)
{
    return static_cast<float>(std::pow(static_cast<double>(omega) / static_cast<double>(sigma),
                                       static_cast<double>(kmer_size)));
}

unsigned int alphabet_size(const std::string& sequence_type)
{
    if (sequence_type == "DNA") return 4;
    if (sequence_type == "Proteins") return 20;
    throw std::runtime_error("Unknown sequence type: " + sequence_type);
}

std::vector<uint32_t> char_class_table(const std::string& sequence_type)
{
    std::vector<uint32_t> table(256, 0u);
    auto set = [&](char c, uint32_t mask) {
        table[(unsigned char)c] = mask;
        if (c >= 'A' && c <= 'Z') table[(unsigned char)(c - 'A' + 'a')] = mask;
    };
    if (sequence_type == "DNA") {
        const char* states = "ACGT";
        for (int i = 0; i < 4; ++i) set(states[i], 1u << i);
        set('U', 1u << 3);
        const struct { char c; const char* m; } amb[] = {
            {'R', "AG"}, {'Y', "CT"}, {'S', "CG"}, {'W', "AT"}, {'K', "GT"}, {'M', "AC"},
            {'B', "CGT"}, {'D', "AGT"}, {'H', "ACT"}, {'V', "ACG"}, {'N', "ACGT"}};
        for (const auto& a : amb) {
            uint32_t mask = 0;
            for (const char* m = a.m; *m; ++m) mask |= table[(unsigned char)*m];
            set(a.c, mask);
        }
    } else if (sequence_type == "Proteins") {
        const char* states = "RHKDESTNQCGPAILMFWYV";
        for (int i = 0; i < 20; ++i) set(states[i], 1u << i);
        set('B', table[(unsigned char)'D'] | table[(unsigned char)'N']);
        set('Z', table[(unsigned char)'E'] | table[(unsigned char)'Q']);
        set('J', table[(unsigned char)'I'] | table[(unsigned char)'L']);
        set('X', (1u << 20) - 1u);
    } else {
        throw std::runtime_error("Unknown sequence type: " + sequence_type);
    }
    return table;
}

namespace {

template <typename T>
T read_pod(std::istream& in)
{
    T v{};
    in.read(reinterpret_cast<char*>(&v), sizeof(T));
    if (!in) throw std::runtime_error("Unexpected end of the database file");
    return v;
}

struct kmer_record {
    uint32_t key;
    uint64_t first;  // into the temporary posting array
    uint32_t n;
};

}  // namespace

phylo_kmer_db load(const std::string& filename, float mu, float omega, size_t max_entries)
{
    std::ifstream in(filename, std::ios::binary);
    if (!in) throw std::runtime_error("Cannot open the database: " + filename);
    char magic[8];
    in.read(magic, 8);
    if (!in) throw std::runtime_error("The database file is too short: " + filename);
    if (std::memcmp(magic, "EPIKAMD1", 8) != 0) {
        // Boost archives start with "22 serialization::archive"; zlib streams with 0x78
        throw std::runtime_error(
            "Unsupported database container: " + filename +
            " is not an EPIKAMD1 file.  IPK's .ipk files (Boost.Serialization inside i2l) cannot be "
            "read by this build; convert the database with epik_amd/dbfile.py");
    }
    phylo_kmer_db db;
    db._version = read_pod<uint32_t>(in);
    const uint32_t seq_type = read_pod<uint32_t>(in);
    db._sequence_type = seq_type == 0 ? "DNA" : "Proteins";
    db._kmer_size = read_pod<uint32_t>(in);
    const float built_omega = read_pod<float>(in);
    const uint64_t num_kmers = read_pod<uint64_t>(in);
    db._num_entries_total = (size_t)read_pod<uint64_t>(in);
    const uint64_t newick_len = read_pod<uint64_t>(in);
    db._tree.resize(newick_len);
    in.read(db._tree.data(), (std::streamsize)newick_len);
    if (!in) throw std::runtime_error("Unexpected end of the database file (tree)");

    const unsigned int sigma = alphabet_size(db._sequence_type);
    // The user's omega replaces the stored one when it is stricter (README.md:125)
    db._omega = std::max(omega, built_omega);
    const float log_thr = std::log10(score_threshold(db._omega, db._kmer_size, sigma));

    uint64_t num_keys = 1;
    for (size_t i = 0; i < db._kmer_size; ++i) {
        num_keys *= sigma;
        if (num_keys > 0xffffffffull) throw std::runtime_error("alphabet_size^k exceeds 2^32 k-mer codes");
    }
    const uint64_t kmers_to_load = (uint64_t)std::ceil((double)mu * (double)num_kmers);
    std::vector<kmer_record> records;
    std::vector<pkdb_value> tmp;
    std::vector<pkdb_value> buf;
    for (uint64_t r = 0; r < num_kmers && r < kmers_to_load; ++r) {
        const uint32_t key = read_pod<uint32_t>(in);
        const uint32_t n = read_pod<uint32_t>(in);
        if (key >= num_keys) throw std::runtime_error("k-mer code out of range in the database");
        buf.resize(n);
        in.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)(n * sizeof(pkdb_value)));
        if (!in) throw std::runtime_error("Unexpected end of the database file (postings)");
        kmer_record rec{key, tmp.size(), 0};
        for (const auto& v : buf)
            if (v.score >= log_thr) {
                tmp.push_back(v);
                ++rec.n;
            }
        if (tmp.size() > max_entries) {  // --max-ram: stop in front of the k-mer that does not fit
            tmp.resize(rec.first);
            break;
        }
        if (rec.n) records.push_back(rec);
    }
    // CSR by k-mer code
    db._offsets.assign(num_keys + 1, 0);
    for (const auto& rec : records) {
        if (db._offsets[rec.key + 1] != 0) throw std::runtime_error("duplicate k-mer in the database");
        db._offsets[rec.key + 1] = rec.n;
    }
    for (uint64_t i = 0; i < num_keys; ++i) db._offsets[i + 1] += db._offsets[i];
    db._values.resize(tmp.size());
    for (const auto& rec : records)
        std::copy(tmp.begin() + (std::ptrdiff_t)rec.first, tmp.begin() + (std::ptrdiff_t)(rec.first + rec.n),
                  db._values.begin() + (std::ptrdiff_t)db._offsets[rec.key]);
    db._tree_index = io::parse_newick(db._tree).tree_index();
    return db;
}

}  // namespace epik_amd
