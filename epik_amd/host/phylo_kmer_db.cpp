#include "phylo_kmer_db.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <limits>
#include <cstddef>
#include <stdexcept>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

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

// a read-only mapping of a whole file
class mapped_file {
public:
    explicit mapped_file(const std::string& filename)
    {
        _fd = ::open(filename.c_str(), O_RDONLY);
        if (_fd < 0) throw std::runtime_error("Cannot open the database: " + filename);
        struct stat st;
        if (::fstat(_fd, &st) != 0 || st.st_size < 0) {
            ::close(_fd);
            throw std::runtime_error("Cannot stat the database: " + filename);
        }
        _size = (size_t)st.st_size;
        if (_size) {
            void* p = ::mmap(nullptr, _size, PROT_READ, MAP_PRIVATE, _fd, 0);
            if (p == MAP_FAILED) {
                ::close(_fd);
                throw std::runtime_error("Cannot map the database: " + filename);
            }
            _data = static_cast<const unsigned char*>(p);
            (void)::madvise(p, _size, MADV_SEQUENTIAL);
        }
    }
    ~mapped_file()
    {
        if (_data) ::munmap(const_cast<unsigned char*>(_data), _size);
        if (_fd >= 0) ::close(_fd);
    }
    mapped_file(const mapped_file&) = delete;
    mapped_file& operator=(const mapped_file&) = delete;
    const unsigned char* data() const { return _data; }
    size_t size() const { return _size; }

private:
    int _fd = -1;
    const unsigned char* _data = nullptr;
    size_t _size = 0;
};

// sequential reads out of the mapping, bounds-checked
struct cursor {
    const unsigned char* at;
    const unsigned char* end;
    template <typename T>
    T pod()
    {
        if ((size_t)(end - at) < sizeof(T)) throw std::runtime_error("Unexpected end of the database file");
        T v;
        std::memcpy(&v, at, sizeof(T));
        at += sizeof(T);
        return v;
    }
    const unsigned char* bytes(size_t n, const char* what)
    {
        if ((size_t)(end - at) < n) throw std::runtime_error(std::string("Unexpected end of the database file (") + what + ")");
        const unsigned char* p = at;
        at += n;
        return p;
    }
};

// crc32 (zlib's polynomial, reflected), a byte at a time over eight tables: ~1 GB/s, once per load of a whole file
uint32_t crc32_of(const unsigned char* p, size_t n, uint32_t crc = 0)
{
    static uint32_t table[8][256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) table[t][i] = (table[t - 1][i] >> 8) ^ table[0][table[t - 1][i] & 0xffu];
        ready = true;
    }
    crc = ~crc;
    while (n >= 8) {
        uint32_t lo, hi;
        std::memcpy(&lo, p, 4);
        std::memcpy(&hi, p + 4, 4);
        lo ^= crc;
        crc = table[7][lo & 0xffu] ^ table[6][(lo >> 8) & 0xffu] ^ table[5][(lo >> 16) & 0xffu] ^ table[4][lo >> 24] ^
              table[3][hi & 0xffu] ^ table[2][(hi >> 8) & 0xffu] ^ table[1][(hi >> 16) & 0xffu] ^ table[0][hi >> 24];
        p += 8, n -= 8;
    }
    while (n--) crc = table[0][(crc ^ *p++) & 0xffu] ^ (crc >> 8);
    return ~crc;
}

struct kmer_record {
    uint32_t key;
    uint32_t kept;               // postings of the record at or above the threshold
    const unsigned char* first;  // its postings in the mapping (all `n` of them)
    uint32_t n;
};

}  // namespace

std::pair<const pkdb_value*, size_t> phylo_kmer_db::search(uint32_t key) const noexcept
{
    const auto it = std::lower_bound(_keys.begin(), _keys.end(), key);
    if (it == _keys.end() || *it != key) return {nullptr, 0};
    const size_t i = (size_t)(it - _keys.begin());
    return {_values.data() + _offsets[i], (size_t)(_offsets[i + 1] - _offsets[i])};
}

phylo_kmer_db load(const std::string& filename, float mu, float omega, size_t max_entries, uint32_t shard_index,
                   uint32_t shard_count)
{
    if (shard_count == 0 || shard_index >= shard_count) throw std::runtime_error("shard_index must be below shard_count");
    const mapped_file file(filename);
    cursor in{file.data(), file.data() + file.size()};
    if (file.size() < 8) throw std::runtime_error("The database file is too short: " + filename);
    if (std::memcmp(in.bytes(8, "magic"), "EPIKAMD1", 8) != 0) {
        // Boost archives start with "22 serialization::archive"; zlib streams with 0x78
        throw std::runtime_error(
            "Unsupported database container: " + filename +
            " is not an EPIKAMD1 file.  IPK's .ipk files (Boost.Serialization inside i2l) cannot be read by this "
            "build.  Convert the database once with tools/ipk2ekdb.cpp, next to a checkout of EPIK that has its i2l "
            "submodule:\n  g++ -std=c++17 -O2 -I<EPIK>/i2l/include tools/ipk2ekdb.cpp -o ipk2ekdb -L<EPIK>/build/i2l "
            "-li2l_dna -lboost_serialization -lboost_iostreams -lboost_filesystem -lz   (proteins: -DSEQ_TYPE_AA -li2l_aa)\n"
            "  ./ipk2ekdb " + filename + " out.ekdb\n(epik_amd/dbfile.py writes the same container from Python)");
    }
    phylo_kmer_db db;
    db._shard_index = shard_index;
    db._shard_count = shard_count;
    db._version = in.pod<uint32_t>();
    const uint32_t seq_type = in.pod<uint32_t>();
    db._sequence_type = seq_type == 0 ? "DNA" : "Proteins";
    db._kmer_size = in.pod<uint32_t>();
    const float built_omega = in.pod<float>();
    const uint64_t num_kmers = in.pod<uint64_t>();
    db._num_entries_total = (size_t)in.pod<uint64_t>();
    const uint64_t newick_len = in.pod<uint64_t>();
    const unsigned char* newick = in.bytes(newick_len, "tree");
    db._tree.assign(reinterpret_cast<const char*>(newick), newick_len);
    if (db._version >= 2) {
        // The trailer (epik_amd/dbfile.py): a conversion that died half way, or a copy cut short, is caught here and
        // not at the first wrong placement.  The counts always; the checksum of the records when the whole file is
        // about to be walked anyway (mu = 1; a smaller mu exists to NOT read the file to its end) or when
        // EPIK_AMD_DB_VERIFY=1 asks for it.
        constexpr size_t kTrailer = 32;
        if ((size_t)(in.end - in.at) < kTrailer || std::memcmp(in.end - kTrailer, "EPIKEND1", 8) != 0)
            throw std::runtime_error("The database file is truncated or was not finished (no EPIKEND1 trailer): " + filename);
        uint64_t kmers_written, entries_written;
        uint32_t crc;
        std::memcpy(&kmers_written, in.end - kTrailer + 8, 8);
        std::memcpy(&entries_written, in.end - kTrailer + 16, 8);
        std::memcpy(&crc, in.end - kTrailer + 24, 4);
        if (kmers_written != num_kmers || entries_written != db._num_entries_total)
            throw std::runtime_error("The database file's trailer counts " + std::to_string(kmers_written) + " k-mers / " +
                                     std::to_string(entries_written) + " phylo-k-mers, its header " + std::to_string(num_kmers) +
                                     " / " + std::to_string(db._num_entries_total) + ": " + filename);
        in.end -= kTrailer;
        const char* verify = std::getenv("EPIK_AMD_DB_VERIFY");
        const bool whole_file = mu >= 1.0f && shard_index == 0;  // (every shard walks the same records: once is enough)
        if ((verify && verify[0] == '1') || (whole_file && !(verify && verify[0] == '0'))) {
            const uint32_t got = crc32_of(in.at, (size_t)(in.end - in.at));
            if (got != crc)
                throw std::runtime_error("The database file's records do not match their checksum (a damaged copy or an unfinished conversion): " + filename);
        }
    }

    const unsigned int sigma = alphabet_size(db._sequence_type);
    // The user's omega replaces the stored one when it is stricter (README.md:125)
    db._omega = std::max(omega, built_omega);
    const float log_thr = std::log10(score_threshold(db._omega, db._kmer_size, sigma));

    uint64_t num_keys = 1;
    for (size_t i = 0; i < db._kmer_size; ++i) {
        num_keys *= sigma;
        if (num_keys > 0xffffffffull) throw std::runtime_error("alphabet_size^k exceeds 2^32 k-mer codes");
    }
    db._num_keys = num_keys;
    const uint64_t kmers_to_load = (uint64_t)std::ceil((double)mu * (double)num_kmers);
    // ---- first walk: which records stay, and how many of their postings (file order: mu and --max-ram cut it)
    std::vector<kmer_record> records;
    uint64_t kept_total = 0;
    // --max-ram with shards: the limit is what ONE shard may keep, but the cut is ONE position in the file for all of
    // them -- where the shards together hold shard_count times the limit -- so that the union of the shards is the
    // prefix an unsharded load with that total would keep (cut shard by shard, every shard would stop somewhere else).
    const bool global_cut = shard_count > 1 && max_entries != std::numeric_limits<size_t>::max();
    const uint64_t limit_all = global_cut && max_entries > std::numeric_limits<uint64_t>::max() / shard_count
                                   ? std::numeric_limits<uint64_t>::max() : (uint64_t)max_entries * (global_cut ? shard_count : 1u);
    uint64_t kept_all = 0;  // every shard's postings so far (global_cut)
    for (uint64_t r = 0; r < num_kmers && r < kmers_to_load; ++r) {
        const uint32_t key = in.pod<uint32_t>();
        const uint32_t n = in.pod<uint32_t>();
        if (key >= num_keys) throw std::runtime_error("k-mer code out of range in the database");
        const unsigned char* first = in.bytes((size_t)n * sizeof(pkdb_value), "postings");
        const bool mine = key % shard_count == shard_index;
        if (!mine && !global_cut) continue;
        uint32_t kept = 0;
        for (uint32_t j = 0; j < n; ++j) {
            float score;
            std::memcpy(&score, first + (size_t)j * sizeof(pkdb_value) + offsetof(pkdb_value, score), sizeof score);
            kept += score >= log_thr;
        }
        // --max-ram: stop in front of the k-mer that does not fit
        if (global_cut ? kept_all + kept > limit_all : kept_total + kept > max_entries) break;
        kept_all += kept;
        if (!mine) continue;
        kept_total += kept;
        if (kept) records.push_back(kmer_record{key, kept, first, n});
    }
    // (the cut holds for the shards TOGETHER: with unevenly long lists over the residues of key % shard_count one shard
    // may end up over the limit the user named for each -- said here, the load goes on: the limit is a budget, and the
    // placer's own plan says what the device takes)
    if (global_cut && kept_total > max_entries + max_entries / 8)
        std::fprintf(stderr,
                     "warning: --max-ram: shard %u of %u keeps %llu postings, %.0f %% over the %llu a shard was to keep "
                     "(the cut is one position of the file for all shards; their lists are unevenly long)\n",
                     shard_index, shard_count, (unsigned long long)kept_total,
                     100.0 * ((double)kept_total / (double)max_entries - 1.0), (unsigned long long)max_entries);
    // ---- by k-mer code: the sparse CSR the C ABI takes
    std::sort(records.begin(), records.end(), [](const kmer_record& a, const kmer_record& b) { return a.key < b.key; });
    db._keys.resize(records.size());
    db._offsets.resize(records.size() + 1);
    db._offsets[0] = 0;
    for (size_t i = 0; i < records.size(); ++i) {
        if (i && records[i].key == records[i - 1].key) throw std::runtime_error("duplicate k-mer in the database");
        db._keys[i] = records[i].key;
        db._offsets[i + 1] = db._offsets[i] + records[i].kept;
    }
    // ---- second walk: every kept posting once, straight to its place
    db._values.resize(kept_total);
    for (size_t i = 0; i < records.size(); ++i) {
        pkdb_value* dst = db._values.data() + db._offsets[i];
        const kmer_record& rec = records[i];
        for (uint32_t j = 0; j < rec.n; ++j) {
            pkdb_value v;
            std::memcpy(&v, rec.first + (size_t)j * sizeof(pkdb_value), sizeof v);
            if (v.score >= log_thr) *dst++ = v;
        }
    }
    db._tree_index = io::parse_newick(db._tree).tree_index();
    return db;
}

}  // namespace epik_amd
