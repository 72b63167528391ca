// ipk2ekdb.cpp -- converts an IPK database (.ipk, what the reference loads at main.cpp:277) into this build's
// EPIKAMD1 container (epik_amd/host/phylo_kmer_db.hpp, epik_amd/dbfile.py), for people who HAVE i2l.
//
// CANNOT BE COMPILED OR VALIDATED IN THIS REPOSITORY'S BUILD ENVIRONMENT: i2l (github.com/phylo42/i2l, a git
// submodule of the reference at path `i2l`, pinned commit unknown), Boost.Serialization / iostreams and zlib are
// not available there, and no sample .ipk exists to test against.  It is written against the i2l surface the
// reference itself uses (every call below names the reference line that makes the same call) and
// is the one-command step between "an .ipk" and "a database this placer loads".  Build it next to a checkout of
// the reference that has its submodule:
//
//   g++ -std=c++17 -O2 -I<EPIK>/i2l/include tools/ipk2ekdb.cpp -o ipk2ekdb \
//       -L<EPIK>/build/i2l -li2l_dna -lboost_serialization -lboost_iostreams -lboost_filesystem -lz
//   (protein databases: -DSEQ_TYPE_AA and -li2l_aa, as the reference's two binaries, epik/CMakeLists.txt:72,124)
//
//   ./ipk2ekdb in.ipk out.ekdb          then:  epik.py place -i out.ekdb -o outdir query.fasta
//
// ASSUMPTION REGISTER (what cannot be checked without i2l; each is one function or one line here):
//   A1  iterating a phylo_kmer_db yields (key, entries) pairs, entries iterable as pkdb_value {branch, score}
//       (place.cpp:300-304,358 use exactly that shape through search()).
//   A2  i2l::decode_kmer(key, k) returns the k-mer's letters (it is how xpas prints k-mers).  The letters, not the
//       key, decide this container's code: base-sigma number of the states A C G T / R H K D E S T N Q C G P A I L
//       M F W Y V, first letter most significant (epik_amd/alphabet.py) -- so i2l's own key packing (2 bits per
//       base; 5 bits per residue or base 20) does not matter.  -DIPK2EKDB_RAW_KEYS skips the decode and takes the
//       key as it is (right for DNA if i2l packs A0 C1 G2 T3, 2 bits each, first base most significant).
//   A3  load(file, 1.0, 1.0, unlimited) loads everything the file holds (main.cpp:277 with --mu 1.0, no --max-ram,
//       an omega no stricter than the one the database was built with), and db.omega() then says which omega the
//       postings were filtered with.  If it says the omega passed in instead (1.0), give the build omega as the
//       third argument: it is what the placer's --omega is compared with (README.md:125).
//   A4  record order of the output: most informative k-mer first, informativeness = the list's best score,
//       descending, ties by code -- the rule epik_amd/dbfile.py uses; IPK's own ranking is inside i2l's file
//       order, which a loaded hash map no longer shows.  --mu and --max-ram of the placer cut THIS order.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include <i2l/phylo_kmer_db.h>   // i2l::phylo_kmer_db, i2l::pkdb_value       (place.cpp:6-12 include the same headers)
#include <i2l/serialization.h>   // i2l::load                                  (main.cpp:277)
#include <i2l/seq.h>             // i2l::decode_kmer, i2l::seq_type             (A2)

namespace {

struct posting {
    uint32_t branch;
    float score;
};

template <typename T>
void put(std::ofstream& out, T v) { out.write(reinterpret_cast<const char*>(&v), sizeof v); }

// crc32 (zlib's polynomial) of the record bytes, for the container's trailer
uint32_t crc32_more(uint32_t crc, const void* data, size_t n)
{
    const unsigned char* p = static_cast<const unsigned char*>(data);
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) {
        crc ^= p[i];
        for (int k = 0; k < 8; ++k) crc = (crc & 1u) ? 0xEDB88320u ^ (crc >> 1) : crc >> 1;
    }
    return ~crc;
}

// A2: the code of a k-mer in this container, from its letters
uint32_t code_of(const std::string& letters, bool amino)
{
    static const std::string dna = "ACGT", aa = "RHKDESTNQCGPAILMFWYV";
    const std::string& states = amino ? aa : dna;
    uint64_t code = 0;
    for (char c : letters) {
        if (c == 'U') c = 'T';
        const auto s = states.find(c);
        if (s == std::string::npos) throw std::runtime_error("a k-mer of the database holds the letter '" + std::string(1, c) + "'");
        code = code * states.size() + s;
    }
    if (code > 0xffffffffull) throw std::runtime_error("k-mer code beyond 32 bits (nucl k <= 15, amino k <= 7)");
    return (uint32_t)code;
}

}  // namespace

int main(int argc, char** argv)
{
    if (argc != 3 && argc != 4) {
        std::cerr << "usage: ipk2ekdb <in.ipk> <out.ekdb> [omega the database was built with]" << std::endl;
        return 1;
    }
    try {
        // A3: everything -- mu 1.0, the lowest omega the format allows (the file's own threshold stays), no limit
        const auto db = i2l::load(argv[1], 1.0f, 1.0f, std::numeric_limits<size_t>::max());   // main.cpp:277
        const bool amino = db.sequence_type() != "DNA";                                      // main.cpp:286
        const size_t k = db.kmer_size();                                                     // main.cpp:287
        struct record {
            uint32_t code;
            float best;
            std::vector<posting> list;
        };
        std::vector<record> records;
        uint64_t total = 0;
        for (const auto& [key, entries] : db) {                                              // A1
            record r;
#ifdef IPK2EKDB_RAW_KEYS
            r.code = (uint32_t)key;
#else
            r.code = code_of(i2l::decode_kmer(key, k), amino);                               // A2
#endif
            r.best = -std::numeric_limits<float>::infinity();
            for (const auto& [branch, score] : entries) {                                    // place.cpp:358
                r.list.push_back({(uint32_t)branch, (float)score});
                r.best = std::max(r.best, (float)score);
            }
            total += r.list.size();
            if (!r.list.empty()) records.push_back(std::move(r));
        }
        std::sort(records.begin(), records.end(), [](const record& a, const record& b) {   // A4
            return a.best != b.best ? a.best > b.best : a.code < b.code;
        });
        // EPIKAMD1 (epik_amd/dbfile.py): magic | u32 version | u32 sequence type | u32 k | f32 omega |
        // u64 k-mers | u64 postings | u64 tree length | tree | per k-mer { u32 code | u32 n | n x {u32 branch, f32 score} } | trailer
        std::ofstream out(argv[2], std::ios::binary);
        if (!out) throw std::runtime_error(std::string("cannot create ") + argv[2]);
        const std::string tree = db.tree();                                                  // main.cpp:294
        out.write("EPIKAMD1", 8);
        put<uint32_t>(out, 2);  // version 2: the trailer below
        put<uint32_t>(out, amino ? 1u : 0u);
        put<uint32_t>(out, (uint32_t)k);
        put<float>(out, argc == 4 ? std::stof(argv[3]) : (float)db.omega());               // main.cpp:288, A3
        put<uint64_t>(out, records.size());
        put<uint64_t>(out, total);
        put<uint64_t>(out, tree.size());
        out.write(tree.data(), (std::streamsize)tree.size());
        uint32_t crc = 0;
        for (const auto& r : records) {
            const uint32_t head[2] = {r.code, (uint32_t)r.list.size()};
            out.write(reinterpret_cast<const char*>(head), sizeof head);
            out.write(reinterpret_cast<const char*>(r.list.data()), (std::streamsize)(r.list.size() * sizeof(posting)));
            crc = crc32_more(crc32_more(crc, head, sizeof head), r.list.data(), r.list.size() * sizeof(posting));
        }
        // the trailer: "EPIKEND1" | u64 k-mers | u64 postings | u32 crc32 of the records | u32 0 -- the loader refuses
        // a file without it (a conversion that died half way)
        out.write("EPIKEND1", 8);
        put<uint64_t>(out, records.size());
        put<uint64_t>(out, total);
        put<uint32_t>(out, crc);
        put<uint32_t>(out, 0u);
        if (!out) throw std::runtime_error(std::string("cannot write ") + argv[2]);
        std::cout << "wrote " << records.size() << " k-mers, " << total << " phylo-k-mers, k = " << k << ", "
                  << (amino ? "Proteins" : "DNA") << std::endl;
    } catch (const std::exception& e) {
        std::cerr << "ipk2ekdb: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
