// host_test.cpp -- CPU unit checks of the host-side callers (FASTA, Newick, jplace
// formatting, --max-ram parsing, database container).  Run by tests/test_host_cpu.py;
// exits non-zero on the first failure.  No GPU call is made.
#include <cmath>
#include <cstdio>
#include <fstream>
#include <limits>
#include <iostream>
#include <sstream>

#include "jplace.hpp"
#include "phylo_kmer_db.hpp"
#include "phylo_tree.hpp"
#include "seq_record.hpp"

#include "report.hpp"

static int failures = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            std::cerr << "FAILED " << __FILE__ << ":" << __LINE__ << ": " #cond << std::endl; \
            ++failures;                                                          \
        }                                                                        \
    } while (0)

int main(int argc, char** argv)
{
    using namespace epik_amd;
    // host_test load <file>: epik_amd::load by itself (tests/test_host_cpu.py hands it damaged containers; the driver
    // asks for its devices first)
    // ... host_test load <file> <max_entries> <shard_index> <shard_count>: the k-mer codes that shard keeps
    if ((argc == 3 || argc == 6) && std::string(argv[1]) == "load") {
        try {
            const size_t limit = argc == 6 ? (size_t)std::stoull(argv[3]) : std::numeric_limits<size_t>::max();
            const auto db = load(argv[2], 1.0f, 1.5f, limit, argc == 6 ? (uint32_t)std::stoul(argv[4]) : 0u,
                                 argc == 6 ? (uint32_t)std::stoul(argv[5]) : 1u);
            std::cout << "loaded " << db.get_num_entries_loaded() << " phylo-k-mers, version " << db.version() << std::endl;
            if (argc == 6) {
                std::cout << "keys";
                for (const auto key : db.keys()) std::cout << ' ' << key;
                std::cout << std::endl;
            }
            return 0;
        } catch (const std::exception& e) {
            std::cerr << "Error: " << e.what() << std::endl;
            return 255;
        }
    }
    const std::string tmp = argc > 1 ? argv[1] : "/tmp";

    // --- FASTA batches (main.cpp:332-340) ---
    {
        const std::string path = tmp + "/host_test.fasta";
        std::ofstream(path) << ">r1 first\nACGT\nAC\n\n>r2\r\nTTTT\r\n>r3\n>r4\nGG\n";
        io::batch_fasta reader(path, 2);
        auto b1 = reader.next_batch();
        CHECK(b1.size() == 2 && b1[0].header() == "r1 first" && b1[0].sequence() == "ACGTAC");
        CHECK(b1[1].header() == "r2" && b1[1].sequence() == "TTTT");
        auto b2 = reader.next_batch();
        CHECK(b2.size() == 2 && b2[0].header() == "r3" && b2[0].sequence().empty());
        CHECK(b2[1].header() == "r4" && b2[1].sequence() == "GG");
        CHECK(reader.next_batch().empty());
        CHECK(reader.bytes_read() > 30);
    }
    // --- Newick: post-order ids, tree index, jplace form ---
    {
        const auto tree = io::parse_newick("((A:0.1,B:0.2)X:0.3,C:0.4)root;");
        CHECK(tree.get_node_count() == 5);
        CHECK((*tree.get_by_postorder_id(0))->get_label() == "A");
        CHECK((*tree.get_by_postorder_id(2))->get_label() == "X");
        CHECK((*tree.get_by_postorder_id(4))->get_label() == "root");
        CHECK(!tree.get_by_postorder_id(5));
        const auto index = tree.tree_index();
        CHECK(index[2].subtree_num_nodes == 3 && std::fabs(index[2].subtree_total_length - 0.3) < 1e-12);
        CHECK(index[4].subtree_num_nodes == 5 && std::fabs(index[4].subtree_total_length - 1.0) < 1e-12);
        CHECK(io::to_newick(tree, true) == "((A:0.1{0},B:0.2{1})X:0.3{2},C:0.4{3})root:0{4};");
        const auto again = io::parse_newick(io::to_newick(tree, true));
        CHECK(again.get_node_count() == 5 && (*again.get_by_postorder_id(3))->get_branch_length() == 0.4);
    }
    // --- --max-ram, --mu (main.cpp:154-202) ---
    {
        CHECK(parse_memory_size("512") == 512);
        CHECK(parse_memory_size("256K") == 256 * 1024);
        CHECK(parse_memory_size("42m") == 42u * 1024 * 1024);
        CHECK(parse_memory_size("4.2Gb") == (size_t)(4.2 * 1024 * 1024 * 1024));
        CHECK(parse_memory_size("7 B") == 7 && parse_memory_size("1.5k") == 1536);
        bool threw = false;
        try { parse_memory_size("12X"); } catch (const std::runtime_error&) { threw = true; }
        CHECK(threw);
        threw = false;
        try { parse_memory_size("lots"); } catch (const std::runtime_error&) { threw = true; }
        CHECK(threw);
        check_mu(0.0f);
        check_mu(1.0f);
        threw = false;
        try { check_mu(1.5f); } catch (const std::runtime_error&) { threw = true; }
        CHECK(threw);
    }
    // --- the report lines (main.cpp:285-292, 368-381): counts in units of 1024 with K / M / B, one
    // decimal unless whole; durations as [D day(s), ][HH:]MM:SS ---
    {
        CHECK(human_count((size_t)0) == "0" && human_count((size_t)1023) == "1023");
        CHECK(human_count((size_t)1024) == "1K" && human_count((size_t)1536) == "1.5K");
        CHECK(human_count((size_t)37192403) == "35.5M" && human_count((size_t)1048576) == "1M");
        CHECK(human_count(5368709120.0, true) == "5B" && human_count(712.25, false) == "712.250000");
        CHECK(human_count(712250.7, false) == "695.6K");
        CHECK(human_duration(0) == "00:00" && human_duration(61500) == "01:01");
        CHECK(human_duration(3600 * 1000) == "01:00:00" && human_duration(86400u * 1000 + 5000) == "1 day, 00:00:05");
        CHECK(human_duration(3u * 86400 * 1000 + 3723 * 1000) == "3 days, 01:02:03");
    }
    // --- thresholds and class tables ---
    {
        CHECK(std::fabs(score_threshold(1.5f, 10, 4) - std::pow(0.375, 10)) < 1e-12);
        const auto dna = char_class_table("DNA");
        CHECK(dna['A'] == 1 && dna['c'] == 2 && dna['G'] == 4 && dna['U'] == 8 && dna['N'] == 15 && dna['-'] == 0);
        const auto aa = char_class_table("Proteins");
        CHECK(aa['R'] == 1 && aa['V'] == (1u << 19) && aa['X'] == (1u << 20) - 1 && aa['*'] == 0);
    }
    // --- jplace document shape (jplace.cpp) ---
    {
        const std::string path = tmp + "/host_test.jplace";
        std::vector<seq_record> batch{{"q1", "ACGT"}, {"q \"2\"", "ACGT"}, {"q3", "TT"}};
        impl::placed_collection placed;
        placed.sequence_map[batch[0].sequence()] = {batch[0].header(), batch[1].header()};
        placed.sequence_map[batch[2].sequence()] = {batch[2].header()};
        placed.placed_seqs.push_back({batch[0].sequence(), {{3, -1.5f, 0.75, 2, 0.05, 0.15}, {1, -2.0f, 0.25, 1, 0.1, 0.2}}});
        placed.placed_seqs.push_back({batch[2].sequence(), {}});
        io::jplace_writer writer(path, "epik-dna -d x ", "(A:1{0},B:2{1}):0{2};");
        writer.start();
        writer << placed;
        writer.end();
        std::stringstream ss;
        ss << std::ifstream(path).rdbuf();
        const std::string doc = ss.str();
        CHECK(doc.find("\"version\": 3") != std::string::npos);
        CHECK(doc.find("[\"edge_num\", \"likelihood\", \"like_weight_ratio\", \"distal_length\", \"pendant_length\"]") !=
              std::string::npos);
        CHECK(doc.find("[3, -1.5, 0.75, 0.05, 0.15]") != std::string::npos);
        CHECK(doc.find("[\"q \\\"2\\\"\", 1]") != std::string::npos);
        CHECK(io::json_double(1.0) == "1.0" && io::json_double(-4.2596874237060547) == "-4.259687423706055");
    }
    // --- database container: mu / omega / max-ram filtering ---
    {
        const std::string path = tmp + "/host_test.ekdb";
        {
            std::ofstream out(path, std::ios::binary);
            const std::string newick = "((A:0.1,B:0.2)X:0.3,C:0.4)root;";
            auto w32 = [&](uint32_t v) { out.write(reinterpret_cast<const char*>(&v), 4); };
            auto w64 = [&](uint64_t v) { out.write(reinterpret_cast<const char*>(&v), 8); };
            auto wf = [&](float v) { out.write(reinterpret_cast<const char*>(&v), 4); };
            out.write("EPIKAMD1", 8);
            w32(1); w32(0); w32(2); wf(1.0f);            // version, DNA, k=2, omega
            w64(3); w64(5); w64(newick.size());          // k-mers, postings, tree
            out << newick;
            w32(5); w32(2); w32(0); wf(-0.1f); w32(2); wf(-0.5f);   // k-mer 5: branches 0, 2
            w32(1); w32(2); w32(1); wf(-0.2f); w32(3); wf(-1.0f);   // k-mer 1: branch 3 is below omega 1.5
            w32(9); w32(1); w32(4); wf(-0.3f);                      // k-mer 9
        }
        auto db = load(path, 1.0f, 1.0f);  // log10((1/4)^2) = -1.20: everything passes
        CHECK(db.kmer_size() == 2 && db.num_keys() == 16 && db.get_num_entries_loaded() == 5);
        // the sparse form: the present codes ascending, a list each
        CHECK(db.keys().size() == 3 && db.keys()[0] == 1 && db.keys()[1] == 5 && db.keys()[2] == 9);
        CHECK(db.offsets().size() == 4 && db.offsets()[1] == 2 && db.offsets()[2] == 4 && db.offsets()[3] == 5);
        CHECK(db.search(5).second == 2 && db.search(5).first[0].branch == 0 && db.search(9).first[0].branch == 4);
        CHECK(db.search(2).first == nullptr && db.search(15).second == 0);
        db = load(path, 1.0f, 1.5f);   // log10((1.5/4)^2) = -0.85: the -1.0 posting goes
        CHECK(db.get_num_entries_loaded() == 4 && db.get_num_entries_total() == 5 && db.omega() == 1.5f);
        db = load(path, 0.5f, 1.0f);   // best half of the k-mers: ceil(1.5) = 2 records
        CHECK(db.get_num_entries_loaded() == 4);
        db = load(path, 1.0f, 1.0f, 3);  // --max-ram: stops in front of the k-mer that does not fit
        CHECK(db.get_num_entries_loaded() == 2);
        // --db-shard: the codes with code % 2 == 1 (1, 5, 9: all of them) / == 0 (none); % 4: 1, 5, 9 -> shard 1
        db = load(path, 1.0f, 1.0f, std::numeric_limits<size_t>::max(), 1, 2);
        CHECK(db.get_num_entries_loaded() == 5 && db.shard_index() == 1 && db.shard_count() == 2 && db.num_keys() == 16);
        db = load(path, 1.0f, 1.0f, std::numeric_limits<size_t>::max(), 0, 2);
        CHECK(db.get_num_entries_loaded() == 0 && db.keys().empty() && db.offsets().size() == 1);
        db = load(path, 1.0f, 1.0f, std::numeric_limits<size_t>::max(), 1, 4);
        CHECK(db.keys().size() == 3);
        db = load(path, 1.0f, 1.0f, std::numeric_limits<size_t>::max(), 2, 4);
        CHECK(db.keys().empty());
        bool threw = false;
        try { load(tmp + "/host_test.fasta"); } catch (const std::runtime_error& e) {
            threw = std::string(e.what()).find("EPIKAMD1") != std::string::npos;
        }
        CHECK(threw);
    }
    if (failures == 0) std::cout << "host tests ok" << std::endl;
    return failures ? 1 : 0;
}
