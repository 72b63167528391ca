// main.cpp -- the `epik-dna` / `epik-aa` drivers over the MI355X placer.
//
// Keeps the command line of the reference driver (epik/src/epik/main.cpp:209-222):
//   -d/--database  -q/--query  -j/--jobs  --batch-size  --omega  --mu  --max-ram
//   -o/--output-dir  --keep-at-most  --keep-factor  -h/--help
// with the same defaults (1, 2000, 1.5, 1.0, -, -, 7, 0.01), the same exit codes
// (0 / -1 on error, main.cpp:272,282,387,390), the same output file name
// (main.cpp:34-37) and the same final report lines (main.cpp:368-382).  Not reproduced:
// the progress bar and colours (indicators/termcolor).  Added: --gpus N | --devices a,b,c (the database on
// every device, batches shared out) and --db-shard G (the database cut in G by k-mer code, shard g on device g:
// every batch is placed by all of them together -- a database larger than one device's memory).
// The two binaries differ as the reference's do (epik/CMakeLists.txt:72,124): epik-dna
// accepts DNA databases, epik-aa protein ones.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <exception>
#include <mutex>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <limits>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "jplace.hpp"
#include "phylo_kmer_db.hpp"
#include "phylo_tree.hpp"
#include "placer.hpp"
#include "report.hpp"
#include "seq_record.hpp"

#ifndef EPIK_AMD_NO_MAIN
namespace {

/// Hand-over between two stages of the driver: at most `capacity` items wait; after close() push()
/// returns false and the pops return what is left, then false.
template <typename T>
class bounded_queue {
public:
    explicit bounded_queue(size_t capacity) : _capacity(capacity) {}
    bool push(T item)
    {
        std::unique_lock<std::mutex> lock(_mutex);
        _not_full.wait(lock, [&] { return _closed || _items.size() < _capacity; });
        if (_closed) return false;
        _items.push_back(std::move(item));
        _not_empty.notify_one();
        return true;
    }
    bool pop(T& item)
    {
        std::unique_lock<std::mutex> lock(_mutex);
        _not_empty.wait(lock, [&] { return _closed || !_items.empty(); });
        if (_items.empty()) return false;  // closed and drained
        item = std::move(_items.front());
        _items.pop_front();
        _not_full.notify_one();
        return true;
    }
    /// Waits for at least one item, then takes what is there, `max_items` at most, in order.
    /// False once the queue is closed and drained.
    bool pop_up_to(std::vector<T>& items, size_t max_items)
    {
        items.clear();
        std::unique_lock<std::mutex> lock(_mutex);
        _not_empty.wait(lock, [&] { return _closed || !_items.empty(); });
        while (!_items.empty() && items.size() < max_items) {
            items.push_back(std::move(_items.front()));
            _items.pop_front();
        }
        _not_full.notify_all();
        return !items.empty();
    }
    bool pop_all(std::vector<T>& items) { return pop_up_to(items, std::numeric_limits<size_t>::max()); }
    void close()
    {
        std::lock_guard<std::mutex> lock(_mutex);
        _closed = true;
        _not_full.notify_all();
        _not_empty.notify_all();
    }

private:
    std::mutex _mutex;
    std::condition_variable _not_full, _not_empty;
    std::deque<T> _items;
    size_t _capacity;
    bool _closed = false;
};

/// Busy time of one stage.
class stage_clock {
public:
    void start() { _begin = std::chrono::steady_clock::now(); }
    void stop() { _total += std::chrono::steady_clock::now() - _begin; }
    double ms() const { return std::chrono::duration<double, std::milli>(_total).count(); }

private:
    std::chrono::steady_clock::time_point _begin;
    std::chrono::steady_clock::duration _total{};
};

#ifdef EPIK_AMD_AA
constexpr const char* kSequenceType = "Proteins";
#else
constexpr const char* kSequenceType = "DNA";
#endif

/// main.cpp:22-32
std::string make_invocation(int argc, char** argv)
{
    std::string invocation;
    for (int i = 0; i < argc; ++i) invocation += std::string(argv[i]) + " ";
    return invocation;
}

/// main.cpp:34-37: <output_dir>/placements_<basename(query)>.jplace
std::string make_output_filename(const std::string& input_file, const std::string& output_dir)
{
    const auto slash = input_file.find_last_of('/');
    const std::string base = slash == std::string::npos ? input_file : input_file.substr(slash + 1);
    std::string dir = output_dir;
    if (!dir.empty() && dir.back() != '/') dir.push_back('/');
    return dir + "placements_" + base + ".jplace";
}

}  // namespace
#endif  // EPIK_AMD_NO_MAIN

#ifndef EPIK_AMD_NO_MAIN

namespace {

const char* kHelp =
    "Evolutionary Placement with Informative K-mers (MI355X placer)\n"
    "Usage:\n"
    "  epik-dna|epik-aa [OPTION...]\n\n"
    "  -d, --database arg      IPK database\n"
    "  -q, --query arg         Input query file (.fasta)\n"
    "  -j, --jobs arg          Num threads (default: 1)\n"
    "      --batch-size arg    Batch size (default: 2000)\n"
    "      --omega arg         Determines the threshold value (default: 1.5)\n"
    "      --mu arg            Proportion of the database to load (default: 1.0)\n"
    "      --max-ram arg       Approximate database size to load, MB\n"
    "  -o, --output-dir arg    Output directory\n"
    "      --keep-at-most arg  Number of branches to report (default: 7)\n"
    "      --keep-factor arg   Minimum LWR to report (default: 0.01)\n"
    "      --gpus arg          Number of MI355X devices to use (default: 1)\n"
    "      --devices arg       Comma-separated HIP device ordinals (overrides --gpus)\n"
    "      --db-shard arg      Cut the database in so many shards by k-mer code, one per device\n"
    "                          (a database larger than one device; default: 1 = replicate it)\n"
    "  -h, --help              Print usage\n";

struct options {
    std::map<std::string, std::string> values;
    bool has(const std::string& k) const { return values.count(k) != 0; }
    std::string get(const std::string& k, const std::string& def) const
    {
        const auto it = values.find(k);
        return it == values.end() ? def : it->second;
    }
    std::string require(const std::string& k) const
    {
        const auto it = values.find(k);
        if (it == values.end()) throw std::runtime_error("Option '" + k + "' has no value");
        return it->second;
    }
};

options parse_args(int argc, char** argv)
{
    const std::map<std::string, std::string> short_names{
        {"-d", "database"}, {"-q", "query"}, {"-j", "jobs"}, {"-o", "output-dir"}, {"-h", "help"}};
    options opt;
    for (int i = 1; i < argc; ++i) {
        std::string arg = argv[i];
        std::string name, value;
        bool have_value = false;
        if (arg.rfind("--", 0) == 0) {
            name = arg.substr(2);
            const auto eq = name.find('=');
            if (eq != std::string::npos) {
                value = name.substr(eq + 1);
                name = name.substr(0, eq);
                have_value = true;
            }
        } else if (short_names.count(arg)) {
            name = short_names.at(arg);
        } else {
            continue;  // positional arguments are ignored (epik.py passes the query twice, epik.py:88,96)
        }
        if (name == "help") {
            opt.values[name] = "1";
            continue;
        }
        if (!have_value) {
            if (i + 1 >= argc) throw std::runtime_error("Option '" + name + "' is missing an argument");
            value = argv[++i];
        }
        opt.values[name] = value;
    }
    return opt;
}

}  // namespace

int main(int argc, char** argv)
{
    std::ios::sync_with_stdio(false);
    if (argc == 1) {
        std::cout << kHelp << std::endl;
        return 0;
    }
    try {
        const options parsed = parse_args(argc, argv);
        if (parsed.has("help")) {
            std::cout << kHelp << std::endl;
            return 0;
        }
        const auto db_file = parsed.require("database");
        const auto query_file = parsed.require("query");
        const auto num_threads = (size_t)std::stoul(parsed.get("jobs", "1"));
        const auto batch_size = (size_t)std::stoul(parsed.get("batch-size", "2000"));
        const auto user_omega = std::stof(parsed.get("omega", "1.5"));
        const auto user_mu = std::stof(parsed.get("mu", "1.0"));
        const auto keep_at_most = (size_t)std::stoul(parsed.get("keep-at-most", "7"));
        const auto keep_factor = std::stod(parsed.get("keep-factor", "0.01"));
        const auto output_dir = parsed.require("output-dir");
        epik_amd::check_mu(user_mu);

        size_t max_entries = std::numeric_limits<size_t>::max();
        if (parsed.has("max-ram")) {
            const auto max_ram = epik_amd::parse_memory_size(parsed.require("max-ram"));
            max_entries = static_cast<size_t>(max_ram / sizeof(epik_amd::pkdb_value));
            if (max_entries == 0) throw std::runtime_error("Memory limit is too low");
            std::cout << "Max-RAM provided: will be loaded not more than " << epik_amd::human_count(max_entries)
                      << " phylo-k-mers." << std::endl;
        }

        const auto db_shards = (uint32_t)std::stoul(parsed.get("db-shard", "1"));
        if (db_shards == 0 || db_shards > EPIK_AMD_MAX_SHARDS)
            throw std::runtime_error("--db-shard must be between 1 and " + std::to_string(EPIK_AMD_MAX_SHARDS));
        std::vector<int> devices;
        if (parsed.has("devices")) {
            std::stringstream ss(parsed.require("devices"));
            for (std::string item; std::getline(ss, item, ',');) devices.push_back(std::stoi(item));
        } else {
            // (sharded: a device per shard while there are devices; --gpus / --devices say otherwise)
            const int n = parsed.has("gpus") || db_shards == 1 ? std::stoi(parsed.get("gpus", "1"))
                                                                : std::min<int>((int)db_shards, std::max(1, epik_amd_device_count()));
            for (int i = 0; i < n; ++i) devices.push_back(i);
        }
        for (int device : devices)
            if (device < 0 || device >= epik_amd_device_count())
                throw std::runtime_error("HIP device " + std::to_string(device) + " is not available: " +
                                         std::to_string(epik_amd_device_count()) +
                                         " visible (this placer has no CPU fallback)");

        std::cout << "Loading database with mu=" << user_mu << " and omega=" << user_omega << "..." << std::endl;
        // --db-shard G: this is shard 0 of G; the placer's constructor loads the others one after the other and
        // drops each as soon as its lists are on its device (--max-ram then bounds what ONE shard keeps)
        auto db = epik_amd::load(db_file, user_mu, user_omega, max_entries, 0, db_shards);
        if (db.version() < epik_amd::protocol::EARLIEST_INDEX) {
            std::cerr << "The serialization protocol version is too old (v" << db.version() << ").\n";
            return -1;
        }
        if (db.sequence_type() != kSequenceType)
            throw std::runtime_error(std::string("This binary places ") + kSequenceType + " databases, the file holds " +
                                     db.sequence_type());

        std::cout << "Database parameters:" << std::endl
                  << "\tSequence type: " << db.sequence_type() << std::endl
                  << "\tk: " << db.kmer_size() << std::endl
                  << "\tomega: " << db.omega() << std::endl
                  << "\tPositions loaded: " << (db.positions_loaded() ? "true" : "false") << std::endl
                  << std::endl;
        std::cout << "Loaded " << epik_amd::human_count(db.get_num_entries_loaded()) << " of "
                  << epik_amd::human_count(db.get_num_entries_total()) << " phylo-k-mers"
                  << (db_shards > 1 ? " (shard 0 of " + std::to_string(db_shards) + ")" : std::string()) << ". " << std::endl
                  << std::endl;

        const auto tree = epik_amd::io::parse_newick(db.tree());
        const epik_amd::placer::shard_loader load_shard = [&](uint32_t g) {
            auto part = epik_amd::load(db_file, user_mu, user_omega, max_entries, g, db_shards);
            std::cout << "Loaded " << epik_amd::human_count(part.get_num_entries_loaded()) << " phylo-k-mers (shard " << g
                      << " of " << db_shards << ")." << std::endl;
            return part;
        };
        epik_amd::placer placer(db, tree, keep_at_most, keep_factor, num_threads, devices, db_shards, load_shard);
        db.drop_lists();  // the lists are on the devices now; tree, k and omega stay for the output
        const auto tree_as_newick = epik_amd::io::to_newick(tree, true);
        const auto jplace_filename = make_output_filename(query_file, output_dir);
        const auto invocation = make_invocation(argc, argv);

        epik_amd::io::jplace_writer jplace(jplace_filename, invocation, tree_as_newick);
        jplace.set_branch_lengths(placer.distal_lengths(), placer.pendant_lengths());
        jplace.start();

        std::cout << "Instruction set: gfx950 (" << placer.handle_count()
                  << (db_shards > 1 ? " shard(s) of the database, one handle each)" : " device(s))") << std::endl;
        std::cout << "Placing " << query_file << "..." << std::endl;
        const auto begin = std::chrono::steady_clock::now();
        size_t num_seq_placed = 0;
        double average_speed = 0.0;
        size_t num_iterations = 0;

        // Stages on their own threads, batches handed on through queues: the FASTA reader; one placer
        // thread per device; the jplace writer (formatting on `--jobs` threads).  The reference runs
        // read, place and write one after the other per batch (main.cpp:336-361); batch boundaries,
        // dedup per batch (place.cpp:207) and output order are the same here.  A placer thread takes
        // every batch that is waiting (up to kGroupBatches) as ONE launch -- a 2000-read batch is far too
        // small to fill a device -- so with several devices whole groups of batches go to them in turn,
        // and the writer puts the batches back in input order.
        constexpr size_t kGroupBatches = 64;
        struct work_item {
            size_t sequence = 0;                          // position of the batch in the input
            std::vector<epik_amd::seq_record> batch;     // owns the bytes the views below point into
            epik_amd::impl::placed_batch placed;
        };
        const size_t n_devices = placer.device_count();
        bounded_queue<work_item> to_place(2 * kGroupBatches * n_devices);
        bounded_queue<work_item> to_write(2 * kGroupBatches * n_devices);
        std::exception_ptr reader_error, writer_error;
        std::vector<std::exception_ptr> placer_errors(n_devices);
        stage_clock read_clock, write_clock;
        std::vector<stage_clock> place_clocks(n_devices);
        std::mutex stats_mutex;
        // (the records of a batch are views into the reader's mapping of the file: it stays until all is written)
        epik_amd::io::batch_fasta reader(query_file, batch_size);
        std::thread reader_thread([&] {
            try {
                for (size_t sequence = 0;; ++sequence) {
                    read_clock.start();
                    auto batch = reader.next_batch();
                    read_clock.stop();
                    if (batch.empty() || !to_place.push(work_item{sequence, std::move(batch), {}})) break;
                }
            } catch (...) {
                reader_error = std::current_exception();
            }
            to_place.close();  // the placer threads drain what is queued, then stop
        });
        std::thread writer_thread([&] {
            try {
                std::map<size_t, work_item> waiting;  // batches that arrived ahead of their turn
                size_t next = 0;
                std::vector<work_item> arrived;
                while (to_write.pop_all(arrived)) {
                    for (auto& item : arrived) waiting.emplace(item.sequence, std::move(item));
                    std::vector<const epik_amd::impl::placed_batch*> group;
                    std::vector<work_item> ready;  // keeps the batches alive while they are written
                    for (auto it = waiting.find(next); it != waiting.end(); it = waiting.find(next)) {
                        ready.push_back(std::move(it->second));
                        waiting.erase(it);
                        ++next;
                    }
                    for (const auto& item : ready) group.push_back(&item.placed);
                    if (group.empty()) continue;
                    write_clock.start();
                    jplace.write(group, num_threads);
                    write_clock.stop();
                }
            } catch (...) {
                writer_error = std::current_exception();
                to_write.close();
                to_place.close();
            }
        });
        std::vector<std::thread> placer_threads;
        for (size_t device = 0; device < n_devices; ++device) {
            placer_threads.emplace_back([&, device] {
                try {
                    std::vector<work_item> group;
                    while (to_place.pop_up_to(group, kGroupBatches)) {
                        const auto begin_group = std::chrono::steady_clock::now();
                        place_clocks[device].start();
                        std::vector<const std::vector<epik_amd::seq_record>*> batches;
                        for (const auto& item : group) batches.push_back(&item.batch);
                        auto placed = placer.place_flat(batches, device, num_threads);
                        place_clocks[device].stop();
                        auto ms_diff = (float)std::chrono::duration_cast<std::chrono::microseconds>(
                                           std::chrono::steady_clock::now() - begin_group).count() / 1000.0f;
                        if (ms_diff <= 0) ms_diff = 0.001f;
                        {
                            std::lock_guard<std::mutex> lock(stats_mutex);
                            for (const auto& item : group) num_seq_placed += item.batch.size();
                            // main.cpp:351-352: nominal batch size over the batch's time; a batch of a group
                            // takes its share of the group's time
                            average_speed += (double)group.size() * 1000.0 * (double)batch_size /
                                             ((double)ms_diff / (double)group.size());
                            num_iterations += group.size();
                        }
                        bool open = true;
                        for (size_t i = 0; i < group.size() && open; ++i) {
                            group[i].placed = std::move(placed[i]);
                            open = to_write.push(std::move(group[i]));
                        }
                        if (!open) break;
                    }
                } catch (...) {
                    placer_errors[device] = std::current_exception();
                    to_place.close();
                }
            });
        }
        for (auto& t : placer_threads) t.join();
        to_place.close();
        to_write.close();
        reader_thread.join();
        writer_thread.join();
        for (const auto& error : placer_errors)
            if (error) std::rethrow_exception(error);
        for (const auto& error : {reader_error, writer_error})
            if (error) std::rethrow_exception(error);
        double place_ms = 0;
        for (const auto& clock : place_clocks) place_ms += clock.ms();
        if (std::getenv("EPIK_AMD_STAGE_TIMES"))
            std::cout << "stage read " << read_clock.ms() << " ms\nstage place " << place_ms
                      << " ms\nstage write " << write_clock.ms() << " ms" << std::endl;
        jplace.end();
        if (num_iterations) average_speed /= (double)num_iterations;
        std::cout << std::endl
                  << "Placed " << num_seq_placed << " sequences.\nAverage speed: " << epik_amd::human_count(average_speed, false)
                  << " seq/s.\n";
        std::cout << "Output: " << jplace_filename << std::endl;
        const auto placement_end = std::chrono::steady_clock::now();
        const auto placement_time = (size_t)std::chrono::duration_cast<std::chrono::milliseconds>(placement_end - begin).count();
        if (std::getenv("EPIK_AMD_STAGE_TIMES"))  // (the reference's line below is in whole milliseconds)
            std::cout << "placement_time_us " << std::chrono::duration_cast<std::chrono::microseconds>(placement_end - begin).count()
                      << std::endl;
        std::cout << "Placement time: " << epik_amd::human_duration(placement_time) << " (" << placement_time << " ms)"
                  << std::endl;
        std::cout << "Done." << '\n' << std::flush;
    } catch (const std::runtime_error& error) {
        std::cerr << "Error: " << error.what() << std::endl;
        return -1;
    } catch (const std::exception& error) {  // std::stoul etc.
        std::cerr << "Error: " << error.what() << std::endl;
        return -1;
    }
    return 0;
}

#endif  // EPIK_AMD_NO_MAIN
