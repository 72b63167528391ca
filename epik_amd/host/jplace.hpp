// jplace.hpp -- jplace v3 writer, the output side of the path.
// Mirrors epik::io::jplace_writer (reference epik/include/epik/jplace.h:16-56,
// epik/src/epik/jplace.cpp): start() writes the header, operator<< appends one object
// per unique sequence of a batch, end() closes the document.  No RapidJSON: the few JSON
// shapes are formatted directly (numbers with the shortest round-trip representation).
#ifndef EPIK_AMD_HOST_JPLACE_HPP
#define EPIK_AMD_HOST_JPLACE_HPP

#include <cstdint>
#include <future>
#include <string>
#include <string_view>
#include <vector>

#include "placer.hpp"

namespace epik_amd::io {

class jplace_writer {
public:
    jplace_writer(const std::string& filename, const std::string& invocation, std::string_view newick_tree);
    void start();
    /// (optional) distal_length and pendant_length of a placement belong to its branch (place.cpp:435-437): with
    /// them known per branch, their text is made once per branch and not once per row
    void set_branch_lengths(const std::vector<double>& distal, const std::vector<double>& pendant);
    jplace_writer& operator<<(const impl::placed_collection& placed);
    /// the same, with the JSON text of the batch formatted by `num_threads` threads
    jplace_writer& write(const impl::placed_collection& placed, size_t num_threads);
    /// several batches, in order, as one piece of work for the formatting threads
    jplace_writer& write(const std::vector<const impl::placed_collection*>& group, size_t num_threads);
    /// ... in the driver's flat form (placer::place_flat)
    jplace_writer& write(const std::vector<const impl::placed_batch*>& group, size_t num_threads);
    void end();

    ~jplace_writer();
    jplace_writer(const jplace_writer&) = delete;
    jplace_writer& operator=(const jplace_writer&) = delete;

private:
    template <typename Batch>
    jplace_writer& write_group(const std::vector<const Batch*>& group, size_t num_threads);
    void append(const char* data, size_t n);  // at the end of the file, by the calling thread
    std::string _filename;
    int _fd = -1;
    uint64_t _size = 0;  // bytes written so far: where the next piece goes
    std::string _invocation;
    std::string _tree;
    bool _first = true;
    std::vector<std::string> _length_text;  // per branch: ", <distal>, <pendant>]"
    // one per formatting thread, kept from group to group (no fresh pages every time); two sets: the pieces of a group
    // go to the file (_flush) while the next group is formatted into the other set
    std::vector<std::string> _buffers[2];
    int _set = 0;
    std::future<void> _flush;
    void wait_for_flush();  // (throws what the write threw)
};

std::string json_escape(std::string_view s);
std::string json_double(double v);

}  // namespace epik_amd::io
#endif
