// jplace.hpp -- jplace v3 writer, the output side of the path.
// Mirrors epik::io::jplace_writer (reference epik/include/epik/jplace.h:16-56,
// epik/src/epik/jplace.cpp): start() writes the header, operator<< appends one object
// per unique sequence of a batch, end() closes the document.  No RapidJSON: the few JSON
// shapes are formatted directly (numbers with the shortest round-trip representation).
#ifndef EPIK_AMD_HOST_JPLACE_HPP
#define EPIK_AMD_HOST_JPLACE_HPP

#include <fstream>
#include <string>
#include <string_view>
#include <vector>

#include "placer.hpp"

namespace epik_amd::io {

class jplace_writer {
public:
    jplace_writer(const std::string& filename, const std::string& invocation, std::string_view newick_tree);
    void start();
    jplace_writer& operator<<(const impl::placed_collection& placed);
    /// the same, with the JSON text of the batch formatted by `num_threads` threads
    jplace_writer& write(const impl::placed_collection& placed, size_t num_threads);
    /// several batches, in order, as one piece of work for the formatting threads
    jplace_writer& write(const std::vector<const impl::placed_collection*>& group, size_t num_threads);
    void end();

private:
    std::string _filename;
    std::ofstream _out;
    std::string _invocation;
    std::string _tree;
    bool _first = true;
};

std::string json_escape(std::string_view s);
std::string json_double(double v);

}  // namespace epik_amd::io
#endif
