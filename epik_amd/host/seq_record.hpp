// seq_record.hpp / fasta -- the FASTA side of the path's callers.
//
// Mirrors the i2l surface EPIK uses (absent submodule; contract from the call sites):
//   i2l::seq_record::{header(), sequence()}                      place.cpp:78, jplace.cpp:152
//   i2l::io::batch_fasta(file, batch).next_batch() / bytes_read() main.cpp:332-358
// ASSUMPTIONS (SURVEY.md 8c register): header = the whole line after '>', multi-line
// sequences are concatenated, blank lines and '\r' are ignored.
#ifndef EPIK_AMD_HOST_SEQ_RECORD_HPP
#define EPIK_AMD_HOST_SEQ_RECORD_HPP

#include <cstddef>
#include <fstream>
#include <string>
#include <string_view>
#include <vector>

namespace epik_amd {

class seq_record {
public:
    seq_record() = default;
    seq_record(std::string header, std::string sequence)
        : _header(std::move(header)), _sequence(std::move(sequence)) {}
    std::string_view header() const noexcept { return _header; }
    std::string_view sequence() const noexcept { return _sequence; }

private:
    std::string _header;    // NUL-terminated storage, outlives the batch (jplace.cpp:152)
    std::string _sequence;
};

namespace io {

/// Reads a FASTA file batch by batch; an empty batch means end of file (main.cpp:336-340).
class batch_fasta {
public:
    batch_fasta(const std::string& filename, size_t batch_size);
    std::vector<seq_record> next_batch();
    size_t bytes_read() const noexcept { return _bytes_read; }

private:
    std::ifstream _in;
    size_t _batch_size;
    size_t _bytes_read = 0;
    std::string _pending_header;
    bool _have_pending = false;
};

}  // namespace io
}  // namespace epik_amd
#endif
