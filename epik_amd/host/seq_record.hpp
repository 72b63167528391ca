// seq_record.hpp / fasta -- the FASTA side of the path's callers.
//
// Mirrors the i2l surface EPIK uses (absent submodule; contract from the call sites):
//   i2l::seq_record::{header(), sequence()}                      place.cpp:78, jplace.cpp:152
//   i2l::io::batch_fasta(file, batch).next_batch() / bytes_read() main.cpp:332-358
// ASSUMPTIONS (SURVEY.md 8c register): header = the whole line after '>', multi-line
// sequences are concatenated, blank lines and '\r' are ignored.
//
// The file is MAPPED, not read: a record is two views into the mapping (a sequence that spans several
// lines -- or carries '\r' -- is joined into storage of its own), so reading a read costs two memchr and no
// allocation.  The mapping lives as long as the reader: keep the reader until the batches are written.
#ifndef EPIK_AMD_HOST_SEQ_RECORD_HPP
#define EPIK_AMD_HOST_SEQ_RECORD_HPP

#include <cstddef>
#include <memory>
#include <string>
#include <string_view>
#include <vector>

namespace epik_amd {

class seq_record {
public:
    seq_record() = default;
    /// owning (tests, callers with strings of their own)
    seq_record(std::string header, std::string sequence) : _owned(std::make_unique<std::string[]>(2)), _n_owned(2)
    {
        _owned[0] = std::move(header);
        _owned[1] = std::move(sequence);
        _header = _owned[0];
        _sequence = _owned[1];
    }
    /// views into memory that outlives the record (the reader's mapping)
    static seq_record view(std::string_view header, std::string_view sequence)
    {
        seq_record r;
        r._header = header;
        r._sequence = sequence;
        return r;
    }
    /// ... with a sequence put together from several lines, owned here
    static seq_record joined(std::string_view header, std::string&& sequence)
    {
        seq_record r;
        r._owned = std::make_unique<std::string[]>(1);
        r._n_owned = 1;
        r._owned[0] = std::move(sequence);
        r._header = header;
        r._sequence = r._owned[0];
        return r;
    }
    seq_record(seq_record&&) noexcept = default;
    seq_record& operator=(seq_record&&) noexcept = default;
    seq_record(const seq_record& o) { *this = o; }
    seq_record& operator=(const seq_record& o)
    {
        if (this == &o) return *this;
        _n_owned = o._n_owned;
        _owned = _n_owned ? std::make_unique<std::string[]>(_n_owned) : nullptr;
        for (unsigned i = 0; i < _n_owned; ++i) _owned[i] = o._owned[i];
        _header = _n_owned == 2 ? std::string_view(_owned[0]) : o._header;
        _sequence = _n_owned ? std::string_view(_owned[_n_owned - 1]) : o._sequence;
        return *this;
    }
    std::string_view header() const noexcept { return _header; }
    std::string_view sequence() const noexcept { return _sequence; }

private:
    // (the views stay valid when the record moves: the strings are on the heap.)  2: header, sequence; 1: sequence
    std::unique_ptr<std::string[]> _owned;
    unsigned _n_owned = 0;
    std::string_view _header, _sequence;
};

namespace io {

/// Reads a FASTA file batch by batch; an empty batch means end of file (main.cpp:336-340).
class batch_fasta {
public:
    batch_fasta(const std::string& filename, size_t batch_size);
    ~batch_fasta();
    batch_fasta(const batch_fasta&) = delete;
    batch_fasta& operator=(const batch_fasta&) = delete;
    std::vector<seq_record> next_batch();
    size_t bytes_read() const noexcept { return (size_t)(_at - _data); }

private:
    int _fd = -1;
    const char* _data = nullptr;
    size_t _size = 0;
    const char* _at = nullptr;  // the next unread byte (always at the start of a line)
    size_t _batch_size;
};

}  // namespace io
}  // namespace epik_amd
#endif
