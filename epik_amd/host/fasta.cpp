#include "seq_record.hpp"

#include <cstring>
#include <stdexcept>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace epik_amd::io {

batch_fasta::batch_fasta(const std::string& filename, size_t batch_size) : _batch_size(batch_size ? batch_size : 1)
{
    _fd = ::open(filename.c_str(), O_RDONLY);
    if (_fd < 0) throw std::runtime_error("Cannot open file: " + filename);
    struct stat st;
    if (::fstat(_fd, &st) != 0 || st.st_size < 0) {
        ::close(_fd);
        throw std::runtime_error("Cannot open file: " + filename);
    }
    _size = (size_t)st.st_size;
    if (_size) {
        void* p = ::mmap(nullptr, _size, PROT_READ, MAP_PRIVATE, _fd, 0);
        if (p == MAP_FAILED) {
            ::close(_fd);
            throw std::runtime_error("Cannot map file: " + filename);
        }
        _data = static_cast<const char*>(p);
        (void)::madvise(p, _size, MADV_SEQUENTIAL);
    }
    _at = _data;
}

batch_fasta::~batch_fasta()
{
    if (_data) ::munmap(const_cast<char*>(_data), _size);
    if (_fd >= 0) ::close(_fd);
}

namespace {
// [line, end of line) without the '\n' and a trailing '\r'; `next` = the start of the line behind it
inline std::string_view take_line(const char* at, const char* end, const char*& next)
{
    const char* nl = static_cast<const char*>(std::memchr(at, '\n', (size_t)(end - at)));
    const char* stop = nl ? nl : end;
    next = nl ? nl + 1 : end;
    if (stop > at && stop[-1] == '\r') --stop;
    return {at, (size_t)(stop - at)};
}
}  // namespace

std::vector<seq_record> batch_fasta::next_batch()
{
    std::vector<seq_record> batch;
    batch.reserve(_batch_size);
    const char* const end = _data + _size;
    while (batch.size() < _batch_size && _at < end) {
        const char* next;
        std::string_view line = take_line(_at, end, next);
        if (line.empty() || line[0] != '>') {  // blank lines, and whatever stands in front of the first header
            _at = next;
            continue;
        }
        const std::string_view header = line.substr(1);
        _at = next;
        // the sequence: usually one line -- then it stays a view; more lines are joined
        std::string_view first;
        std::string joined;
        bool many = false;
        while (_at < end && *_at != '>') {
            line = take_line(_at, end, next);
            _at = next;
            if (line.empty()) continue;
            if (first.data() == nullptr && !many) {
                first = line;
            } else {
                if (!many) joined.assign(first), many = true;
                joined.append(line);
            }
        }
        batch.push_back(many ? seq_record::joined(header, std::move(joined)) : seq_record::view(header, first));
    }
    return batch;
}

}  // namespace epik_amd::io
