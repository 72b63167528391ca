#include "seq_record.hpp"

#include <stdexcept>

namespace epik_amd::io {

batch_fasta::batch_fasta(const std::string& filename, size_t batch_size)
    : _in(filename, std::ios::binary), _batch_size(batch_size ? batch_size : 1)
{
    if (!_in) throw std::runtime_error("Cannot open file: " + filename);
}

std::vector<seq_record> batch_fasta::next_batch()
{
    std::vector<seq_record> batch;
    batch.reserve(_batch_size);
    std::string line, sequence;
    std::string header = _pending_header;
    bool in_record = _have_pending;
    _have_pending = false;
    while (batch.size() < _batch_size && std::getline(_in, line)) {
        _bytes_read += line.size() + 1;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '>') {
            if (in_record) {
                batch.emplace_back(std::move(header), std::move(sequence));
                sequence.clear();
            }
            header = line.substr(1);
            in_record = true;
            if (batch.size() == _batch_size) {  // keep this header for the next batch
                _pending_header = header;
                _have_pending = true;
                return batch;
            }
        } else if (in_record) {
            sequence += line;
        }
    }
    if (in_record && batch.size() < _batch_size) batch.emplace_back(std::move(header), std::move(sequence));
    return batch;
}

}  // namespace epik_amd::io
