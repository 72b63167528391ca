#include "jplace.hpp"

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cerrno>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include <fcntl.h>
#include <unistd.h>

#include "parallel.hpp"

namespace epik_amd::io {

std::string json_escape(std::string_view s)
{
    std::string out;
    out.reserve(s.size() + 2);
    for (unsigned char c : s) {
        switch (c) {
            case '"': out += "\\\""; break;
            case '\\': out += "\\\\"; break;
            case '\n': out += "\\n"; break;
            case '\r': out += "\\r"; break;
            case '\t': out += "\\t"; break;
            default:
                if (c < 0x20) {
                    char buf[8];
                    std::snprintf(buf, sizeof buf, "\\u%04x", c);
                    out += buf;
                } else {
                    out.push_back((char)c);
                }
        }
    }
    return out;
}

std::string json_double(double v)
{
    if (!std::isfinite(v)) return "null";  // JSON has no inf/nan (RapidJSON refuses them too)
    char buf[40];
    const auto res = std::to_chars(buf, buf + sizeof buf, v);  // shortest round-trip form
    std::string out(buf, res.ptr);
    if (out.find_first_of(".eE") == std::string::npos) out += ".0";  // Writer::Double keeps a fraction
    return out;
}

jplace_writer::jplace_writer(const std::string& filename, const std::string& invocation,
                             std::string_view newick_tree)
    : _filename(filename), _invocation(invocation), _tree(newick_tree)
{
    _fd = ::open(filename.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (_fd < 0) throw std::runtime_error("Could not create file " + filename);  // jplace.cpp:15-18
}

jplace_writer::~jplace_writer()
{
    try {
        wait_for_flush();
    } catch (const std::exception& e) {  // (end() was not reached: an error is on its way up already -- this one is said, not lost)
        std::fprintf(stderr, "jplace: a write in flight failed while the writer was being torn down: %s\n", e.what());
    } catch (...) {
        std::fprintf(stderr, "jplace: a write in flight failed while the writer was being torn down\n");
    }
    if (_fd >= 0) ::close(_fd);
}

void jplace_writer::wait_for_flush()
{
    if (_flush.valid()) _flush.get();
}

namespace {
// all of [data, data + n) at `offset` of the file (several threads write their own pieces side by side)
void write_at(int fd, const char* data, size_t n, uint64_t offset, const std::string& filename)
{
    while (n) {
        const ssize_t done = ::pwrite(fd, data, n, (off_t)offset);
        if (done < 0) {
            if (errno == EINTR) continue;
            throw std::runtime_error("Could not write " + filename + ": " + std::strerror(errno));
        }
        data += done, n -= (size_t)done, offset += (uint64_t)done;
    }
}
}  // namespace

void jplace_writer::append(const char* data, size_t n)
{
    write_at(_fd, data, n, _size, _filename);
    _size += n;
}

void jplace_writer::start()
{
    // jplace.cpp:40-59, 71-102: metadata, tree, version 3, fields, then the open array
    std::string head = "{\n    \"metadata\": {\"invocation\": \"" + json_escape(_invocation) + "\"},\n" +
                       "    \"tree\": \"" + json_escape(_tree) + "\",\n" +
                       "    \"version\": 3,\n" +
                       "    \"fields\": [\"edge_num\", \"likelihood\", \"like_weight_ratio\", \"distal_length\", "
                       "\"pendant_length\"],\n" +
                       "    \"placements\": [";
    append(head.data(), head.size());
}

namespace {

void append_double(std::string& out, double v)
{
    if (!std::isfinite(v)) {  // JSON has no inf/nan (RapidJSON refuses them too)
        out += "null";
        return;
    }
    char buf[40];
    const auto res = std::to_chars(buf, buf + sizeof buf, v);  // shortest round-trip form
    const std::string_view text(buf, (size_t)(res.ptr - buf));
    out += text;
    if (text.find_first_of(".eE") == std::string_view::npos) out += ".0";  // Writer::Double keeps a fraction
}

void append_uint(std::string& out, uint64_t v)
{
    char buf[24];
    const auto res = std::to_chars(buf, buf + sizeof buf, v);
    out.append(buf, (size_t)(res.ptr - buf));
}

// json_escape() straight into the buffer; a header without anything to escape (nearly all) is one append
void append_escaped(std::string& out, std::string_view s)
{
    bool plain = true;
    for (unsigned char c : s)
        if (c < 0x20 || c == '"' || c == '\\') {
            plain = false;
            break;
        }
    if (plain)
        out.append(s.data(), s.size());
    else
        out += json_escape(s);
}

// one object of the "placements" array: the rows of a sequence and its names (jplace.cpp:104-158)
template <typename RowIt, typename NameIt>
void format_object(std::string& buffer, bool first, RowIt row, RowIt row_end, NameIt name, NameIt name_end,
                   const std::vector<std::string>* length_text)
{
    buffer += first ? "\n        {\n" : ",\n        {\n";
    buffer += "            \"p\": [";
    bool first_row = true;
    for (; row != row_end; ++row) {  // jplace.cpp:121-139; `count` is not written
        const auto& p = *row;
        buffer += first_row ? "\n                [" : ",\n                [";
        first_row = false;
        append_uint(buffer, p.branch_id);
        buffer += ", ";
        append_double(buffer, (double)p.score);
        buffer += ", ";
        append_double(buffer, p.weight_ratio);
        // the two lengths belong to the branch (place.cpp:435-437): their text is made once per branch
        if (length_text && p.count != 0 && p.branch_id < length_text->size()) {
            buffer += (*length_text)[p.branch_id];
        } else {
            buffer += ", ";
            append_double(buffer, p.distal_length);
            buffer += ", ";
            append_double(buffer, p.pendant_length);
            buffer += "]";
        }
    }
    buffer += first_row ? "],\n" : "\n            ],\n";
    buffer += "            \"nm\": [";
    bool first_name = true;
    for (; name != name_end; ++name) {  // jplace.cpp:141-158
        buffer += first_name ? "\n                [\"" : ",\n                [\"";
        first_name = false;
        append_escaped(buffer, *name);
        buffer += "\", 1]";
    }
    buffer += first_name ? "]\n" : "\n            ]\n";
    buffer += "        }";
}

/// The objects [begin, end) of a batch, joined with "," (no leading or trailing comma).
void format_objects(const impl::placed_collection& placed, size_t begin, size_t end, std::string& buffer,
                    const std::vector<std::string>* length_text)
{
    for (size_t i = begin; i < end; ++i) {
        const auto& placed_seq = placed.placed_seqs[i];
        const auto& names = placed.sequence_map.at(placed_seq.sequence);
        format_object(buffer, i == begin, placed_seq.placements.begin(), placed_seq.placements.end(), names.begin(), names.end(),
                      length_text);
    }
}
void format_objects(const impl::placed_batch& placed, size_t begin, size_t end, std::string& buffer,
                    const std::vector<std::string>* length_text)
{
    for (size_t u = begin; u < end; ++u)
        format_object(buffer, u == begin, placed.rows.begin() + placed.row_begin[u], placed.rows.begin() + placed.row_begin[u + 1],
                      placed.names.begin() + placed.name_begin[u], placed.names.begin() + placed.name_begin[u + 1], length_text);
}
inline size_t object_count(const impl::placed_collection& placed) { return placed.placed_seqs.size(); }
inline size_t object_count(const impl::placed_batch& placed) { return placed.size(); }

}  // namespace

void jplace_writer::set_branch_lengths(const std::vector<double>& distal, const std::vector<double>& pendant)
{
    _length_text.clear();
    _length_text.resize(std::min(distal.size(), pendant.size()));
    for (size_t b = 0; b < _length_text.size(); ++b) {
        std::string& text = _length_text[b];
        text = ", ";
        append_double(text, distal[b]);
        text += ", ";
        append_double(text, pendant[b]);
        text += "]";
    }
}

jplace_writer& jplace_writer::operator<<(const impl::placed_collection& placed)
{
    return write(placed, 1);
}

jplace_writer& jplace_writer::write(const impl::placed_collection& placed, size_t num_threads)
{
    return write(std::vector<const impl::placed_collection*>{&placed}, num_threads);
}

jplace_writer& jplace_writer::write(const std::vector<const impl::placed_collection*>& group, size_t num_threads)
{
    return write_group(group, num_threads);
}

jplace_writer& jplace_writer::write(const std::vector<const impl::placed_batch*>& group, size_t num_threads)
{
    return write_group(group, num_threads);
}

template <typename Batch>
jplace_writer& jplace_writer::write_group(const std::vector<const Batch*>& group, size_t num_threads)
{
    // The objects of all batches of the group, in order, cut into one run per thread (about 2048
    // objects at least): the runs are formatted side by side and written side by side.
    struct run {
        size_t batch, begin, end;  // objects [begin, end) of group[batch]
    };
    size_t total = 0;
    for (const auto* placed : group) total += object_count(*placed);
    if (total == 0) return *this;
    const size_t parts = std::max<size_t>(1, std::min(num_threads, total / 2048 + 1));
    const size_t per_part = (total + parts - 1) / parts;
    std::vector<std::vector<run>> work(parts);
    {
        size_t part = 0, room = per_part;
        for (size_t b = 0; b < group.size(); ++b) {
            size_t at = 0;
            const size_t n = object_count(*group[b]);
            while (at < n) {
                const size_t take = std::min(room, n - at);
                work[part].push_back({b, at, at + take});
                at += take;
                room -= take;
                if (room == 0 && part + 1 < parts) {
                    ++part;
                    room = per_part;
                }
            }
        }
    }
    // Every part formats its runs into a buffer of its own, then -- the sizes known -- writes it at its place in
    // the file: the pieces of a group go out side by side (pwrite), not one after the other through one thread.
    // ... while the pieces of the group before this one are still on their way to the file: buffered writes to one
    // file pass through its inode one after the other to a good part (16 threads: 873 MB in 90 ms), and the
    // formatting threads of the next group have the cores meanwhile.
    std::vector<std::string>& buffers = _buffers[_set];
    if (buffers.size() < parts) buffers.resize(parts);
    const auto t0 = std::chrono::steady_clock::now();
    parallel_for(parts, parts, [&](size_t part) {
        // (a string of the thread's own while it grows: the size fields of neighbouring strings share cache
        // lines, and every append writes one -- eight threads formatting into buffers[] directly ran no faster
        // than one)
        std::string buffer = std::move(buffers[part]);
        buffer.clear();
        size_t objects = 0;
        for (const auto& r : work[part]) objects += r.end - r.begin;
        if (buffer.capacity() < objects * 900) buffer.reserve(objects * 900);
        if (part != 0 || !_first) buffer += ",";  // (total != 0: every part holds at least one object)
        bool first_run = true;
        for (const auto& r : work[part]) {
            if (!first_run) buffer += ",";
            first_run = false;
            format_objects(*group[r.batch], r.begin, r.end, buffer, _length_text.empty() ? nullptr : &_length_text);
        }
        buffers[part] = std::move(buffer);
    });
    const auto t1 = std::chrono::steady_clock::now();
    wait_for_flush();  // (the other set's pieces: at most one group is on its way)
    std::vector<uint64_t> at(parts + 1, _size);
    for (size_t part = 0; part < parts; ++part) at[part + 1] = at[part] + buffers[part].size();
    const bool times = std::getenv("EPIK_AMD_WRITE_TIMES") != nullptr;
    const double format_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    _flush = std::async(std::launch::async, [this, &buffers, at, parts, total, times, format_ms] {
        const auto t2 = std::chrono::steady_clock::now();
        parallel_for(parts, parts, [&](size_t part) {
            write_at(_fd, buffers[part].data(), buffers[part].size(), at[part], _filename);
        });
        if (times)
            std::fprintf(stderr, "write: %zu objects in %zu parts: format %.1f ms, pwrite %.1f ms (%.1f MB)\n", total, parts, format_ms,
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2).count(),
                         (double)(at[parts] - at[0]) / 1e6);
    });
    _set ^= 1;
    _size = at[parts];
    _first = false;
    return *this;
}

void jplace_writer::end()
{
    wait_for_flush();
    const std::string tail = _first ? "]\n}\n" : "\n    ]\n}\n";  // jplace.cpp:61-69
    append(tail.data(), tail.size());
    if (::close(_fd) != 0) {
        _fd = -1;
        throw std::runtime_error("Could not close " + _filename);
    }
    _fd = -1;
}

}  // namespace epik_amd::io
