#include "jplace.hpp"

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <stdexcept>
#include <thread>
#include <vector>

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
    : _filename(filename), _out(filename), _invocation(invocation), _tree(newick_tree)
{
    if (!_out) throw std::runtime_error("Could not create file " + filename);  // jplace.cpp:15-18
}

void jplace_writer::start()
{
    // jplace.cpp:40-59, 71-102: metadata, tree, version 3, fields, then the open array
    _out << "{\n    \"metadata\": {\"invocation\": \"" << json_escape(_invocation) << "\"},\n"
         << "    \"tree\": \"" << json_escape(_tree) << "\",\n"
         << "    \"version\": 3,\n"
         << "    \"fields\": [\"edge_num\", \"likelihood\", \"like_weight_ratio\", \"distal_length\", "
            "\"pendant_length\"],\n"
         << "    \"placements\": [";
    _out.flush();
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

/// The objects of placed_seqs[begin, end), joined with "," (no leading or trailing comma).
void format_objects(const impl::placed_collection& placed, size_t begin, size_t end, std::string& buffer)
{
    for (size_t i = begin; i < end; ++i) {
        const auto& placed_seq = placed.placed_seqs[i];
        buffer += i == begin ? "\n        {\n" : ",\n        {\n";
        buffer += "            \"p\": [";
        bool first_row = true;
        for (const auto& p : placed_seq.placements) {  // jplace.cpp:121-139; `count` is not written
            buffer += first_row ? "\n                [" : ",\n                [";
            first_row = false;
            append_uint(buffer, p.branch_id);
            buffer += ", ";
            append_double(buffer, (double)p.score);
            buffer += ", ";
            append_double(buffer, p.weight_ratio);
            buffer += ", ";
            append_double(buffer, p.distal_length);
            buffer += ", ";
            append_double(buffer, p.pendant_length);
            buffer += "]";
        }
        buffer += first_row ? "],\n" : "\n            ],\n";
        buffer += "            \"nm\": [";
        bool first_name = true;
        for (const auto header : placed.sequence_map.at(placed_seq.sequence)) {  // jplace.cpp:141-158
            buffer += first_name ? "\n                [\"" : ",\n                [\"";
            first_name = false;
            buffer += json_escape(header);
            buffer += "\", 1]";
        }
        buffer += first_name ? "]\n" : "\n            ]\n";
        buffer += "        }";
    }
}

}  // namespace

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
    // The objects of all batches of the group, in order, cut into one run per thread (about 2048
    // objects at least): the runs are formatted side by side and written one after the other.
    struct run {
        size_t batch, begin, end;  // objects [begin, end) of group[batch]
    };
    size_t total = 0;
    for (const auto* placed : group) total += placed->placed_seqs.size();
    if (total == 0) return *this;
    const size_t parts = std::max<size_t>(1, std::min(num_threads, total / 2048 + 1));
    const size_t per_part = (total + parts - 1) / parts;
    std::vector<std::vector<run>> work(parts);
    {
        size_t part = 0, room = per_part;
        for (size_t b = 0; b < group.size(); ++b) {
            size_t at = 0;
            const size_t n = group[b]->placed_seqs.size();
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
    std::vector<std::string> buffers(parts);
    auto format_part = [&](size_t part) {
        size_t objects = 0;
        for (const auto& r : work[part]) objects += r.end - r.begin;
        buffers[part].reserve(objects * 900);
        for (const auto& r : work[part]) {
            if (!buffers[part].empty()) buffers[part] += ",";
            format_objects(*group[r.batch], r.begin, r.end, buffers[part]);
        }
    };
    if (parts == 1) {
        format_part(0);
    } else {
        std::vector<std::thread> threads;
        for (size_t part = 1; part < parts; ++part) threads.emplace_back(format_part, part);
        format_part(0);
        for (auto& t : threads) t.join();
    }
    for (const auto& buffer : buffers) {
        if (buffer.empty()) continue;
        if (!_first) _out.put(',');
        _first = false;
        _out.write(buffer.data(), (std::streamsize)buffer.size());
    }
    _out.flush();
    return *this;
}

void jplace_writer::end()
{
    _out << (_first ? "]\n}\n" : "\n    ]\n}\n");  // jplace.cpp:61-69
    _out.close();
}

}  // namespace epik_amd::io
