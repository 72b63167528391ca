#include "jplace.hpp"

#include <charconv>
#include <cmath>
#include <cstdio>
#include <stdexcept>

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

jplace_writer& jplace_writer::operator<<(const impl::placed_collection& placed)
{
    std::string buffer;
    for (const auto& placed_seq : placed.placed_seqs) {
        buffer += _first ? "\n        {\n" : ",\n        {\n";
        _first = false;
        buffer += "            \"p\": [";
        bool first_row = true;
        for (const auto& p : placed_seq.placements) {  // jplace.cpp:121-139; `count` is not written
            buffer += first_row ? "\n                [" : ",\n                [";
            first_row = false;
            buffer += std::to_string(p.branch_id);
            buffer += ", " + json_double((double)p.score);
            buffer += ", " + json_double(p.weight_ratio);
            buffer += ", " + json_double(p.distal_length);
            buffer += ", " + json_double(p.pendant_length);
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
    _out << buffer;
    _out.flush();
    return *this;
}

void jplace_writer::end()
{
    _out << (_first ? "]\n}\n" : "\n    ]\n}\n");  // jplace.cpp:61-69
    _out.close();
}

}  // namespace epik_amd::io
