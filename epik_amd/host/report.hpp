// report.hpp -- the small text conventions of the driver's command line and report lines.
// Written from the reference driver's observable behaviour (epik/src/epik/main.cpp: the
// "Loaded ... of ... phylo-k-mers" / "Average speed" numbers :285-292, :368-376, the
// "Placement time" line :378-381, --max-ram :252-265, --mu :241), not from its code;
// host_test.cpp pins the outputs.
#ifndef EPIK_AMD_HOST_REPORT_HPP
#define EPIK_AMD_HOST_REPORT_HPP

#include <cstddef>
#include <string>

namespace epik_amd {

/// Counts as the driver prints them: below 1024 the number itself (six decimals for a fractional
/// type, as a fixed-notation stream prints it), from there on in units of 1024 with the suffixes
/// K, M, B -- "2K", "37.2M" -- one decimal unless the value is whole.
std::string human_count(double value, bool integral);
inline std::string human_count(size_t value) { return human_count((double)value, true); }

/// "[D day(s), ][HH:]MM:SS" of a duration in milliseconds (days and hours only when there are any).
std::string human_duration(size_t milliseconds);

/// --max-ram: a number and an optional unit of which only the first letter counts, case-insensitive
/// (B, K, M, G; powers of 1024; none = bytes): "128K", "50M", "4.2Gb".  Throws std::runtime_error.
size_t parse_memory_size(const std::string& text);

/// --mu is a fraction of the database.  Throws std::runtime_error outside [0, 1].
void check_mu(float mu);

}  // namespace epik_amd
#endif
