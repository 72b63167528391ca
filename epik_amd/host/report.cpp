#include "report.hpp"

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

namespace epik_amd {

std::string human_count(double value, bool integral)
{
    static const char* const kSuffix[] = {"", "K", "M", "B"};
    int unit = 0;
    double scaled = value;
    while (unit < 3 && scaled >= 1024.0) {
        scaled /= 1024.0;
        ++unit;
    }
    char text[64];
    if (unit == 0)
        std::snprintf(text, sizeof text, integral ? "%.0f" : "%f", value);
    else if (scaled == std::floor(scaled))
        std::snprintf(text, sizeof text, "%.0f%s", scaled, kSuffix[unit]);
    else
        std::snprintf(text, sizeof text, "%.1f%s", scaled, kSuffix[unit]);
    return text;
}

std::string human_duration(size_t milliseconds)
{
    const size_t total_seconds = milliseconds / 1000;
    const size_t days = total_seconds / 86400, hours = total_seconds / 3600 % 24;
    const size_t minutes = total_seconds / 60 % 60, seconds = total_seconds % 60;
    char text[96];
    int at = 0;
    if (days) at += std::snprintf(text + at, sizeof text - (size_t)at, "%zu day%s, ", days, days > 1 ? "s" : "");
    if (days || hours) at += std::snprintf(text + at, sizeof text - (size_t)at, "%02zu:", hours);
    std::snprintf(text + at, sizeof text - (size_t)at, "%02zu:%02zu", minutes, seconds);
    return text;
}

size_t parse_memory_size(const std::string& text)
{
    const char* begin = text.c_str();
    char* rest = nullptr;
    const double amount = std::strtod(begin, &rest);
    if (rest == begin || !(amount >= 0.0)) throw std::runtime_error("--max-ram: '" + text + "' does not start with a size");
    while (*rest && std::isspace((unsigned char)*rest)) ++rest;
    double factor = 1.0;
    switch (std::toupper((unsigned char)*rest)) {
        case '\0':
        case 'B': break;
        case 'K': factor = 1024.0; break;
        case 'M': factor = 1024.0 * 1024.0; break;
        case 'G': factor = 1024.0 * 1024.0 * 1024.0; break;
        default: throw std::runtime_error("--max-ram: unknown unit in '" + text + "' (B, K, M or G)");
    }
    return (size_t)(amount * factor);
}

void check_mu(float mu)
{
    if (!(mu >= 0.0f && mu <= 1.0f)) throw std::runtime_error("--mu must lie in [0, 1]");
}

}  // namespace epik_amd
