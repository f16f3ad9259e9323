// TEST INFRASTRUCTURE ONLY — stand-in for the reference's include/core/logger.hpp (spdlog + <format>, neither of which this image
// has) so that its COLMAP reader (src/loader/formats/colmap.cpp) compiles unmodified from where it lies: the LOG_* macros swallow
// their arguments, and std::format — used by the reader for three messages only — is a stub that returns the format string.
#pragma once
#include <string>
#include <string_view>

#define LOG_TRACE(...) ((void)0)
#define LOG_DEBUG(...) ((void)0)
#define LOG_INFO(...) ((void)0)
#define LOG_WARN(...) ((void)0)
#define LOG_ERROR(...) ((void)0)
#define LOG_CRITICAL(...) ((void)0)
#define LOG_PERF(...) ((void)0)
#define LOG_TIMER(...) ((void)0)
#define LOG_TIMER_TRACE(...) ((void)0)
#define LOG_TIMER_DEBUG(...) ((void)0)

#if !__has_include(<format>)
namespace std {
    template <class... Args>
    inline string format(string_view fmt, Args&&...) { return string(fmt); }
}  // namespace std
#else
#include <format>
#endif
