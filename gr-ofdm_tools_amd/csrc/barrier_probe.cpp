// Host-only probe of the ABI's exception barrier (csrc/abi_barrier.h), built on the fly by
// tests/test_abi_cpu.py::test_exception_barrier_at_the_abi with g++ - NOT part of libofdmtools_hip.so.
//   g++ -std=c++17 -shared -fPIC -I../../include barrier_probe.cpp -o barrier_probe.so
// oth_probe_throw(kind): 0 std::bad_alloc, 1 std::runtime_error, 2 a non-std exception, 3 a real over-sized std::vector,
// anything else returns OTH_OK; oth_probe_last_error(): the text the barrier stored.
#include "../../include/ofdm_tools_hip.h"
#include "abi_barrier.h"

#include <stdexcept>
#include <string>
#include <vector>

namespace {
std::string g_text = "no error";
int fail_nothrow(void *, int code, const char *what) noexcept {
    try {
        g_text = what;
    } catch (...) {
    }
    return code;
}
}  // namespace

extern "C" {
int oth_probe_throw(int kind) {
    OTH_TRY
    if (kind == 0) throw std::bad_alloc();
    if (kind == 1) throw std::runtime_error("debug: runtime_error");
    if (kind == 2) throw 42;
    if (kind == 3) {
        std::vector<double> v;
        v.resize(v.max_size());          // std::length_error or std::bad_alloc, whichever the runtime raises first
        return (int)v.size();
    }
    return OTH_OK;
    OTH_CATCH((void *)nullptr)
}
const char *oth_probe_last_error(void) { return g_text.c_str(); }
}
