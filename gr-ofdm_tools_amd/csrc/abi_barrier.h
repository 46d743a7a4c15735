// The exception barrier of the C ABI (include/ofdm_tools_hip.h: "nothing throws or aborts").  Every extern "C" body of
// api.hip sits between OTH_TRY and OTH_CATCH(context): a std::bad_alloc (std::vector / std::string growth), a
// std::system_error (the context's recursive mutex) or anything else a C++ runtime call may raise becomes an error code
// + last-error text instead of std::terminate() inside the host's ctypes call.  The includer provides
//     int fail_nothrow(<context type> *, int code, const char *what) noexcept
// which stores the text without throwing.  csrc/barrier_probe.cpp builds the same macros into a tiny host-only library
// whose one entry point raises on request (tests/test_abi_cpu.py::test_exception_barrier_at_the_abi): the product
// library carries no such hook.
#pragma once
#include <exception>
#include <new>

#define OTH_TRY try {
#define OTH_CATCH(ctxexpr)                                                                                     \
    }                                                                                                          \
    catch (const std::bad_alloc &) { return fail_nothrow((ctxexpr), OTH_ERR_NOMEM, "out of host memory"); }    \
    catch (const std::exception &e_) { return fail_nothrow((ctxexpr), OTH_ERR_INTERNAL, e_.what()); }          \
    catch (...) { return fail_nothrow((ctxexpr), OTH_ERR_INTERNAL, "unknown C++ exception"); }
