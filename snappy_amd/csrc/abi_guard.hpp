// Nothing may unwind through the C ABI (a P/Invoke caller would be terminated):
// every extern "C" entry point runs its body under this guard.  Failed
// allocations of the C++ runtime become SNAPPY_E_NOMEM, any other exception
// SNAPPY_E_INVALID_ARG.  Header-only and free of HIP, so that the host parsers
// build on their own.
#pragma once

#include <new>
#include <stdexcept>

#include "../../include/snappyhash.h"

namespace snappy {

inline int status_of_current_exception() noexcept
{
    try {
        throw;
    } catch (const std::bad_alloc &) {
        return SNAPPY_E_NOMEM;
    } catch (const std::length_error &) {
        return SNAPPY_E_NOMEM;  // a container asked for more than it can ever hold
    } catch (...) {
        return SNAPPY_E_INVALID_ARG;
    }
}

template <typename Fn>
int abi_guard(Fn &&fn) noexcept
{
    try {
        return fn();
    } catch (...) {
        return status_of_current_exception();
    }
}

}  // namespace snappy
