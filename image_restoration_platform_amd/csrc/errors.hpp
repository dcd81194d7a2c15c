// errors.hpp -- the engine's internal exception and the status codes of include/ire.h; no HIP, no device code: shared by the
// library proper (common.hpp) and by the host-only build of the batcher under the sanitizers (tests/native/batcher_stress.cpp).
#pragma once
#include <string>

#include "../../include/ire.h"

namespace ire {

// Internal exception; converted to an ire_status + thread-local message at the C ABI.
struct Error {
    int code;
    std::string msg;
};

[[noreturn]] inline void fail(int code, const std::string& msg) { throw Error{code, msg}; }

}  // namespace ire
