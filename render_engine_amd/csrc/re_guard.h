// re_guard.h -- the C ABI's promise that no C++ exception crosses it (include/re_hip.h: "no exceptions/aborts cross the ABI").
// Every extern "C" entry point of the library is a function-try-block closed by one of these handler lists: an exception thrown by
// the host code behind it (std::vector / std::map growth, std::bad_alloc, a length_error from a size that came out of a file or off the
// device) becomes a status code and a message in re_last_error instead of std::terminate -> abort() inside the caller's process.
#pragma once
#include <exception>
#include <new>
#include <string>
#include "re_hip.h"

namespace re {
// (noexcept: the handler itself must not throw -- building the message may hit the same out-of-memory condition)
inline void guard_message(std::string &dst, const char *entry, const char *what) noexcept {
    try { dst = entry; dst += ": "; dst += what; } catch (...) { }
}
}  // namespace re

#define RE_ABI_GUARD(CTX, ENTRY) \
    catch (const std::bad_alloc &) { if (CTX) re::guard_message((CTX)->err, ENTRY, "out of host memory (std::bad_alloc)"); return RE_E_CAPACITY; } \
    catch (const std::exception &e_) { if (CTX) re::guard_message((CTX)->err, ENTRY, e_.what()); return RE_E_STATE; } \
    catch (...) { if (CTX) re::guard_message((CTX)->err, ENTRY, "unknown C++ exception"); return RE_E_STATE; }
#define RE_ABI_GUARD_NOCTX(ERRSTR, ENTRY) \
    catch (const std::bad_alloc &) { re::guard_message(ERRSTR, ENTRY, "out of host memory (std::bad_alloc)"); return RE_E_CAPACITY; } \
    catch (const std::exception &e_) { re::guard_message(ERRSTR, ENTRY, e_.what()); return RE_E_STATE; } \
    catch (...) { re::guard_message(ERRSTR, ENTRY, "unknown C++ exception"); return RE_E_STATE; }
