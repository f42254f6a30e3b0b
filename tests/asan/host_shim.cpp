// TEST INFRASTRUCTURE - not part of the product.  The host-only part of the C ABI (readmap.inc, rowsbatch.inc, bedtext.inc: the
// alignment walk, the row builder of a worker batch, the BED formatter) compiled by gcc with -fsanitize=address,undefined, so that
// tests/test_asan_host.py can drive it with valid and with corrupted container tables and see every out-of-bounds access.
// The sources are the product's own files, included where they lie; only the error plumbing of deepmod_hip.hip is restated here.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <thread>
#include <vector>

#include "../../include/deepmod_hip.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

}  // namespace

extern "C" const char* dm_last_error(void) { return g_err.c_str(); }

#include "../../deepmod_amd/csrc/readmap.inc"
#include "../../deepmod_amd/csrc/rowsbatch.inc"
#include "../../deepmod_amd/csrc/bedtext.inc"
