// guards.hpp -- canary zones around device allocations (debug option guard_bytes; guards.hip).
//
// The kernels of this library walk B before the input check has spoken and place rows by predictions they verify
// afterwards: the design RESTS on "every kernel stays inside its buffers whatever the inputs hold".  With the option on,
// every device buffer the library allocates -- C's three arrays, the scratch arena and the regions carved from it, the
// scratch and spill pools, the copies of row offsets and inputs -- carries N bytes of a fixed pattern on either side, and
// a check kernel at the end of every multiply reports a touched byte as SPECK_ERR_HIP (what was touched on stderr).
// (Role: the reference has no counterpart -- SURVEY.md 5, "Race detection / sanitizers": none.)
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <vector>

namespace speck {

constexpr unsigned char kGuardPattern = 0xA5;

struct GuardZone {
    const unsigned char* ptr;
    size_t len;
};

size_t guard_bytes();                      // width of a zone now (0: off), a multiple of 256
void set_guard_bytes(size_t n);            // applies to allocations made from now on
// hipMalloc / hipFree with zones when the option is on.  guarded_free takes pointers of either kind.
hipError_t guarded_malloc(void** p, size_t bytes);
hipError_t guarded_free(void* p);
// the two zones of a guarded allocation (nothing for a plain one) appended to `out`
void guard_zones_of(const void* user_ptr, std::vector<GuardZone>* out);
// fill zones with the pattern (on `s`) / count the zones that hold anything else (blocking); first_bad: its index
hipError_t guard_fill(const std::vector<GuardZone>& zones, hipStream_t s);
int guard_check(const std::vector<GuardZone>& zones, hipStream_t s, int* first_bad, size_t* first_bad_offset);

}  // namespace speck
