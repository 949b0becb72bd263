// guards.hip -- see guards.hpp
#include "guards.hpp"

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <unordered_map>

namespace speck {
namespace {

struct Guarded {
    void* base;
    size_t bytes, n;
};
std::mutex g_mu;
std::unordered_map<const void*, Guarded> g_reg;
// (SPECK_GUARD_BYTES in the environment switches the zones on for a whole process -- the way the test suite and the
//  stress runs are repeated with them, scripts/stress_round.sh)
size_t env_guard()
{
    const char* e = std::getenv("SPECK_GUARD_BYTES");
    const long long v = e ? std::atoll(e) : 0;
    return v > 0 ? ((size_t)v + 255) & ~size_t(255) : 0;
}
size_t g_guard = env_guard();

__global__ __launch_bounds__(256) void guard_check_kernel(const GuardZone* __restrict__ zones, int n, unsigned long long* __restrict__ bad)
{
    // bad[0]: zones touched; bad[1]: (zone << 40 | offset) of the first touched byte seen
    for (int z = blockIdx.x; z < n; z += gridDim.x) {
        const unsigned char* p = zones[z].ptr;
        const size_t len = zones[z].len;
        size_t at = ~size_t(0);
        for (size_t i = threadIdx.x; i < len; i += 256)
            if (p[i] != kGuardPattern && i < at) at = i;
        if (at != ~size_t(0)) {
            if (atomicMin(&bad[2 + z], (unsigned long long)at) == ~0ull) atomicAdd(&bad[0], 1ull);
        }
    }
}

}  // namespace

size_t guard_bytes() { return g_guard; }
void set_guard_bytes(size_t n) { g_guard = (n + 255) & ~size_t(255); }

hipError_t guarded_malloc(void** p, size_t bytes)
{
    const size_t n = g_guard;
    if (!n) return hipMalloc(p, bytes);
    void* base = nullptr;
    const size_t user = (bytes + 255) & ~size_t(255);
    hipError_t e = hipMalloc(&base, user + 2 * n);
    if (e != hipSuccess) return e;
    unsigned char* b = static_cast<unsigned char*>(base);
    // (the tail zone starts right behind the LAST BYTE the caller asked for: an overrun by one entry is seen)
    e = hipMemset(b, kGuardPattern, n);
    if (e == hipSuccess) e = hipMemset(b + n + bytes, kGuardPattern, user - bytes + n);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipFree(base);
        return e;
    }
    *p = b + n;
    std::lock_guard<std::mutex> lk(g_mu);
    g_reg[*p] = Guarded{base, bytes, n};
    return hipSuccess;
}

hipError_t guarded_free(void* p)
{
    if (!p) return hipSuccess;
    void* base = p;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_reg.find(p);
        if (it != g_reg.end()) {
            base = it->second.base;
            g_reg.erase(it);
        }
    }
    return hipFree(base);
}

void guard_zones_of(const void* user_ptr, std::vector<GuardZone>* out)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_reg.find(user_ptr);
    if (it == g_reg.end()) return;
    const Guarded& g = it->second;
    const unsigned char* b = static_cast<const unsigned char*>(g.base);
    const size_t user = (g.bytes + 255) & ~size_t(255);
    out->push_back(GuardZone{b, g.n});
    out->push_back(GuardZone{b + g.n + g.bytes, user - g.bytes + g.n});
}

hipError_t guard_fill(const std::vector<GuardZone>& zones, hipStream_t s)
{
    for (const auto& z : zones) {
        hipError_t e = hipMemsetAsync(const_cast<unsigned char*>(z.ptr), kGuardPattern, z.len, s);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

int guard_check(const std::vector<GuardZone>& zones, hipStream_t s, int* first_bad, size_t* first_bad_offset)
{
    if (zones.empty()) return 0;
    const int n = (int)zones.size();
    GuardZone* d_z = nullptr;
    unsigned long long* d_bad = nullptr;
    std::vector<unsigned long long> h_bad(2 + n, ~0ull);
    h_bad[0] = h_bad[1] = 0;
    if (hipMalloc(reinterpret_cast<void**>(&d_z), n * sizeof(GuardZone)) != hipSuccess) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&d_bad), h_bad.size() * 8) != hipSuccess) {
        (void)hipFree(d_z);
        return -1;
    }
    int rc = -1;
    if (hipStreamSynchronize(s) == hipSuccess &&
        hipMemcpy(d_z, zones.data(), n * sizeof(GuardZone), hipMemcpyHostToDevice) == hipSuccess &&
        hipMemcpy(d_bad, h_bad.data(), h_bad.size() * 8, hipMemcpyHostToDevice) == hipSuccess) {
        hipLaunchKernelGGL(guard_check_kernel, dim3(n < 1024 ? n : 1024), dim3(256), 0, nullptr, d_z, n, d_bad);
        if (hipMemcpy(h_bad.data(), d_bad, h_bad.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            rc = (int)h_bad[0];
            if (rc > 0)
                for (int z = 0; z < n; ++z)
                    if (h_bad[2 + z] != ~0ull) {
                        if (first_bad) *first_bad = z;
                        if (first_bad_offset) *first_bad_offset = (size_t)h_bad[2 + z];
                        break;
                    }
        }
    }
    (void)hipFree(d_z);
    (void)hipFree(d_bad);
    return rc;
}

}  // namespace speck
