// spECKConfig.h -- the reference's spECKConfig (include/spECKConfig.h:8-53) over the C ABI.
// Same public fields: sm, maxStaticSharedMemoryPerBlock, maxDynamicSharedMemoryPerBlock, the six streams
// (std::vector<hipStream_t>, reference: CUstream) and the four events completeStart / completeEnd /
// individualStart / individualEnd (hipEvent_t, reference: cudaEvent_t); same initialize(device) / cleanup() pair.
// The handles are the ones the library created and uses (speck_config_handles): they stay owned by the opaque
// `handle`, cleanup() releases everything (on gfx950 both LDS limits read 65536 / 163840).
#pragma once
#include <hip/hip_runtime_api.h>

#include <stdexcept>
#include <vector>

#include "speck_c_api.h"

namespace spECK {
struct spECKConfig {
    int sm = 0;
    int maxStaticSharedMemoryPerBlock = 0;
    int maxDynamicSharedMemoryPerBlock = 0;
    std::vector<hipStream_t> streams;
    hipEvent_t completeStart = nullptr, completeEnd = nullptr, individualStart = nullptr, individualEnd = nullptr;
    speck_config* handle = nullptr;

    static spECKConfig initialize(int deviceNumber)
    {
        spECKConfig config;
        if (speck_config_create(deviceNumber, &config.handle) != SPECK_OK)
            throw std::runtime_error("spECKConfig::initialize: no such HIP device");
        speck_config_info(config.handle, &config.sm, &config.maxStaticSharedMemoryPerBlock,
                          &config.maxDynamicSharedMemoryPerBlock);
        void* s[6] = {};
        void* e[4] = {};
        speck_config_handles(config.handle, s, e);
        for (void* x : s) config.streams.push_back(static_cast<hipStream_t>(x));
        config.completeStart = static_cast<hipEvent_t>(e[0]);
        config.completeEnd = static_cast<hipEvent_t>(e[1]);
        config.individualStart = static_cast<hipEvent_t>(e[2]);
        config.individualEnd = static_cast<hipEvent_t>(e[3]);
        return config;
    }
    void cleanup()
    {
        if (handle) speck_config_destroy(handle);  // destroys the streams and events as well
        handle = nullptr;
        streams.clear();
        completeStart = completeEnd = individualStart = individualEnd = nullptr;
    }
    ~spECKConfig() {}  // as in the reference: the destructor does NOT clean up

private:
    spECKConfig() {}
};
}  // namespace spECK
