// spECKConfig.h -- the reference's spECKConfig (include/spECKConfig.h:8-53) over the C ABI.
// Same public fields (sm, maxStaticSharedMemoryPerBlock, maxDynamicSharedMemoryPerBlock) and the
// same initialize(device) / cleanup() pair; the streams and events of the reference live inside
// the opaque handle (HIP streams / events; on gfx950 both LDS limits read 65536 / 163840).
#pragma once
#include <stdexcept>

#include "speck_c_api.h"

namespace spECK {
struct spECKConfig {
    int sm = 0;
    int maxStaticSharedMemoryPerBlock = 0;
    int maxDynamicSharedMemoryPerBlock = 0;
    speck_config* handle = nullptr;

    static spECKConfig initialize(int deviceNumber)
    {
        spECKConfig config;
        if (speck_config_create(deviceNumber, &config.handle) != SPECK_OK)
            throw std::runtime_error("spECKConfig::initialize: no such HIP device");
        speck_config_info(config.handle, &config.sm, &config.maxStaticSharedMemoryPerBlock,
                          &config.maxDynamicSharedMemoryPerBlock);
        return config;
    }
    void cleanup()
    {
        if (handle) speck_config_destroy(handle);
        handle = nullptr;
    }
    ~spECKConfig() {}  // as in the reference: the destructor does NOT clean up

private:
    spECKConfig() {}
};
}  // namespace spECK
