// Developer helper for tools/buffer_speed_probe.py: a contiguous VIRTUAL range backed by physical chunks that are far apart.
//   striped_alloc(nbytes, chunk, spread): reserve nbytes of address space; create ceil(nbytes / chunk) * spread physical chunks
//   of `chunk` bytes back to back (the driver hands them out more or less sequentially), map every spread-th of them into the
//   range and release the others' mappings but KEEP their memory allocated (so that the used ones stay spread out).
// build: hipcc -O2 --offload-arch=gfx950 -shared -fPIC -o tools/libvmm_alloc.so tools/vmm_alloc.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern "C" void *striped_alloc(size_t nbytes, size_t chunk, int spread, int device) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) return nullptr;
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t n = (nbytes + chunk - 1) / chunk;
    void *base = nullptr;
    if (hipMemAddressReserve(&base, n * chunk, 0, nullptr, 0) != hipSuccess) return nullptr;
    std::vector<hipMemGenericAllocationHandle_t> pool(n * (size_t)spread);
    for (size_t i = 0; i < pool.size(); ++i)
        if (hipMemCreate(&pool[i], chunk, &prop, 0) != hipSuccess) { std::printf("hipMemCreate %zu failed\n", i); return nullptr; }
    for (size_t i = 0; i < n; ++i)
        if (hipMemMap((char *)base + i * chunk, chunk, 0, pool[i * spread], 0) != hipSuccess) { std::printf("hipMemMap failed\n"); return nullptr; }
    hipMemAccessDesc acc = {};
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(base, n * chunk, &acc, 1) != hipSuccess) { std::printf("hipMemSetAccess failed\n"); return nullptr; }
    return base;   // (the pool's handles are leaked on purpose: a probe)
}
