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

// ---- chunk pool: regions assembled from chosen physical chunks (probe for "is the speed a property of the chunks?") ----
static std::vector<hipMemGenericAllocationHandle_t> g_pool;
static size_t g_chunk = 0;
static hipMemAllocationProp g_prop = {};
extern "C" size_t pool_create(size_t n, size_t chunk, int device) {
    g_prop.type = hipMemAllocationTypePinned; g_prop.location.type = hipMemLocationTypeDevice; g_prop.location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &g_prop, hipMemAllocationGranularityRecommended) != hipSuccess) return 0;
    g_chunk = (chunk + gran - 1) / gran * gran;
    g_pool.resize(n);
    for (size_t i = 0; i < n; ++i) if (hipMemCreate(&g_pool[i], g_chunk, &g_prop, 0) != hipSuccess) { g_pool.resize(i); break; }
    return g_pool.size();
}
extern "C" size_t pool_chunk_bytes() { return g_chunk; }
// maps the chunks idx[0..n) back to back; returns the base address (0 on failure)
static size_t g_va_align = 0;
extern "C" void pool_set_va_alignment(size_t a) { g_va_align = a; }
extern "C" void *pool_map(const int *idx, int n) {
    void *base = nullptr;
    if (hipMemAddressReserve(&base, (size_t)n * g_chunk, g_va_align, nullptr, 0) != hipSuccess) return nullptr;
    for (int i = 0; i < n; ++i)
        if (hipMemMap((char *)base + (size_t)i * g_chunk, g_chunk, 0, g_pool[idx[i]], 0) != hipSuccess) return nullptr;
    hipMemAccessDesc acc = {};
    acc.location = g_prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(base, (size_t)n * g_chunk, &acc, 1) != hipSuccess) return nullptr;
    return base;
}
extern "C" int pool_unmap(void *base, int n) {
    if (hipMemUnmap(base, (size_t)n * g_chunk) != hipSuccess) return 1;
    return hipMemAddressFree(base, (size_t)n * g_chunk) != hipSuccess;
}
