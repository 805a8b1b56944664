// tools/ubench/vmm_alloc.hip: a device array whose physical memory is made of separately created granules mapped into one
// virtual range in a shuffled order (HIP virtual memory management) - to see whether "scattered pages" can be had on purpose
// (profiles/r3_oligo_placement.txt: arrays on scattered pages are the fast placement class of the oligo kernel).
//   hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ubench/libvmm_alloc.so tools/ubench/vmm_alloc.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

struct VmmArray {
    void *ptr;
    size_t size, chunk;
    std::vector<hipMemGenericAllocationHandle_t> handles;
};

extern "C" int vmm_alloc(size_t bytes, size_t chunk_bytes, int shuffle, void **out_ptr, void **out_handle) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess) return 1;
    size_t chunk = (chunk_bytes + gran - 1) / gran * gran;
    size_t size = (bytes + chunk - 1) / chunk * chunk;
    VmmArray *a = new VmmArray{nullptr, size, chunk, {}};
    if (hipMemAddressReserve(&a->ptr, size, 0, nullptr, 0) != hipSuccess) return 2;
    const size_t n = size / chunk;
    a->handles.resize(n);
    for (size_t i = 0; i < n; i++)
        if (hipMemCreate(&a->handles[i], chunk, &prop, 0) != hipSuccess) return 3;
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; i++) order[i] = i;
    if (shuffle) {
        uint64_t x = 0x9E3779B97F4A7C15ull;
        for (size_t i = n - 1; i > 0; i--) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            const size_t j = x % (i + 1);
            const size_t t = order[i]; order[i] = order[j]; order[j] = t;
        }
    }
    for (size_t i = 0; i < n; i++)
        if (hipMemMap((char *)a->ptr + i * chunk, chunk, 0, a->handles[order[i]], 0) != hipSuccess) return 4;
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = 0;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(a->ptr, size, &acc, 1) != hipSuccess) return 5;
    *out_ptr = a->ptr;
    *out_handle = a;
    printf("vmm_alloc: %zu bytes in %zu chunks of %zu (granularity %zu)%s\n", size, n, chunk, gran, shuffle ? ", shuffled" : "");
    return 0;
}

extern "C" int vmm_free(void *handle) {
    VmmArray *a = (VmmArray *)handle;
    (void)hipMemUnmap(a->ptr, a->size);
    for (auto h : a->handles) (void)hipMemRelease(h);
    (void)hipMemAddressFree(a->ptr, a->size);
    delete a;
    return 0;
}
