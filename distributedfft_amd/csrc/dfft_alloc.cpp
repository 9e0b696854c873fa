// dfft_alloc.cpp -- device memory for the plan's internal slabs (hand-over buffer, bufferDev1).
//
// The reference allocates its plan buffers with plain hipMalloc (fft_mpi_3d_api.cpp:56-82, fft_mpi_alloc_local_memory :216-230).
// Here the plan-owned slabs can also be built with the HIP virtual-memory API: one virtual range, physical memory created in
// chunks of a chosen size and mapped in order.  Why: the X pass reads N0 segments of 128 bytes one plane apart, so how a
// multi-GiB buffer is laid out physically (how large its contiguous pieces are, where they start) decides which translations
// and memory channels a tile touches -- profiles/r03/README.md has the measurements that led to the default.
//   DFFT_W_ALLOC=malloc          hipMalloc
//   DFFT_W_ALLOC=vmm[:chunk_MiB[:va_align_MiB]]   chunk 0 = one physical allocation for the whole buffer
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "dfft_internal.h"

namespace dfft {

namespace {
struct VmmBlock {
    size_t                                    bytes;  // reserved = mapped size
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<size_t>                       sizes;
};
std::mutex                 g_vmm_mutex;
std::map<void*, VmmBlock>  g_vmm_blocks;

void vmm_release(void* base, VmmBlock& b, size_t mapped) {
    size_t off = 0;
    for (size_t i = 0; i < b.handles.size(); ++i) {
        if (off < mapped) (void)hipMemUnmap((char*)base + off, b.sizes[i]);
        (void)hipMemRelease(b.handles[i]);
        off += b.sizes[i];
    }
    (void)hipMemAddressFree(base, b.bytes);
}

hipError_t vmm_alloc(void** out, size_t bytes, size_t chunk, size_t va_align) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipMemAllocationProp prop;
    std::memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    if (e != hipSuccess) return e;
    if (gran == 0) gran = 2u << 20;
    const size_t total = (bytes + gran - 1) / gran * gran;
    if (chunk == 0 || chunk > total) chunk = total;
    chunk = (chunk + gran - 1) / gran * gran;
    if (va_align < gran) va_align = gran;
    void* base = nullptr;
    e = hipMemAddressReserve(&base, total, va_align, nullptr, 0);
    if (e != hipSuccess) return e;
    VmmBlock b;
    b.bytes = total;
    size_t mapped = 0;
    for (size_t off = 0; off < total; off += chunk) {
        const size_t                    sz = std::min(chunk, total - off);
        hipMemGenericAllocationHandle_t h;
        e = hipMemCreate(&h, sz, &prop, 0);
        if (e != hipSuccess) break;
        b.handles.push_back(h);
        b.sizes.push_back(sz);
        e = hipMemMap((char*)base + off, sz, 0, h, 0);
        if (e != hipSuccess) break;
        mapped = off + sz;
    }
    if (e == hipSuccess) {
        hipMemAccessDesc acc;
        std::memset(&acc, 0, sizeof(acc));
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = dev;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(base, total, &acc, 1);
    }
    if (e != hipSuccess) {
        vmm_release(base, b, mapped);
        return e;
    }
    {
        std::lock_guard<std::mutex> lk(g_vmm_mutex);
        g_vmm_blocks[base] = std::move(b);
    }
    *out = base;
    return hipSuccess;
}
}  // namespace

hipError_t slab_alloc(void** p, size_t bytes) {
    const char* m = getenv("DFFT_W_ALLOC");
    if (m && !strncmp(m, "vmm", 3)) {
        size_t chunk_mb = 0, align_mb = 0;
        if (m[3] == ':') {
            char* end = nullptr;
            chunk_mb = strtoull(m + 4, &end, 10);
            if (end && *end == ':') align_mb = strtoull(end + 1, nullptr, 10);
        }
        return vmm_alloc(p, bytes ? bytes : 16, chunk_mb << 20, align_mb << 20);
    }
    return hipMalloc(p, bytes ? bytes : 16);
}

hipError_t slab_free(void* p) {
    if (!p) return hipSuccess;
    {
        std::lock_guard<std::mutex> lk(g_vmm_mutex);
        auto it = g_vmm_blocks.find(p);
        if (it != g_vmm_blocks.end()) {
            VmmBlock b = std::move(it->second);
            g_vmm_blocks.erase(it);
            (void)hipDeviceSynchronize();  // like hipFree: nothing may still be using the range
            vmm_release(p, b, b.bytes);
            return hipSuccess;
        }
    }
    return hipFree(p);
}

}  // namespace dfft
