// guard_alloc.cpp -- TEST INFRASTRUCTURE (never linked into libawq_hip.so): a torch pluggable device allocator that gives
// every allocation its own virtual-memory mapping with UNMAPPED address space on both sides, and places the block so that
// it ENDS at the end of the mapping (AWQ_GUARD_ALLOC=end, default: a read or write one byte past an operand faults) or
// STARTS at its start (AWQ_GUARD_ALLOC=start: one byte before).  VERDICT r03 item 1(c): an over-read that lands in a
// neighbouring tensor of torch's caching allocator goes unnoticed; here it is a GPU memory access fault, deterministically.
//
//   hipMemAddressReserve(mapped + 2 * gran) -> [gran unmapped][mapped = roundup(size, gran)][gran unmapped]
//
// Blocks are pooled by exact size (hipGraph capture may not call the driver, and replays must see the same addresses).
// A virtual address range is NEVER reused for a different mapping: when the pool is over its budget the physical memory of the
// largest idle blocks is released but their reservation is kept (leaked).  The first version freed and re-reserved addresses;
// kernels then read stale data through the old translation at a re-used address (4 tests of 600 failed with zero-filled
// regions, only where > 256 MB temporaries had been unmapped just before: profiles/r04_fault_hunt/guard_first_run.txt).
// Build: hipcc -shared -fPIC -O2 -o libguard_alloc.so guard_alloc.cpp   (tests/guard/build.py)
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace {
struct Block {
    void* va;      // reservation base
    size_t va_bytes;
    void* map;     // mapped range
    size_t map_bytes;
    hipMemGenericAllocationHandle_t h;
    size_t size;   // requested bytes
    void* user;
};
std::mutex g_mu;
std::unordered_map<void*, Block> g_live;
std::unordered_map<size_t, std::vector<Block>> g_pool;
size_t g_gran = 0;
int g_place_end = -1;
long g_allocs = 0, g_driver_allocs = 0, g_released = 0;
size_t g_pooled_bytes = 0;
const size_t g_pool_budget = (size_t)96 << 30;  // idle physical memory kept mapped (the GPU has 288 GB)

void die(const char* what, hipError_t e) {
    std::fprintf(stderr, "guard_alloc: %s failed: %s\n", what, hipGetErrorString(e));
    std::abort();
}
#define CK(call)                         \
    do {                                 \
        hipError_t e_ = (call);          \
        if (e_ != hipSuccess) die(#call, e_); \
    } while (0)
}  // namespace

extern "C" {

__attribute__((visibility("default"))) void* guard_alloc(ssize_t size, int device, void* /*stream*/) {
    if (size <= 0) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_place_end < 0) {
        const char* m = std::getenv("AWQ_GUARD_ALLOC");
        g_place_end = (m && std::strcmp(m, "start") == 0) ? 0 : 1;
    }
    ++g_allocs;
    auto it = g_pool.find((size_t)size);
    if (it != g_pool.end() && !it->second.empty()) {
        Block b = it->second.back();
        it->second.pop_back();
        g_pooled_bytes -= b.map_bytes;
        g_live[b.user] = b;
        return b.user;
    }
    hipMemAllocationProp prop;
    std::memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    if (!g_gran) CK(hipMemGetAllocationGranularity(&g_gran, &prop, hipMemAllocationGranularityMinimum));
    Block b;
    b.size = (size_t)size;
    b.map_bytes = ((size_t)size + g_gran - 1) / g_gran * g_gran;
    b.va_bytes = b.map_bytes + 2 * g_gran;
    CK(hipMemAddressReserve(&b.va, b.va_bytes, g_gran, nullptr, 0));
    b.map = static_cast<char*>(b.va) + g_gran;
    CK(hipMemCreate(&b.h, b.map_bytes, &prop, 0));
    CK(hipMemMap(b.map, b.map_bytes, 0, b.h, 0));
    hipMemAccessDesc acc;
    std::memset(&acc, 0, sizeof(acc));
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(b.map, b.map_bytes, &acc, 1));
    const size_t sz16 = ((size_t)size + 15) & ~(size_t)15;  // the C ABI wants 16-byte aligned operands: up to 15 bytes of slack
    b.user = g_place_end ? static_cast<char*>(b.map) + (b.map_bytes - sz16) : b.map;
    g_live[b.user] = b;
    ++g_driver_allocs;
    return b.user;
}

__attribute__((visibility("default"))) void guard_free(void* ptr, size_t /*size*/, int /*device*/, void* /*stream*/) {
    if (!ptr) return;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_live.find(ptr);
    if (it == g_live.end()) {
        std::fprintf(stderr, "guard_alloc: free of unknown pointer %p\n", ptr);
        std::abort();
    }
    Block b = it->second;
    g_live.erase(it);
    g_pool[b.size].push_back(b);
    g_pooled_bytes += b.map_bytes;
    if (g_pooled_bytes <= g_pool_budget) return;
    (void)hipDeviceSynchronize();
    while (g_pooled_bytes > g_pool_budget / 2) {  // drop the largest idle blocks; their address ranges stay reserved for good
        size_t best = 0, best_bytes = 0;
        for (auto& kv : g_pool)
            if (!kv.second.empty() && kv.second.back().map_bytes > best_bytes) {
                best = kv.first;
                best_bytes = kv.second.back().map_bytes;
            }
        if (!best_bytes) break;
        Block d = g_pool[best].back();
        g_pool[best].pop_back();
        CK(hipMemUnmap(d.map, d.map_bytes));
        CK(hipMemRelease(d.h));
        g_pooled_bytes -= d.map_bytes;
        ++g_released;
    }
}

// (granularity, allocations served, allocations that went to the driver, placement: 1 = end, 0 = start)
__attribute__((visibility("default"))) void guard_stats(long* out4) {
    std::lock_guard<std::mutex> lk(g_mu);
    out4[0] = (long)g_gran;
    out4[1] = g_allocs;
    out4[2] = g_driver_allocs;
    out4[3] = g_place_end;
}
}
