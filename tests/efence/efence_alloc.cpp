// GPU "electric fence" allocator for torch (test infrastructure, never part of the product path).
//
// torch.cuda.memory.CUDAPluggableAllocator(libefence.so, "efence_malloc", "efence_free") makes EVERY device allocation of
// the process its own virtual-memory mapping whose END coincides with the end of the mapped pages: the address range behind
// it (and in front of it) is reserved and never mapped, so a kernel that reads or writes past the end of ANY tensor - an
// input slot, the workspace, the flat parameter buffer - takes a memory-access fault at once, on every box, instead of one
// run in four (torch's caching allocator packs tensors into 2 MB / 20 MB segments: an overrun usually lands in a
// neighbour).  New mappings are filled with a poison byte (EFENCE_FILL, default 0xFF: NaN as float, -1 as int32) so that
// reads of memory nobody wrote show up in the results.
//
//   EFENCE_ALIGN   alignment of the returned address (default 16; 4 puts every float tensor flush against the fence)
//   EFENCE_FILL    poison byte (default 255), -1 = leave the fresh pages alone
//   EFENCE_LOG    path of an allocation log (the fault message names an address, the log names the tensor)
//   EFENCE_REUSE_VA 1: give freed address ranges back (default: keep them reserved, a stale pointer faults forever)
//   EFENCE_FRONT   1: align the allocation to the START of its mapping instead (catches under-runs)
//
// Build: hipcc -O2 -fPIC -shared tests/efence/efence_alloc.cpp -o tests/efence/libefence.so   (tests/efence/build.py)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace {
struct Rec { void* va; size_t reserved; void* map_at; size_t mapped; hipMemGenericAllocationHandle_t handle; };
std::mutex g_mu;
std::unordered_map<void*, Rec> g_live;
size_t g_gran = 0;
// While a hipGraph is being captured nothing may wait for the device (hipDeviceSynchronize inside a capture ends the
// process): tests/efence/__init__.py brackets torch.cuda.graph with efence_capture(1) / (0), frees that arrive in between
// (Python's garbage collector runs when it likes) are parked and carried out by the first free behind the capture.
std::atomic<int> g_capturing{0};
std::vector<void*> g_parked;

size_t env_size(const char* name, long dflt) { const char* s = getenv(name); return (size_t)(s ? atol(s) : dflt); }

// EFENCE_LOG=<path>: one line per allocation ("A <ptr> <bytes> end <first unmapped address>") and per free ("F <ptr>"): the
// runtime's fault message names a page address only, tests/efence/whose.py finds the tensor in front of it
FILE* log_file() {
    static FILE* f = nullptr;
    static bool tried = false;
    if (!tried) { tried = true; if (const char* p = getenv("EFENCE_LOG")) f = fopen(p, "a"); }
    return f;
}

void die(const char* what, hipError_t e) {
    fprintf(stderr, "[efence] %s failed: %s\n", what, hipGetErrorString(e));
    abort();
}
}  // namespace

extern "C" void* efence_malloc(ssize_t size, int device, hipStream_t stream) {
    (void)stream;
    if (size <= 0) size = 1;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    hipError_t e;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_gran) {
        if ((e = hipMemGetAllocationGranularity(&g_gran, &prop, hipMemAllocationGranularityMinimum)) != hipSuccess) die("granularity", e);
        if (g_gran < 4096) g_gran = 4096;
    }
    const size_t align = env_size("EFENCE_ALIGN", 16);
    const long fill = getenv("EFENCE_FILL") ? atol(getenv("EFENCE_FILL")) : 255;
    const bool front = env_size("EFENCE_FRONT", 0) != 0;
    const size_t need = ((size_t)size + align - 1) / align * align;
    const size_t mapped = (need + g_gran - 1) / g_gran * g_gran;
    const size_t reserved = mapped + 2 * g_gran;            // one unmapped granule on either side
    void* va = nullptr;
    if ((e = hipMemAddressReserve(&va, reserved, g_gran, nullptr, 0)) != hipSuccess) die("hipMemAddressReserve", e);
    Rec r{};
    r.va = va; r.reserved = reserved; r.map_at = static_cast<char*>(va) + g_gran; r.mapped = mapped;
    if ((e = hipMemCreate(&r.handle, mapped, &prop, 0)) != hipSuccess) die("hipMemCreate", e);
    if ((e = hipMemMap(r.map_at, mapped, 0, r.handle, 0)) != hipSuccess) die("hipMemMap", e);
    hipMemAccessDesc desc = {};
    desc.location = prop.location;
    desc.flags = hipMemAccessFlagsProtReadWrite;
    if ((e = hipMemSetAccess(r.map_at, mapped, &desc, 1)) != hipSuccess) die("hipMemSetAccess", e);
    if (fill >= 0 && (e = hipMemset(r.map_at, (int)fill, mapped)) != hipSuccess) die("hipMemset", e);
    void* user = front ? r.map_at : static_cast<char*>(r.map_at) + (mapped - need);
    g_live.emplace(user, r);
    if (FILE* f = log_file()) { fprintf(f, "A %p %zd end %p\n", user, (ssize_t)size, (void*)(static_cast<char*>(r.map_at) + mapped)); fflush(f); }
    return user;
}

static void release_now(void* ptr);

extern "C" void efence_capture(int on) {
    g_capturing.store(on ? 1 : 0);
    if (getenv("EFENCE_VERBOSE")) fprintf(stderr, "[efence] capture %d\n", on);
}

extern "C" void efence_free(void* ptr, ssize_t size, int device, hipStream_t stream) {
    (void)size; (void)device; (void)stream;
    if (!ptr) return;
    std::vector<void*> due;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_capturing.load()) {
            g_parked.push_back(ptr);
            if (getenv("EFENCE_VERBOSE")) fprintf(stderr, "[efence] free of %p parked (graph capture)\n", ptr);
            return;
        }
        due.swap(g_parked);
    }
    // torch frees a tensor the moment its last reference goes, kernels that use it may still be queued (its own caching
    // allocator relies on stream order for that): wait for the device before the pages disappear.
    (void)hipDeviceSynchronize();
    for (void* q : due) release_now(q);
    release_now(ptr);
}

static void release_now(void* ptr) {
    Rec r;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_live.find(ptr);
        if (it == g_live.end()) { fprintf(stderr, "[efence] free of an unknown pointer %p\n", ptr); return; }
        r = it->second;
        g_live.erase(it);
        if (FILE* f = log_file()) { fprintf(f, "F %p\n", ptr); fflush(f); }
    }
    hipError_t e;
    if ((e = hipMemUnmap(r.map_at, r.mapped)) != hipSuccess) die("hipMemUnmap", e);
    if ((e = hipMemRelease(r.handle)) != hipSuccess) die("hipMemRelease", e);
    // The address range is NOT handed back by default (EFENCE_REUSE_VA=1 does): a stale pointer into a freed tensor then
    // faults for the rest of the process instead of landing in whatever tensor is mapped there next - use-after-free
    // detection - and a new mapping never inherits translations of an old one.
    static const bool reuse = env_size("EFENCE_REUSE_VA", 0) != 0;
    if (reuse && (e = hipMemAddressFree(r.va, r.reserved)) != hipSuccess) die("hipMemAddressFree", e);
}

