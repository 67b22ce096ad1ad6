// hostregister_repro.cpp -- minimal reproduction attempt for the round-1 finding behind the withdrawn avifgpu_host_pin_planes():
// "hipHostRegister / hipHostUnregister of ordinary heap memory was followed by intermittent aborts in LATER pageable copies"
// (2 of 6 runs of the GPU suite, 0 of 8 without; commit a6957d6).  The pattern, stripped of the library: register a heap block,
// DMA into it asynchronously, unregister, free it, let the allocator hand the same addresses out again, and run plain pageable
// hipMemcpy calls on them.  Build: hipcc --offload-arch=gfx950 -O2 tools/hostregister_repro.cpp -o tools/hostregister_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s -> %s (iteration %d)\n", #x, hipGetErrorString(e_), it); return 1; } } while (0)

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    const size_t big = (size_t)(argc > 2 ? atoi(argv[2]) : 96) << 20;
    void* dev = nullptr;
    int it = 0;
    CK(hipMalloc(&dev, big));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (it = 0; it < iters; ++it) {
        // 1. heap block, registered for the duration of one "save", D2H lands in it asynchronously
        void* planes = malloc(big);
        memset(planes, 1, big);
        CK(hipHostRegister(planes, big, hipHostRegisterDefault));
        CK(hipMemsetAsync(dev, it & 0xff, big, st));
        CK(hipMemcpyAsync(planes, dev, big, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        CK(hipHostUnregister(planes));
        if (((unsigned char*)planes)[big / 2] != (unsigned char)(it & 0xff)) { printf("FAIL data mismatch after registered copy (iteration %d)\n", it); return 1; }
        free(planes);
        // 2. the allocator reuses the range: ordinary pageable copies of assorted sizes, both directions
        std::vector<void*> blocks;
        for (size_t sz : { big, big / 2, big / 3, (size_t)4096 * 3 + 17, big / 7 }) {
            void* p = malloc(sz);
            memset(p, 2, sz);
            CK(hipMemcpy(dev, p, sz, hipMemcpyHostToDevice));
            CK(hipMemcpy(p, dev, sz, hipMemcpyDeviceToHost));
            blocks.push_back(p);
        }
        for (void* p : blocks) free(p);
    }
    printf("ok: %d iterations of register -> async D2H -> unregister -> free -> pageable copies on recycled addresses, no failure\n", iters);
    return 0;
}
