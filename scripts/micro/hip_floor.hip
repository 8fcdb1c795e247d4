// What a process on this box pays for the HIP runtime whatever it computes (the command line's fixed cost): runtime start, first
// stream, a 2 GiB allocation cleared, one trivial kernel (code object load), exit.  Build: hipcc --offload-arch=gfx950 -O2 -o hip_floor hip_floor.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <unistd.h>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_touch(unsigned* p) { p[threadIdx.x] = threadIdx.x; }
int main(int argc, char** argv) {
    const double t0 = now();
    int n = 0;
    hipGetDeviceCount(&n);
    const double t1 = now();
    hipStream_t s;
    hipStreamCreate(&s);
    const double t2 = now();
    void* p = nullptr;
    hipMalloc(&p, 2ULL << 30);
    hipMemsetAsync(p, 0xFF, 2ULL << 30, s);
    hipStreamSynchronize(s);
    const double t3 = now();
    k_touch<<<1, 64, 0, s>>>((unsigned*)p);
    hipStreamSynchronize(s);
    const double t4 = now();
    printf("devices %d | runtime start %.1f ms | first stream %.1f | 2 GiB allocated and cleared %.1f | first kernel %.1f | main so far %.1f\n", n, t1 - t0, t2 - t1, t3 - t2,
           t4 - t3, t4 - t0);
    fflush(stdout);
    if (argc > 1) _exit(0);      // as the command line leaves
    return 0;
}
