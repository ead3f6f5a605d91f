// how long hipMalloc / first touch / hipFree take for large buffers on this box (dev tool)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    size_t f, t;
    (void)hipMemGetInfo(&f, &t);
    printf("free %.1f GB total %.1f GB\n", f / 1e9, t / 1e9);
    {   // several buffers of 24 GB held at once
        std::vector<void *> ps;
        for (int i = 0; i < 8; ++i) {
            void *p = nullptr;
            double t0 = now();
            hipError_t e = hipMalloc(&p, (size_t)24 << 30);
            printf("24 GB #%d: malloc %.1f ms (%s)\n", i, now() - t0, hipGetErrorString(e));
            if (e == hipSuccess) ps.push_back(p);
        }
        double t0 = now();
        for (void *p : ps) (void)hipFree(p);
        printf("free all: %.1f ms\n", now() - t0);
    }
    const size_t gbs[] = {40, 48, 56, 64, 200};
    for (size_t gb : gbs) {
        void *p = nullptr;
        double t0 = now();
        hipError_t e = hipMalloc(&p, gb << 30);
        double t1 = now();
        if (e != hipSuccess) { printf("%zu GB: %s\n", gb, hipGetErrorString(e)); continue; }
        (void)hipMemset(p, 0, gb << 30);
        (void)hipDeviceSynchronize();
        double t2 = now();
        (void)hipFree(p);
        double t3 = now();
        printf("%4zu GB: malloc %.1f ms, memset %.1f ms, free %.1f ms\n", gb, t1 - t0, t2 - t1, t3 - t2);
    }
    return 0;
}
