// Diagnostic (not part of the library): what does a launch boundary cost on this GPU?  Back-to-back launches on one stream of
//   (a) an empty kernel, (b) a kernel whose every wave spins for a fixed number of cycles, (c) the same plus one 16-byte
//   load and store per lane over an 8 MB array (the step kernel's state traffic), at the step kernel's grid (1024 x 64).
// Prints microseconds per launch; (b) minus the spin time is the boundary cost the step kernel cannot avoid.
// Build: hipcc -O3 --offload-arch=gfx950 tools/launch_floor.hip -o tools/bin/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(64) void k_empty() {}
__global__ __launch_bounds__(64) void k_spin(unsigned long long cycles)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) __builtin_amdgcn_s_sleep(1);
}
__global__ __launch_bounds__(64) void k_spin_mem(unsigned long long cycles, uint4* a, int n_rec)
{
    const unsigned e = blockIdx.x * 64 + threadIdx.x;
    uint4 v[8];
    for (int r = 0; r < n_rec; ++r) v[r] = a[(size_t)r * gridDim.x * 64 + e];
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) __builtin_amdgcn_s_sleep(1);
    for (int r = 0; r < n_rec; ++r) { v[r].x += 1; a[(size_t)r * gridDim.x * 64 + e] = v[r]; }
}

// the same with the stores' cache policy chosen: 0 plain, 1 nt (streaming), 2 sc1, 3 sc0 sc1 (write-through to memory)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(64) void k_spin_mem_pol(unsigned long long cycles, u32x4* a, int)
{
    const unsigned e = blockIdx.x * 64 + threadIdx.x;
    u32x4 v0 = a[e], v1 = a[65536 + e], v2 = a[2 * 65536 + e], v3 = a[3 * 65536 + e];
    const unsigned long long t0 = __builtin_readcyclecounter();
    // waves finish at different times, as the step kernel's do: a third of them spin the full time, the rest half of it
    const unsigned long long mine = (blockIdx.x % 3 == 0) ? cycles : cycles / 2;
    while (__builtin_readcyclecounter() - t0 < mine) __builtin_amdgcn_s_sleep(1);
    v0.x += 1; v1.x += 1; v2.x += 1; v3.x += 1;
    u32x4 *p0 = a + e, *p1 = a + 65536 + e, *p2 = a + 2 * 65536 + e, *p3 = a + 3 * 65536 + e;
#define ST4(POL) asm volatile("global_store_dwordx4 %0, %4, off " POL "\n\tglobal_store_dwordx4 %1, %5, off " POL "\n\t" \
                              "global_store_dwordx4 %2, %6, off " POL "\n\tglobal_store_dwordx4 %3, %7, off " POL           \
                              ::"v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "memory")
    if (MODE == 0) ST4("");
    if (MODE == 1) ST4("nt");
    if (MODE == 2) ST4("sc1");
    if (MODE == 3) ST4("sc0 sc1");
}

// host-side enqueue cost: a kernel with the step kernel's argument block (7 leading scalars + 1.2 KB of structs + 3 pointers)
struct Big { char b[1176]; };
__global__ __launch_bounds__(64) void k_args(unsigned* a, double* b, unsigned* c, unsigned char* d, const int* e, const int* f, unsigned n,
                                             Big big, int* o, float* r, unsigned char* dn)
{
    if (n == 0xffffffffu && a) a[0] = (unsigned)big.b[17];
}

__global__ __launch_bounds__(64) void k_small(unsigned* a, double* b, unsigned* c, unsigned char* d, const int* e, const int* f, unsigned n, int nd,
                                              int* o, float* r, unsigned char* dn)
{
    if (n == 0xffffffffu && a) a[0] = (unsigned)nd;
}
__global__ __launch_bounds__(64) void k_one(unsigned* a) { if (a == (unsigned*)1) a[0] = 0; }

template <class F> double enqueue_us(F launch, int n, hipStream_t s)
{
    for (int i = 0; i < 200; ++i) launch();
    CK(hipStreamSynchronize(s));
    double best = 1e30;
    for (int rep = 0; rep < 7; ++rep) {
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int i = 0; i < n; ++i) launch();
        clock_gettime(CLOCK_MONOTONIC, &t1);
        CK(hipStreamSynchronize(s));
        const double us = ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) * 1e-3 / n;
        if (us < best) best = us;
    }
    return best;
}

template <class F> double per_launch_us(F launch, int n, hipStream_t s)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 200; ++i) launch();
    CK(hipStreamSynchronize(s));
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(a, s));
        for (int i = 0; i < n; ++i) launch();
        CK(hipEventRecord(b, s));
        CK(hipEventSynchronize(b));
        float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
        if (ms * 1e3 / n < best) best = ms * 1e3 / n;
    }
    return best;
}

int main(int argc, char** argv)
{
    hipStream_t s; CK(hipStreamCreate(&s));
    const int n = 2000;
    uint4* a; CK(hipMalloc(&a, (size_t)8 * 65536 * 16)); CK(hipMemset(a, 0, (size_t)8 * 65536 * 16));
    for (int grid : {256, 1024, 4096}) {
        printf("grid %4d x 64: empty %.2f us/launch", grid, per_launch_us([&] { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(64), 0, s); }, n, s));
        for (unsigned long long cyc : {0ull, 2000ull, 6500ull, 12500ull}) {
            const double t = per_launch_us([&] { hipLaunchKernelGGL(k_spin, dim3(grid), dim3(64), 0, s, cyc); }, n, s);
            printf(" | spin %5llu: %.2f", cyc, t);
        }
        printf("\n");
        if (grid <= 1024)
            for (int n_rec : {1, 4, 8}) {
                printf("   + %d x 16 B load/store per lane:", n_rec);
                for (unsigned long long cyc : {0ull, 6500ull, 12500ull})
                    printf(" spin %5llu: %.2f", cyc, per_launch_us([&] { hipLaunchKernelGGL(k_spin_mem, dim3(grid), dim3(64), 0, s, cyc, a, n_rec); }, n, s));
                printf("\n");
            }
    }
    printf("store policy (grid 1024 x 64, 4 x 16 B load/store per lane, 1/3 of the waves spin 12500 cycles, the rest 6250):\n");
    printf("   plain %.2f", per_launch_us([&] { hipLaunchKernelGGL(k_spin_mem_pol<0>, dim3(1024), dim3(64), 0, s, 12500ull, (u32x4*)a, 4); }, n, s));
    printf(" | nt %.2f", per_launch_us([&] { hipLaunchKernelGGL(k_spin_mem_pol<1>, dim3(1024), dim3(64), 0, s, 12500ull, (u32x4*)a, 4); }, n, s));
    printf(" | sc1 %.2f", per_launch_us([&] { hipLaunchKernelGGL(k_spin_mem_pol<2>, dim3(1024), dim3(64), 0, s, 12500ull, (u32x4*)a, 4); }, n, s));
    printf(" | sc0 sc1 %.2f us/launch\n", per_launch_us([&] { hipLaunchKernelGGL(k_spin_mem_pol<3>, dim3(1024), dim3(64), 0, s, 12500ull, (u32x4*)a, 4); }, n, s));
    printf("   all waves spin 12500:  plain %.2f", per_launch_us([&] { hipLaunchKernelGGL(k_spin_mem, dim3(1024), dim3(64), 0, s, 12500ull, a, 4); }, n, s));
    printf("\n");
    {   // host enqueue cost per launch (clock stopped before the synchronize; 64 launches so that the queue never fills)
        Big big; memset(&big, 1, sizeof big);
        unsigned* pa = (unsigned*)a;
        const double t_chevron = enqueue_us([&] { hipLaunchKernelGGL(k_args, dim3(1024), dim3(64), 0, s, pa, (double*)a, pa, (unsigned char*)a,
                                                                     (const int*)a, (const int*)a, 65536u, big, (int*)a, (float*)a, (unsigned char*)a); }, 64, s);
        const double t_empty = enqueue_us([&] { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(64), 0, s); }, 64, s);
        // the same kernel through hipModuleLaunchKernel with ONE pre-built argument buffer
        hipFunction_t fn = nullptr;
        double t_mod = -1;
        if (hipGetFuncBySymbol(&fn, (const void*)k_args) == hipSuccess && fn) {
            struct __attribute__((packed, aligned(8))) Args { unsigned* a; double* b; unsigned* c; unsigned char* d; const int* e; const int* f; unsigned n; unsigned pad; Big big; int* o; float* r; unsigned char* dn; } ab;
            memset(&ab, 0, sizeof ab);
            ab.a = pa; ab.b = (double*)a; ab.c = pa; ab.d = (unsigned char*)a; ab.e = (const int*)a; ab.f = (const int*)a; ab.n = 65536u; ab.big = big;
            ab.o = (int*)a; ab.r = (float*)a; ab.dn = (unsigned char*)a;
            size_t sz = sizeof ab;
            void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &ab, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
            t_mod = enqueue_us([&] { (void)hipModuleLaunchKernel(fn, 1024, 1, 1, 64, 1, 1, 0, s, nullptr, extra); }, 64, s);
            CK(hipGetLastError());
        }
        const double t_small = enqueue_us([&] { hipLaunchKernelGGL(k_small, dim3(1024), dim3(64), 0, s, pa, (double*)a, pa, (unsigned char*)a,
                                                                   (const int*)a, (const int*)a, 65536u, 4, (int*)a, (float*)a, (unsigned char*)a); }, 64, s);
        const double t_one = enqueue_us([&] { hipLaunchKernelGGL(k_one, dim3(1024), dim3(64), 0, s, pa); }, 64, s);
        double t_small_mod = -1;
        {
            hipFunction_t fs = nullptr;
            if (hipGetFuncBySymbol(&fs, (const void*)k_small) == hipSuccess && fs) {
                struct SArgs { unsigned* a; double* b; unsigned* c; unsigned char* d; const int* e; const int* f; unsigned n; int nd; int* o; float* r; unsigned char* dn; } sa;
                sa.a = pa; sa.b = (double*)a; sa.c = pa; sa.d = (unsigned char*)a; sa.e = (const int*)a; sa.f = (const int*)a; sa.n = 65536u; sa.nd = 4;
                sa.o = (int*)a; sa.r = (float*)a; sa.dn = (unsigned char*)a;
                size_t ssz = sizeof sa;
                void* ex[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &sa, HIP_LAUNCH_PARAM_BUFFER_SIZE, &ssz, HIP_LAUNCH_PARAM_END};
                t_small_mod = enqueue_us([&] { (void)hipModuleLaunchKernel(fs, 1024, 1, 1, 64, 1, 1, 0, s, nullptr, ex); }, 64, s);
                CK(hipGetLastError());
            }
        }
        printf("host enqueue per launch: one pointer argument %.2f us | 11 scalar arguments (80 B) %.2f us | the same through hipModuleLaunchKernel + one buffer %.2f us\n", t_one, t_small, t_small_mod);
        printf("host enqueue per launch: empty kernel %.2f us | 11 arguments, 1.2 KB, <<<>>> %.2f us | same through hipModuleLaunchKernel + one buffer %.2f us\n",
               t_empty, t_chevron, t_mod);
    }
    {   // the small runtime calls gw_step makes around its launch
        auto t = [&](auto f) { timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0); for (int i = 0; i < 100000; ++i) f();
                               clock_gettime(CLOCK_MONOTONIC, &t1); return ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / 100000.0; };
        hipStreamCaptureStatus cs;
        int dev = 0;
        printf("ns per call: hipSetDevice %.0f | hipGetDevice %.0f | hipStreamIsCapturing %.0f | hipGetLastError %.0f\n",
               t([&] { (void)hipSetDevice(0); }), t([&] { (void)hipGetDevice(&dev); }), t([&] { (void)hipStreamIsCapturing(s, &cs); }),
               t([&] { (void)hipGetLastError(); }));
    }
    // the counter's rate: cycles per microsecond (spin a long while, time it)
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 100000000ull);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("__builtin_readcyclecounter (s_memtime): %.1f ticks per microsecond\n", 1e8 / (ms * 1e3));
    }
    return 0;
}
