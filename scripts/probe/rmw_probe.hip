// Read-modify-write of a 2 GiB array of doubles (rows of 16) in several access shapes: which of them the memory
// system serves at what rate.  hipcc --offload-arch=gfx950 -O3 rmw_probe.hip -o rmw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// A: a wavefront owns 4 KiB tiles (32 rows x 16), lane = (row & 3, column), 8-byte loads and stores 512 bytes apart
__global__ void __launch_bounds__(256) k_a(double *w, long tiles, double f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long t = (long)blockIdx.x * 4 + wave; t < tiles; t += (long)gridDim.x * 4) {
        double *p = w + t * 512 + lane;
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = p[u * 64];
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u * 64] = x[u] * f;
    }
}
// B: the same tiles, 16-byte loads and stores (lane = two neighbouring doubles), 1 KiB apart
template <bool NT>
__global__ void __launch_bounds__(256) k_b(double *w, long tiles, double f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long t = (long)blockIdx.x * 4 + wave; t < tiles; t += (long)gridDim.x * 4) {
        d2v *p = reinterpret_cast<d2v *>(w + t * 512) + lane;
        d2v x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = p[u * 64];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (NT) __builtin_nontemporal_store(x[u] * f, p + u * 64); else p[u * 64] = x[u] * f;
        }
    }
}
// C: a workgroup owns 16 KiB tiles, thread = four 16-byte pieces 4 KiB apart (the single-vector kernels' shape)
template <bool NT>
__global__ void __launch_bounds__(256) k_c(double *w, long tiles, double f) {
    for (long t = blockIdx.x; t < tiles / 4; t += gridDim.x) {
        d2v *p = reinterpret_cast<d2v *>(w + t * 2048) + threadIdx.x;
        d2v x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = p[u * 256];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (NT) __builtin_nontemporal_store(x[u] * f, p + u * 256); else p[u * 256] = x[u] * f;
        }
    }
}
// D: read only, shape A (sum kept alive through a never-taken store)
__global__ void __launch_bounds__(256) k_d(double *w, long tiles, double f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s = 0.0;
    for (long t = (long)blockIdx.x * 4 + wave; t < tiles; t += (long)gridDim.x * 4) {
        const double *p = w + t * 512 + lane;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += p[u * 64];
    }
    if (s == f) w[0] = s;
}
// E: shape C reading one array and writing another (no line is read and written)
template <bool NT>
__global__ void __launch_bounds__(256) k_e(const double *src, double *w, long tiles, double f) {
    for (long t = blockIdx.x; t < tiles / 4; t += gridDim.x) {
        const d2v *q = reinterpret_cast<const d2v *>(src + t * 2048) + threadIdx.x;
        d2v *p = reinterpret_cast<d2v *>(w + t * 2048) + threadIdx.x;
        d2v x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = q[u * 256];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (NT) __builtin_nontemporal_store(x[u] * f, p + u * 256); else p[u * 256] = x[u] * f;
        }
    }
}

int main() {
    const long n = 1L << 24, tiles = n / 32;
    const size_t bytes = sizeof(double) * n * 16;
    double *w, *src;
    CK(hipMalloc(&w, bytes));
    CK(hipMalloc(&src, bytes));
    CK(hipMemset(w, 0, bytes));
    CK(hipMemset(src, 0, bytes));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    auto run = [&](const char *name, int groups, double passes, auto launch) {
        for (int i = 0; i < 2; ++i) launch(groups);
        CK(hipEventRecord(a));
        for (int i = 0; i < 5; ++i) launch(groups);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        printf("%-34s groups/CU %d  %.3f ms  %.0f GB/s\n", name, groups / 256, ms / 5, passes * bytes / (ms / 5 * 1e-3) / 1e9);
    };
    for (int g : {1, 2, 4, 8}) {
        const int G = 256 * g;
        run("A wave tiles, 8 B", G, 2, [&](int gr) { hipLaunchKernelGGL(k_a, dim3(gr), dim3(256), 0, 0, w, tiles, 1.0); });
        run("B wave tiles, 16 B", G, 2, [&](int gr) { hipLaunchKernelGGL(k_b<false>, dim3(gr), dim3(256), 0, 0, w, tiles, 1.0); });
        run("B wave tiles, 16 B, nt stores", G, 2, [&](int gr) { hipLaunchKernelGGL(k_b<true>, dim3(gr), dim3(256), 0, 0, w, tiles, 1.0); });
        run("C workgroup tiles, 16 B", G, 2, [&](int gr) { hipLaunchKernelGGL(k_c<false>, dim3(gr), dim3(256), 0, 0, w, tiles, 1.0); });
        run("C workgroup tiles, 16 B, nt stores", G, 2, [&](int gr) { hipLaunchKernelGGL(k_c<true>, dim3(gr), dim3(256), 0, 0, w, tiles, 1.0); });
        run("D read only, 8 B", G, 1, [&](int gr) { hipLaunchKernelGGL(k_d, dim3(gr), dim3(256), 0, 0, w, tiles, 1.0); });
        run("E copy to another array", G, 2, [&](int gr) { hipLaunchKernelGGL(k_e<false>, dim3(gr), dim3(256), 0, 0, src, w, tiles, 1.0); });
        run("E copy to another array, nt stores", G, 2, [&](int gr) { hipLaunchKernelGGL(k_e<true>, dim3(gr), dim3(256), 0, 0, src, w, tiles, 1.0); });
    }
    return 0;
}
