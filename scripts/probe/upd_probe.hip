// w -= sum_l h[l] V_l over L = 150 vectors of 2^24 doubles (the headline's reorthogonalisation update at k = 150)
// in several shapes: what rate the basis stream reaches in each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// C: a workgroup owns 16 KiB tiles, a thread four 16-byte pieces 4 KiB apart, UN vectors' loads unrolled together
template <int UN>
__global__ void __launch_bounds__(256) k_c(const d2v *V, long ld2, int L, d2v *w, long tiles, const double *h) {
    for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
        const long base = t * 1024 + threadIdx.x;
        d2v a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = w[base + u * 256];
        const d2v *vl = V + base;
#pragma unroll UN
        for (int l = 0; l < L; ++l, vl += ld2) {
            const double c = h[l];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] -= __builtin_nontemporal_load(vl + u * 256) * c;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(a[u], w + base + u * 256);
    }
}
// A: a wavefront owns 4 KiB tiles, 16-byte loads (lane = 2 doubles, 4 pieces 1 KiB apart)
template <int UN>
__global__ void __launch_bounds__(256) k_a(const d2v *V, long ld2, int L, d2v *w, long tiles, const double *h) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long t = (long)blockIdx.x * 4 + wave; t < tiles * 4; t += (long)gridDim.x * 4) {
        const long base = t * 256 + lane;
        d2v a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = w[base + u * 64];
        const d2v *vl = V + base;
#pragma unroll UN
        for (int l = 0; l < L; ++l, vl += ld2) {
            const double c = h[l];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] -= __builtin_nontemporal_load(vl + u * 64) * c;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(a[u], w + base + u * 64);
    }
}
// B: a workgroup owns 32 KiB tiles (eight pieces a thread)
template <int UN>
__global__ void __launch_bounds__(256) k_b(const d2v *V, long ld2, int L, d2v *w, long tiles, const double *h) {
    for (long t = blockIdx.x; t < tiles / 2; t += gridDim.x) {
        const long base = t * 2048 + threadIdx.x;
        d2v a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = w[base + u * 256];
        const d2v *vl = V + base;
#pragma unroll UN
        for (int l = 0; l < L; ++l, vl += ld2) {
            const double c = h[l];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] -= __builtin_nontemporal_load(vl + u * 256) * c;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) __builtin_nontemporal_store(a[u], w + base + u * 256);
    }
}

// P pieces of 16 bytes a thread (a workgroup owns P x 4 KiB), UN vectors unrolled together
template <int P, int UN>
__global__ void __launch_bounds__(256) k_p(const d2v *V, long ld2, int L, d2v *w, long tiles, const double *h) {
    const long ptiles = tiles * 4 / P;   // tiles counts 16 KiB
    for (long t = blockIdx.x; t < ptiles; t += gridDim.x) {
        const long base = t * (256 * P) + threadIdx.x;
        d2v a[P];
#pragma unroll
        for (int u = 0; u < P; ++u) a[u] = w[base + u * 256];
        const d2v *vl = V + base;
#pragma unroll UN
        for (int l = 0; l < L; ++l, vl += ld2) {
            const double c = h[l];
#pragma unroll
            for (int u = 0; u < P; ++u) a[u] -= __builtin_nontemporal_load(vl + u * 256) * c;
        }
#pragma unroll
        for (int u = 0; u < P; ++u) __builtin_nontemporal_store(a[u], w + base + u * 256);
    }
}

int main() {
    const long n = 1L << 24, ld2 = n / 2, tiles = n / 2048;
    const int L = 150;
    d2v *V, *w;
    double *h;
    CK(hipMalloc(&V, sizeof(double) * n * L));
    CK(hipMalloc(&w, sizeof(double) * n));
    CK(hipMalloc(&h, sizeof(double) * L));
    CK(hipMemset(V, 0, sizeof(double) * n * L));
    CK(hipMemset(w, 0, sizeof(double) * n));
    CK(hipMemset(h, 0, sizeof(double) * L));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    auto run = [&](const char *name, int g, auto kern) {
        auto launch = [&]() { hipLaunchKernelGGL(kern, dim3(256 * g), dim3(256), 0, 0, (const d2v *)V, ld2, L, w, tiles, (const double *)h); };
        launch();
        CK(hipEventRecord(a));
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        printf("%-44s groups/CU %d  %.3f ms  %.0f GB/s\n", name, g, ms / 3, (L + 2.0) * n * 8 / (ms / 3 * 1e-3) / 1e9);
    };
    for (int g : {2, 3, 4, 5, 6, 8}) {
        run("P=8 (32 KiB), 1 vector", g, k_p<8, 1>);
        run("P=8 (32 KiB), 2 vectors", g, k_p<8, 2>);
        run("P=16 (64 KiB), 1 vector", g, k_p<16, 1>);
        run("P=16 (64 KiB), 2 vectors", g, k_p<16, 2>);
        run("P=32 (128 KiB), 1 vector", g, k_p<32, 1>);
    }
    for (int g : {4}) {
        run("C workgroup tiles 16 KiB, 2 vectors unrolled", g, k_c<2>);
        run("C workgroup tiles 16 KiB, 4 vectors unrolled", g, k_c<4>);
        run("C workgroup tiles 16 KiB, 1 vector", g, k_c<1>);
        run("A wavefront tiles 4 KiB, 2 vectors unrolled", g, k_a<2>);
        run("A wavefront tiles 4 KiB, 4 vectors unrolled", g, k_a<4>);
        run("B workgroup tiles 32 KiB, 2 vectors unrolled", g, k_b<2>);
    }
    return 0;
}
