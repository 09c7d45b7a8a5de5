// w <- w + sum of R arrays (2 GiB each) with the writes gathered: a workgroup adds up K tiles of 16 KiB in LDS
// (reads only), then writes the K tiles in one go.  K = 1 is the plain order.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int K>
__global__ void __launch_bounds__(256) kb(double *w, const double *x, long stride, int R, long tiles) {
    extern __shared__ d2v keep[];  // K x 1024 d2v
    const long per = (tiles + gridDim.x - 1) / gridDim.x;
    const long t0 = blockIdx.x * per, t1 = t0 + per < tiles ? t0 + per : tiles;
    for (long tb = t0; tb < t1; tb += K) {
        const int kn = (int)(t1 - tb < K ? t1 - tb : K);
        for (int k = 0; k < kn; ++k) {
            const long t = tb + k;
            const d2v *p = reinterpret_cast<const d2v *>(w + t * 2048) + threadIdx.x;
            d2v a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = p[u * 256];
            for (int r = 0; r < R; ++r) {
                const d2v *q = reinterpret_cast<const d2v *>(x + r * stride + t * 2048) + threadIdx.x;
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] += __builtin_nontemporal_load(q + u * 256);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) keep[k * 1024 + u * 256 + threadIdx.x] = a[u];
        }
        for (int k = 0; k < kn; ++k) {
            d2v *p = reinterpret_cast<d2v *>(w + (tb + k) * 2048) + threadIdx.x;
#pragma unroll
            for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(keep[k * 1024 + u * 256 + threadIdx.x], p + u * 256);
        }
    }
}
int main() {
    const long n = 1L << 28, tiles = n / 2048;
    double *buf;
    CK(hipMalloc(&buf, sizeof(double) * n * 9));
    CK(hipMemset(buf, 0, sizeof(double) * n * 9));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    auto run = [&](int K, int g, int R, auto kern) {
        CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, K * 16384));
        auto launch = [&]() { hipLaunchKernelGGL(kern, dim3(256 * g), dim3(256), K * 16384, 0, buf, buf + n, n, R, tiles); };
        launch();
        CK(hipEventRecord(a));
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        printf("K=%d tiles gathered  groups/CU %d  R=%d  %.3f ms  %.0f GB/s\n", K, g, R, ms / 3, (R + 2.0) * n * 8 / (ms / 3 * 1e-3) / 1e9);
    };
    for (int R : {2, 8}) {
        run(1, 2, R, kb<1>);
        run(1, 4, R, kb<1>);
        run(2, 2, R, kb<2>);
        run(4, 2, R, kb<4>);
        run(4, 1, R, kb<4>);
        run(8, 1, R, kb<8>);
    }
    return 0;
}
