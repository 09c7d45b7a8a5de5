// w <- w + sum of R other arrays, all of the block variant's size (2 GiB): the rate at R reads + 1 read + 1 write,
// for arrays exactly 2 GiB apart and a little more, in the block kernels' shape (a wavefront owns 4 KiB tiles and
// walks the arrays one after the other) and with all arrays' loads issued together.
// hipcc --offload-arch=gfx950 -O3 mix_probe.hip -o mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// one array after the other, next array's tile in flight while this one is added (the update kernel's order)
template <bool WRITE>
__global__ void __launch_bounds__(256) k_seq(double *w, const double *q, long stride, int R, long tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double keep = 0.0;
    for (long t = (long)blockIdx.x * 4 + wave; t < tiles; t += (long)gridDim.x * 4) {
        double *p = w + t * 512 + lane;
        double acc[8], x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = p[u * 64];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = __builtin_nontemporal_load(q + t * 512 + lane + u * 64);
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += x[u];
            if (r + 1 < R) {
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = __builtin_nontemporal_load(q + (r + 1) * stride + t * 512 + lane + u * 64);
            }
        }
        if (WRITE) {
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u * 64] = acc[u];
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) keep += acc[u];
        }
    }
    if (!WRITE && keep == 1.2345e300) w[0] = keep;
}

// all arrays' tiles asked for together, then the sum written (the single-vector kernels' order)
template <int R, bool WRITE>
__global__ void __launch_bounds__(256) k_all(double *w, const double *q, long stride, long tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double keep = 0.0;
    for (long t = (long)blockIdx.x * 4 + wave; t < tiles; t += (long)gridDim.x * 4) {
        double *p = w + t * 512 + lane;
        double acc[8], x[R][8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = p[u * 64];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int u = 0; u < 8; ++u) x[r][u] = __builtin_nontemporal_load(q + r * stride + t * 512 + lane + u * 64);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += x[r][u];
        if (WRITE) {
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u * 64] = acc[u];
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) keep += acc[u];
        }
    }
    if (!WRITE && keep == 1.2345e300) w[0] = keep;
}

template <int R>
void run_all(double *buf, long stride, long tiles, long blk, int g, hipEvent_t a, hipEvent_t b, const char *tag) {
    for (int write = 0; write < 2; ++write) {
        auto launch = [&]() {
            if (write)
                hipLaunchKernelGGL((k_all<R, true>), dim3(256 * g), dim3(256), 0, 0, buf, buf + stride, stride, tiles);
            else
                hipLaunchKernelGGL((k_all<R, false>), dim3(256 * g), dim3(256), 0, 0, buf, buf + stride, stride, tiles);
        };
        launch();
        CK(hipEventRecord(a));
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        const double passes = R + 1 + write;
        printf("%s together  groups/CU %d  R=%d %s  %.3f ms  %.0f GB/s\n", tag, g, R, write ? "read+write" : "read only ", ms / 3,
               passes * blk * 8 / (ms / 3 * 1e-3) / 1e9);
    }
}

int main(int argc, char **argv) {
    const long n = 1L << (argc > 1 ? atoi(argv[1]) : 24), tiles = n / 32, blk = n * 16;
    printf("rows 2^%d, %ld MiB an array\n", argc > 1 ? atoi(argv[1]) : 24, blk * 8 >> 20);
    const int RMAX = 8;
    double *buf;
    const long pad = 3 * 4096 / 8 + 0;  // doubles
    CK(hipMalloc(&buf, sizeof(double) * ((RMAX + 1) * (blk + pad))));
    CK(hipMemset(buf, 0, sizeof(double) * ((RMAX + 1) * (blk + pad))));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int padded = 0; padded < 2; ++padded) {
        const long stride = blk + (padded ? pad : 0);
        for (int g : {2, 4}) {
            run_all<1>(buf, stride, tiles, blk, g, a, b, padded ? "padded" : "exact ");
            run_all<2>(buf, stride, tiles, blk, g, a, b, padded ? "padded" : "exact ");
            run_all<4>(buf, stride, tiles, blk, g, a, b, padded ? "padded" : "exact ");
            run_all<8>(buf, stride, tiles, blk, g, a, b, padded ? "padded" : "exact ");
            for (int R : {1, 2, 4, 8}) {
                for (int write = 0; write < 2; ++write) {
                    auto launch = [&]() {
                        if (write)
                            hipLaunchKernelGGL(k_seq<true>, dim3(256 * g), dim3(256), 0, 0, buf, buf + stride, stride, R, tiles);
                        else
                            hipLaunchKernelGGL(k_seq<false>, dim3(256 * g), dim3(256), 0, 0, buf, buf + stride, stride, R, tiles);
                    };
                    launch();
                    CK(hipEventRecord(a));
                    for (int i = 0; i < 3; ++i) launch();
                    CK(hipEventRecord(b));
                    CK(hipEventSynchronize(b));
                    float ms;
                    CK(hipEventElapsedTime(&ms, a, b));
                    const double passes = R + 1 + write;
                    printf("stride 1x%s  groups/CU %d  R=%d %s  %.3f ms  %.0f GB/s\n", padded ? " + 12 KiB" : "         ", g, R,
                           write ? "read+write" : "read only ", ms / 3, passes * blk * 8 / (ms / 3 * 1e-3) / 1e9);
                }
            }
        }
    }
    return 0;
}
