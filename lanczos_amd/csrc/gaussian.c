/*
 * Deterministic N(0,1) start vector (host, plain C).
 *
 * The reference draws the start vector from MKL dlarnv seeded with time(NULL)
 * on the CPU path (src/util.c:48-56) and from cuRAND XORWOW, seed 123456789,
 * on the CUDA path (src/lanczos_cuda.c:394-432); neither can be reproduced
 * without those libraries.  This generator is counter based: element i depends
 * only on (seed, i), so any row partition of the vector sees the same numbers
 * and the CPU checker can be fed the identical vector.
 */
#include <math.h>
#include <stdint.h>

#include "../../include/lanczos_hip.h"

static inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void lanczos_hip_gaussian_fill(uint64_t seed, int64_t row_begin, int64_t n, double *out) {
    const double two_pi = 6.283185307179586476925286766559;
    const uint64_t key = mix64(seed);
#pragma omp parallel for schedule(static) if (n >= 65536)
    for (int64_t i = 0; i < n; i++) {
        const uint64_t g = (uint64_t)(row_begin + i);
        const uint64_t r1 = mix64(key ^ (2 * g));
        const uint64_t r2 = mix64(key ^ (2 * g + 1));
        const double u1 = ((double)(r1 >> 11) + 1.0) / 9007199254740992.0; /* (0, 1] */
        const double u2 = (double)(r2 >> 11) / 9007199254740992.0;         /* [0, 1) */
        out[i] = sqrt(-2.0 * log(u1)) * cos(two_pi * u2);
    }
}
