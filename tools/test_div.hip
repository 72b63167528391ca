// test_div.hip -- exhaustive check that the 3-instruction division by a small integer
// constant used in the score correction (x / k, place.cpp:421) is bit-identical to the
// IEEE-754 correctly rounded quotient for EVERY float x with 2^-120 <= |x| <= 2^120:
//     y = RN(1/k);  q = RN(x*y);  r = fma(-q, k, x) (exact);  q' = fma(r, y, q)
// (Markstein's correction step).  Prints the number of mismatches per k = 1..32.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void check(uint32_t kmer, uint32_t lo_exp, unsigned long long *bad, uint32_t *first_bad)
{
    const float k = (float)kmer;
    const float y = __fdiv_rn(1.0f, k);
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long local = 0;
    for (uint64_t u = tid; u < (1ull << 32); u += stride) {
        const uint32_t bits = (uint32_t)u;
        const uint32_t exp = (bits >> 23) & 0xffu;
        if (exp == 255) continue;  // inf, nan
        if (exp < lo_exp) continue;
        const float x = __uint_as_float(bits);
        const float ref = __fdiv_rn(x, k);
        const float q = __fmul_rn(x, y);
        const float r = __fmaf_rn(-q, k, x);
        const float q2 = __fmaf_rn(r, y, q);
        if (__float_as_uint(q2) != __float_as_uint(ref)) {
            if (local == 0) atomicMin(first_bad, bits);
            ++local;
        }
    }
    if (local) atomicAdd(bad, local);
}

int main(int argc, char **argv)
{
    unsigned long long *d_bad;
    uint32_t *d_first;
    hipMalloc(&d_bad, 8);
    hipMalloc(&d_first, 4);
    int total_bad = 0;
    const uint32_t lo_exp = argc > 1 ? (uint32_t)atoi(argv[1]) : 7u;  // smallest biased exponent checked
    for (uint32_t k = 1; k <= 32; ++k) {
        unsigned long long zero = 0;
        uint32_t ff = 0xffffffffu;
        hipMemcpy(d_bad, &zero, 8, hipMemcpyHostToDevice);
        hipMemcpy(d_first, &ff, 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, k, lo_exp, d_bad, d_first);
        unsigned long long bad = 0;
        uint32_t first = 0;
        hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost);
        hipMemcpy(&first, d_first, 4, hipMemcpyDeviceToHost);
        printf("k=%2u mismatches=%llu first=0x%08x\n", k, bad, first);
        total_bad += bad != 0;
    }
    printf("%s\n", total_bad ? "SOME k FAIL" : "all k exact");
    return 0;
}
