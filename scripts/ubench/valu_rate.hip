// VALU issue-rate microbenchmark (development aid): wave-instructions per cycle per SIMD for the integer ops
// the K4/K6 kernels are made of, at 1..8 waves per SIMD.   hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a = threadIdx.x ^ seed, b = a * 3u + 1u, c = a + 7u, d = a ^ 0x55u, e = a + 11u, f = a ^ 0x33u, g = a + 5u, h = a ^ 9u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (KIND == 0) { a += b; c += d; e += f; g += h; b ^= a; d ^= c; f ^= e; h ^= g; }  // add / xor
            if (KIND == 1) { a = __builtin_amdgcn_alignbit(a, b, 7); c = __builtin_amdgcn_alignbit(c, d, 9); e = __builtin_amdgcn_alignbit(e, f, 11); g = __builtin_amdgcn_alignbit(g, h, 13);
                             b = __builtin_amdgcn_alignbit(b, a, 3); d = __builtin_amdgcn_alignbit(d, c, 5); f = __builtin_amdgcn_alignbit(f, e, 17); h = __builtin_amdgcn_alignbit(h, g, 19); }
            if (KIND == 2) { a = (a > b) ? c : a; c = (c > d) ? e : c; e = (e > f) ? g : e; g = (g > h) ? a : g; b += 1; d += 1; f += 1; h += 1; }  // cmp + cndmask + add
            if (KIND == 3) { a = ((int32_t)(a << 12)) >> 22; c = (c >> 5) & 0xF; e = ((int32_t)(e << 2)) >> 22; g = (g >> 9) & 0xF; a += b; c += d; e += f; g += h; }  // bfe + add
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}

template <int KIND>
void run(const char *name, int waves_per_simd) {
    int cus = 256, blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
    uint32_t *out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 16, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts_per_wave = (double)iters * 16 * 8;  // nominal: 8 ops per unrolled body
    double per_simd = insts_per_wave * waves_per_simd;
    printf("%-22s waves/SIMD %d  %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, waves_per_simd, ms,
           ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4, 8}) { run<0>("add/xor", w); run<1>("alignbit", w); run<2>("cmp+cndmask+add(12)", w); run<3>("bfe/shift+add", w); }
    return 0;
}
