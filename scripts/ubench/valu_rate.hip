// VALU issue-rate microbenchmark (development aid): cycles per wavefront-instruction and SIMD for the integer
// ops the K4/K6 kernels are made of, at 8 waves per SIMD.  Inline asm keeps the compiler from fusing the ops.
//   hipcc -O3 --offload-arch=gfx950 -w valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define OP4(INS)                                                                                  \
    asm volatile(INS "\n" INS "\n" INS "\n" INS : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "s"(m));

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a = threadIdx.x ^ seed, b = a * 3u + 1u, c = a + 7u, d = a ^ 0x55u, e = a + 11u, f = a ^ 0x33u;
    unsigned long long m = 0x5555AAAA5555AAAAull ^ seed;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            // four independent chains per statement so that dependency latency is not what is measured
            if (KIND == 0) asm volatile("v_add_u32 %0, %0, %4\nv_add_u32 %1, %1, %5\nv_add_u32 %2, %2, %4\nv_add_u32 %3, %3, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (KIND == 1) asm volatile("v_alignbit_b32 %0, %0, %4, 7\nv_alignbit_b32 %1, %1, %5, 9\nv_alignbit_b32 %2, %2, %4, 11\nv_alignbit_b32 %3, %3, %5, 13" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (KIND == 2) asm volatile("v_bfe_i32 %0, %4, 10, 10\nv_bfe_i32 %1, %5, 10, 10\nv_bfe_i32 %2, %4, 20, 10\nv_bfe_i32 %3, %5, 0, 10" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (KIND == 3) asm volatile("v_cndmask_b32_e64 %0, %0, %4, %6\nv_cndmask_b32_e64 %1, %1, %5, %6\nv_cndmask_b32_e64 %2, %2, %4, %6\nv_cndmask_b32_e64 %3, %3, %5, %6" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "s"(m));
            if (KIND == 4) asm volatile("v_add3_u32 %0, %0, %4, %5\nv_add3_u32 %1, %1, %5, %4\nv_add3_u32 %2, %2, %4, %5\nv_add3_u32 %3, %3, %5, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (KIND == 5) asm volatile("v_max_i32 %0, %0, %4\nv_max_i32 %1, %1, %5\nv_max_i32 %2, %2, %4\nv_max_i32 %3, %3, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (KIND == 6) asm volatile("v_lshl_or_b32 %0, %0, 4, %4\nv_lshl_or_b32 %1, %1, 4, %5\nv_lshl_or_b32 %2, %2, 4, %4\nv_lshl_or_b32 %3, %3, 4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (KIND == 7) asm volatile("v_cmp_gt_i32 vcc, %0, %4\nv_cmp_gt_i32 vcc, %1, %5\nv_cmp_gt_i32 vcc, %2, %4\nv_cmp_gt_i32 vcc, %3, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "vcc");
            if (KIND == 8) asm volatile("v_lshrrev_b32 %0, 3, %0\nv_and_b32 %1, 15, %1\nv_lshrrev_b32 %2, 5, %2\nv_and_b32 %3, 15, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (KIND == 10) asm volatile("v_pk_add_i16 %0, %0, %4\nv_pk_add_i16 %1, %1, %5\nv_pk_add_i16 %2, %2, %4\nv_pk_add_i16 %3, %3, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (KIND == 11) asm volatile("v_pk_max_i16 %0, %0, %4\nv_pk_max_i16 %1, %1, %5\nv_pk_max_i16 %2, %2, %4\nv_pk_max_i16 %3, %3, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (KIND == 12) asm volatile("v_cndmask_b32_e32 %0, %0, %4, vcc\nv_cndmask_b32_e32 %1, %1, %5, vcc\nv_cndmask_b32_e32 %2, %2, %4, vcc\nv_cndmask_b32_e32 %3, %3, %5, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : );
            if (KIND == 13) asm volatile("v_bfi_b32 %0, %4, %0, %5\nv_bfi_b32 %1, %5, %1, %4\nv_bfi_b32 %2, %4, %2, %5\nv_bfi_b32 %3, %5, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (KIND == 14) asm volatile("v_pk_ashrrev_i16 %0, 15, %0\nv_pk_sub_i16 %1, %1, %5\nv_pk_min_i16 %2, %2, %4\nv_pk_lshlrev_b16 %3, 1, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (KIND == 15) asm volatile("v_xor_b32 %0, %0, %4\nv_or_b32 %1, %1, %5\nv_sub_u32 %2, %2, %4\nv_ashrrev_i32 %3, 3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (KIND == 9) asm volatile("v_cmp_gt_i32_e64 s[10:11], %0, %4\nv_cndmask_b32_e64 %1, %1, %5, s[10:11]\nv_cmp_gt_i32_e64 s[12:13], %2, %4\nv_cndmask_b32_e64 %3, %3, %5, s[12:13]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "s10", "s11", "s12", "s13");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f;
}

template <int KIND>
void run(const char *name) {
    const int waves_per_simd = 8, blocks = 256 * waves_per_simd, iters = 4096;
    uint32_t *out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 16, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)iters * 8 * 4 * waves_per_simd;  // wave-instructions one SIMD executes
    printf("%-34s %.3f ms  -> %.2f ns = %.2f cycles (at 2.4 GHz) per wavefront-instruction per SIMD\n", name, ms, ms * 1e6 / per_simd,
           ms * 1e6 / per_simd * 2.4);
    hipFree(out);
}
int main() {
    run<0>("v_add_u32 (VOP2)"); run<5>("v_max_i32 (VOP2)"); run<8>("v_lshrrev / v_and (VOP2)"); run<1>("v_alignbit_b32 (VOP3)");
    run<2>("v_bfe_i32 (VOP3)"); run<4>("v_add3_u32 (VOP3)"); run<6>("v_lshl_or_b32 (VOP3)"); run<3>("v_cndmask_b32_e64, SGPR mask");
    run<7>("v_cmp_gt_i32 -> vcc"); run<9>("v_cmp_e64 -> sgpr ; v_cndmask_e64");
    run<12>("v_cndmask_b32_e32 (vcc)"); run<13>("v_bfi_b32 (VOP3)"); run<10>("v_pk_add_i16"); run<11>("v_pk_max_i16");
    run<14>("v_pk ashr/sub/min/lshl mix"); run<15>("xor/or/sub/ashr (VOP2)");
    return 0;
}
