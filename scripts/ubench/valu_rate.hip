// VALU issue-rate microbenchmark (development aid, evidence for DESIGN.md §4 "what bounds K34"): cycles per
// wavefront-instruction and SIMD for the integer / float / packed / cross-lane ops the K34, K4 and K6 kernels are made
// of (or could be made of), at 8 waves per SIMD, four independent chains per statement so that dependency latency is not
// what is measured.  Inline asm keeps the compiler from fusing or hoisting the ops.  The clock is read from the device
// (hipDeviceAttributeClockRate) AND derived from s_memtime so that "cycles" are shader cycles, not a nominal 2.4 GHz.
//   hipcc -O3 --offload-arch=gfx950 -w valu_rate.hip -o valu_rate && ./valu_rate > r03_valu_rate.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define ASM4(S) asm volatile(S : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "s"(m) : "vcc", "s10", "s11", "s12", "s13")
// 64-bit register pairs for the packed-fp32 ops
#define ASM4P(S) asm volatile(S : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd) : "v"(pe), "v"(pf))

#define KINDS(X)                                                                                                                                  \
    X(0, "v_add_u32 (VOP2)", "v_add_u32 %0, %0, %4\nv_add_u32 %1, %1, %5\nv_add_u32 %2, %2, %4\nv_add_u32 %3, %3, %5")                                 \
    X(1, "v_xor/or/sub/ashr (VOP2)", "v_xor_b32 %0, %0, %4\nv_or_b32 %1, %1, %5\nv_sub_u32 %2, %2, %4\nv_ashrrev_i32 %3, 3, %3")                     \
    X(2, "v_lshrrev / v_and (VOP2)", "v_lshrrev_b32 %0, 3, %0\nv_and_b32 %1, 15, %1\nv_lshrrev_b32 %2, 5, %2\nv_and_b32 %3, 15, %3")                 \
    X(3, "v_max_i32 (VOP2)", "v_max_i32 %0, %0, %4\nv_max_i32 %1, %1, %5\nv_max_i32 %2, %2, %4\nv_max_i32 %3, %3, %5")                                 \
    X(4, "v_min_u32 (VOP2)", "v_min_u32 %0, %0, %4\nv_min_u32 %1, %1, %5\nv_min_u32 %2, %2, %4\nv_min_u32 %3, %3, %5")                                 \
    X(5, "v_mul_u32_u24 (VOP2)", "v_mul_u32_u24 %0, %0, %4\nv_mul_u32_u24 %1, %1, %5\nv_mul_u32_u24 %2, %2, %4\nv_mul_u32_u24 %3, %3, %5")             \
    X(6, "v_mul_i32_i24 (VOP2)", "v_mul_i32_i24 %0, %0, %4\nv_mul_i32_i24 %1, %1, %5\nv_mul_i32_i24 %2, %2, %4\nv_mul_i32_i24 %3, %3, %5")             \
    X(7, "v_mad_u32_u24 (VOP3)", "v_mad_u32_u24 %0, %0, %4, %5\nv_mad_u32_u24 %1, %1, %5, %4\nv_mad_u32_u24 %2, %2, %4, %5\nv_mad_u32_u24 %3, %3, %5, %4") \
    X(8, "v_mad_i32_i24 (VOP3)", "v_mad_i32_i24 %0, %0, %4, %5\nv_mad_i32_i24 %1, %1, %5, %4\nv_mad_i32_i24 %2, %2, %4, %5\nv_mad_i32_i24 %3, %3, %5, %4") \
    X(9, "v_mul_lo_u32 (VOP3)", "v_mul_lo_u32 %0, %0, %4\nv_mul_lo_u32 %1, %1, %5\nv_mul_lo_u32 %2, %2, %4\nv_mul_lo_u32 %3, %3, %5")                  \
    X(10, "v_bcnt_u32_b32 (VOP3)", "v_bcnt_u32_b32 %0, %4, %0\nv_bcnt_u32_b32 %1, %5, %1\nv_bcnt_u32_b32 %2, %4, %2\nv_bcnt_u32_b32 %3, %5, %3")       \
    X(11, "v_alignbit_b32 (VOP3)", "v_alignbit_b32 %0, %0, %4, 7\nv_alignbit_b32 %1, %1, %5, 9\nv_alignbit_b32 %2, %2, %4, 11\nv_alignbit_b32 %3, %3, %5, 13") \
    X(12, "v_bfe_u32 (VOP3)", "v_bfe_u32 %0, %4, 10, 10\nv_bfe_u32 %1, %5, 10, 10\nv_bfe_u32 %2, %4, 20, 10\nv_bfe_u32 %3, %5, 0, 10")                \
    X(13, "v_add3_u32 (VOP3)", "v_add3_u32 %0, %0, %4, %5\nv_add3_u32 %1, %1, %5, %4\nv_add3_u32 %2, %2, %4, %5\nv_add3_u32 %3, %3, %5, %4")           \
    X(14, "v_or3_b32 (VOP3)", "v_or3_b32 %0, %0, %4, %5\nv_or3_b32 %1, %1, %5, %4\nv_or3_b32 %2, %2, %4, %5\nv_or3_b32 %3, %3, %5, %4")               \
    X(15, "v_and_or_b32 (VOP3)", "v_and_or_b32 %0, %0, %4, %5\nv_and_or_b32 %1, %1, %5, %4\nv_and_or_b32 %2, %2, %4, %5\nv_and_or_b32 %3, %3, %5, %4") \
    X(16, "v_lshl_or_b32 (VOP3)", "v_lshl_or_b32 %0, %0, 4, %4\nv_lshl_or_b32 %1, %1, 4, %5\nv_lshl_or_b32 %2, %2, 4, %4\nv_lshl_or_b32 %3, %3, 4, %5") \
    X(17, "v_lshl_add_u32 (VOP3)", "v_lshl_add_u32 %0, %0, 1, %4\nv_lshl_add_u32 %1, %1, 1, %5\nv_lshl_add_u32 %2, %2, 1, %4\nv_lshl_add_u32 %3, %3, 1, %5") \
    X(18, "v_bfi_b32 (VOP3)", "v_bfi_b32 %0, %4, %0, %5\nv_bfi_b32 %1, %5, %1, %4\nv_bfi_b32 %2, %4, %2, %5\nv_bfi_b32 %3, %5, %3, %4")               \
    X(19, "v_perm_b32 (VOP3)", "v_perm_b32 %0, %0, %4, %5\nv_perm_b32 %1, %1, %5, %4\nv_perm_b32 %2, %2, %4, %5\nv_perm_b32 %3, %3, %5, %4")           \
    X(20, "v_max3_i32 (VOP3)", "v_max3_i32 %0, %0, %4, %5\nv_max3_i32 %1, %1, %5, %4\nv_max3_i32 %2, %2, %4, %5\nv_max3_i32 %3, %3, %5, %4")           \
    X(21, "v_med3_i32 (VOP3)", "v_med3_i32 %0, %0, %4, %5\nv_med3_i32 %1, %1, %5, %4\nv_med3_i32 %2, %2, %4, %5\nv_med3_i32 %3, %3, %5, %4")           \
    X(22, "v_sad_u8 (VOP3)", "v_sad_u8 %0, %4, %5, %0\nv_sad_u8 %1, %5, %4, %1\nv_sad_u8 %2, %4, %5, %2\nv_sad_u8 %3, %5, %4, %3")                    \
    X(23, "v_dot4_u32_u8 (VOP3P)", "v_dot4_u32_u8 %0, %4, %5, %0\nv_dot4_u32_u8 %1, %5, %4, %1\nv_dot4_u32_u8 %2, %4, %5, %2\nv_dot4_u32_u8 %3, %5, %4, %3") \
    X(24, "v_dot8_u32_u4 (VOP3P)", "v_dot8_u32_u4 %0, %4, %5, %0\nv_dot8_u32_u4 %1, %5, %4, %1\nv_dot8_u32_u4 %2, %4, %5, %2\nv_dot8_u32_u4 %3, %5, %4, %3") \
    X(25, "v_cmp_gt_i32 -> vcc (VOPC)", "v_cmp_gt_i32 vcc, %0, %4\nv_cmp_gt_i32 vcc, %1, %5\nv_cmp_gt_i32 vcc, %2, %4\nv_cmp_gt_i32 vcc, %3, %5")     \
    X(26, "v_cmp_e64 -> sgpr ; v_cndmask_e64", "v_cmp_gt_i32_e64 s[10:11], %0, %4\nv_cndmask_b32_e64 %1, %1, %5, s[10:11]\nv_cmp_gt_i32_e64 s[12:13], %2, %4\nv_cndmask_b32_e64 %3, %3, %5, s[12:13]") \
    X(27, "v_cndmask_b32_e64, SGPR mask", "v_cndmask_b32_e64 %0, %0, %4, %6\nv_cndmask_b32_e64 %1, %1, %5, %6\nv_cndmask_b32_e64 %2, %2, %4, %6\nv_cndmask_b32_e64 %3, %3, %5, %6") \
    X(28, "v_cndmask_b32_e32 (vcc)", "v_cndmask_b32_e32 %0, %0, %4, vcc\nv_cndmask_b32_e32 %1, %1, %5, vcc\nv_cndmask_b32_e32 %2, %2, %4, vcc\nv_cndmask_b32_e32 %3, %3, %5, vcc") \
    X(29, "v_pk_add_i16 (VOP3P)", "v_pk_add_i16 %0, %0, %4\nv_pk_add_i16 %1, %1, %5\nv_pk_add_i16 %2, %2, %4\nv_pk_add_i16 %3, %3, %5")               \
    X(30, "v_pk_max_i16 (VOP3P)", "v_pk_max_i16 %0, %0, %4\nv_pk_max_i16 %1, %1, %5\nv_pk_max_i16 %2, %2, %4\nv_pk_max_i16 %3, %3, %5")               \
    X(31, "v_pk_mad_i16 (VOP3P)", "v_pk_mad_i16 %0, %0, %4, %5\nv_pk_mad_i16 %1, %1, %5, %4\nv_pk_mad_i16 %2, %2, %4, %5\nv_pk_mad_i16 %3, %3, %5, %4") \
    X(32, "v_fma_f32 (VOP3)", "v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %5, %4\nv_fma_f32 %2, %2, %4, %5\nv_fma_f32 %3, %3, %5, %4")               \
    X(33, "v_fmac_f32 (VOP2)", "v_fmac_f32 %0, %4, %5\nv_fmac_f32 %1, %5, %4\nv_fmac_f32 %2, %4, %5\nv_fmac_f32 %3, %5, %4")                           \
    X(34, "v_add_f32 (VOP2)", "v_add_f32 %0, %0, %4\nv_add_f32 %1, %1, %5\nv_add_f32 %2, %2, %4\nv_add_f32 %3, %3, %5")                               \
    X(35, "v_max_f32 (VOP2)", "v_max_f32 %0, %0, %4\nv_max_f32 %1, %1, %5\nv_max_f32 %2, %2, %4\nv_max_f32 %3, %3, %5")                               \
    X(36, "v_cmp_lt_f32 -> vcc (VOPC)", "v_cmp_lt_f32 vcc, %0, %4\nv_cmp_lt_f32 vcc, %1, %5\nv_cmp_lt_f32 vcc, %2, %4\nv_cmp_lt_f32 vcc, %3, %5")     \
    X(37, "v_cvt_f32_u32 (VOP1)", "v_cvt_f32_u32 %0, %4\nv_cvt_f32_u32 %1, %5\nv_cvt_f32_u32 %2, %4\nv_cvt_f32_u32 %3, %5")                           \
    X(38, "v_cvt_f32_ubyte0 (VOP1)", "v_cvt_f32_ubyte0 %0, %4\nv_cvt_f32_ubyte1 %1, %5\nv_cvt_f32_ubyte2 %2, %4\nv_cvt_f32_ubyte3 %3, %5")            \
    X(39, "v_mov_b32 dpp row_shr:1", "v_mov_b32_dpp %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %5 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %2, %4 row_shr:2 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %5 row_shr:2 row_mask:0xf bank_mask:0xf") \
    X(40, "v_add_u32 dpp row_shr:1", "v_add_u32_dpp %0, %4, %0 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %1, %5, %1 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %2, %4, %2 row_shr:2 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %3, %5, %3 row_shr:2 row_mask:0xf bank_mask:0xf") \
    X(41, "v_add_u32 sdwa (byte select)", "v_add_u32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\nv_add_u32_sdwa %1, %1, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\nv_add_u32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\nv_add_u32_sdwa %3, %3, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3") \
    X(42, "ds_bpermute_b32", "ds_bpermute_b32 %0, %4, %0\nds_bpermute_b32 %1, %5, %1\nds_bpermute_b32 %2, %4, %2\nds_bpermute_b32 %3, %5, %3\ns_waitcnt lgkmcnt(0)") \
    X(43, "v_readlane_b32 -> sgpr", "v_readlane_b32 s10, %0, 3\nv_readlane_b32 s11, %1, 5\nv_readlane_b32 s12, %2, 7\nv_readlane_b32 s13, %3, 9")      \
    X(44, "v_mbcnt_lo/hi (VOP3)", "v_mbcnt_lo_u32_b32 %0, %4, %0\nv_mbcnt_hi_u32_b32 %1, %5, %1\nv_mbcnt_lo_u32_b32 %2, %4, %2\nv_mbcnt_hi_u32_b32 %3, %5, %3")

#define KINDS_PK(X)                                                                                                                               \
    X(100, "v_pk_fma_f32 (2 x fp32)", "v_pk_fma_f32 %0, %0, %4, %5\nv_pk_fma_f32 %1, %1, %5, %4\nv_pk_fma_f32 %2, %2, %4, %5\nv_pk_fma_f32 %3, %3, %5, %4") \
    X(101, "v_pk_add_f32 (2 x fp32)", "v_pk_add_f32 %0, %0, %4\nv_pk_add_f32 %1, %1, %5\nv_pk_add_f32 %2, %2, %4\nv_pk_add_f32 %3, %3, %5")           \
    X(102, "v_pk_mul_f32 (2 x fp32)", "v_pk_mul_f32 %0, %0, %4\nv_pk_mul_f32 %1, %1, %5\nv_pk_mul_f32 %2, %2, %4\nv_pk_mul_f32 %3, %3, %5")           \
    X(103, "v_lshlrev_b64 (VOP3)", "v_lshlrev_b64 %0, 3, %0\nv_lshlrev_b64 %1, 5, %1\nv_lshlrev_b64 %2, 7, %2\nv_lshlrev_b64 %3, 9, %3")

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed, unsigned long long *ticks) {
    uint32_t a = threadIdx.x ^ seed, b = a * 3u + 1u, c = a + 7u, d = a ^ 0x55u, e = a + 11u, f = a ^ 0x33u;
    unsigned long long m = 0x5555AAAA5555AAAAull ^ seed;
    unsigned long long pa = a, pb = b, pc = c, pd = d, pe = e, pf = f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#define X(N, NAME, S) if (KIND == N) ASM4(S);
            KINDS(X)
#undef X
#define X(N, NAME, S) if (KIND == N) ASM4P(S);
            KINDS_PK(X)
#undef X
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && threadIdx.x == 0) *ticks = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ (uint32_t)(pa ^ pb ^ pc ^ pd);
}

static double g_clock_ghz = 2.4;

template <int KIND>
void run(const char *name, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = 2048;
    uint32_t *out;
    unsigned long long *ticks, h_ticks = 0;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipMalloc(&ticks, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 16, 1u, ticks);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u, ticks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h_ticks, ticks, 8, hipMemcpyDeviceToHost);
    const double per_simd = (double)iters * 8 * 4 * waves_per_simd;  // wave-instructions one SIMD executes
    printf("%-36s waves/SIMD %d  %.3f ms -> %.3f ns per wavefront-instruction per SIMD = %.2f cycles at %.2f GHz\n", name, waves_per_simd, ms,
           ms * 1e6 / per_simd, ms * 1e6 / per_simd * g_clock_ghz, g_clock_ghz);
    hipFree(out); hipFree(ticks);
}

// the shader clock under this load: a kernel of dependent v_add_u32 chains of known length against wall time
__global__ void k_clock(unsigned long long *out, int iters) {
    uint32_t a = threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();   // s_memtime: counts at the constant 100 MHz reference
    const unsigned long long c0 = clock64();
    for (int i = 0; i < iters; i++) asm volatile("v_add_u32 %0, %0, %0\nv_add_u32 %0, %0, %0\nv_add_u32 %0, %0, %0\nv_add_u32 %0, %0, %0" : "+v"(a));
    const unsigned long long c1 = clock64();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = c1 - c0; out[2] = a; }
}

int main() {
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    printf("# hipDeviceAttributeClockRate %.3f GHz (the figure cycles are quoted at; the part may hold less under load)\n", khz * 1e-6);
    if (khz > 0) g_clock_ghz = khz * 1e-6;
    {
        unsigned long long *d, h[3];
        hipMalloc(&d, 24);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_clock, dim3(2048), dim3(256), 0, 0, d, 1000);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_clock, dim3(2048), dim3(256), 0, 0, d, 200000);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("# clock probe: kernel %.3f ms, readcyclecounter %llu ticks, clock64 %llu ticks (one wave's lifetime)\n", ms, h[0], h[1]);
        hipFree(d);
    }
#define X(N, NAME, S) run<N>(NAME, 8);
    KINDS(X)
    KINDS_PK(X)
#undef X
    // fewer waves: what one or two wavefronts per SIMD sustain
    run<0>("v_add_u32 (VOP2)", 1); run<0>("v_add_u32 (VOP2)", 2); run<0>("v_add_u32 (VOP2)", 4);
    run<11>("v_alignbit_b32 (VOP3)", 1); run<11>("v_alignbit_b32 (VOP3)", 2); run<11>("v_alignbit_b32 (VOP3)", 4);
    run<10>("v_bcnt_u32_b32 (VOP3)", 2); run<10>("v_bcnt_u32_b32 (VOP3)", 4);
    run<32>("v_fma_f32 (VOP3)", 2); run<32>("v_fma_f32 (VOP3)", 4);
    return 0;
}
