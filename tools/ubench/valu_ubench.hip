// Issue interval of single VALU instructions on gfx950 (cycles per wave64 instruction per SIMD), one instruction kind per kernel: eight independent
// destination registers, ITER x 8 x UNROLL instructions per wave, 1 and 4 waves per SIMD.  Used to price the instruction mix of the regime-B pair
// kernels (tools/kernel_mix.py): which of their VALU instructions are full-rate, which quarter-rate?
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/valu_ubench.hip -o tools/ubench/valu_ubench
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define KERNEL64(NAME, ASM)                                                                                             \
  __global__ void __launch_bounds__(256) NAME(int iters, double* out) {                                                 \
    double d0 = threadIdx.x, d1 = 1.5, d2 = 2.5, d3 = 3.5, d4 = 4.5, d5 = 5.5, d6 = 6.5, d7 = 7.5, a = 1.0000001, b = 0.5; \
    int e = 1;                                                                                                         \
    for (int it = 0; it < iters; ++it) {                                                                               \
      asm volatile(ASM(%0) ASM(%1) ASM(%2) ASM(%3) ASM(%4) ASM(%5) ASM(%6) ASM(%7) ASM(%0) ASM(%1) ASM(%2) ASM(%3) ASM(%4) ASM(%5) ASM(%6) ASM(%7) \
                   : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a), "v"(b), "v"(e)); \
    }                                                                                                                  \
    if (d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 == 123.456) out[0] = d0;                                                 \
  }
#define KERNEL32(NAME, ASM)                                                                                             \
  __global__ void __launch_bounds__(256) NAME(int iters, double* out) {                                                 \
    int d0 = threadIdx.x, d1 = 1, d2 = 2, d3 = 3, d4 = 4, d5 = 5, d6 = 6, d7 = 7, a = 3, b = 5;                          \
    double f = 1.25;                                                                                                   \
    for (int it = 0; it < iters; ++it) {                                                                               \
      asm volatile(ASM(%0) ASM(%1) ASM(%2) ASM(%3) ASM(%4) ASM(%5) ASM(%6) ASM(%7) ASM(%0) ASM(%1) ASM(%2) ASM(%3) ASM(%4) ASM(%5) ASM(%6) ASM(%7) \
                   : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a), "v"(b), "v"(f)); \
    }                                                                                                                  \
    if (d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 == 123456789) out[0] = d0;                                               \
  }
#define A_FMA(D) "v_fma_f64 " #D ", " #D ", %8, %9\n\t"
#define A_MUL(D) "v_mul_f64 " #D ", " #D ", %8\n\t"
#define A_ADD(D) "v_add_f64 " #D ", " #D ", %9\n\t"
#define A_LDEXP(D) "v_ldexp_f64 " #D ", " #D ", %10\n\t"
#define A_RNDNE(D) "v_rndne_f64 " #D ", " #D "\n\t"
#define A_MOV64(D) "v_mov_b64 " #D ", %8\n\t"
#define A_CNDMASK64A(D) "v_cndmask_b32 " #D ", " #D ", %8, vcc\n\t"
#define A_ADDU(D) "v_add_u32 " #D ", " #D ", %8\n\t"
#define A_AND(D) "v_and_b32 " #D ", " #D ", %8\n\t"
#define A_LSHL(D) "v_lshlrev_b32 " #D ", 1, " #D "\n\t"
#define A_LSHLADD(D) "v_lshl_add_u32 " #D ", " #D ", 1, %8\n\t"
#define A_MULLO(D) "v_mul_lo_u32 " #D ", " #D ", %8\n\t"
#define A_BFE(D) "v_bfe_u32 " #D ", " #D ", 3, 7\n\t"
#define A_CVTI(D) "v_cvt_i32_f64 " #D ", %10\n\t"
#define A_MOVDPP(D) "v_mov_b32_dpp " #D ", " #D " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define A_MOV32(D) "v_mov_b32 " #D ", %8\n\t"
// the accumulate form of the pair kernels' exponent, fma(s, v, acc): a 64-bit SGPR operand; and the same with only FOUR independent accumulators
// (the four points of a trip): does a dependent v_fma_f64 issue every 4 x 4.3 cycles?
#define KERNEL64S(NAME, ASM, NACC)                                                                                      \
  __global__ void __launch_bounds__(256) NAME(int iters, double* out) {                                                 \
    double d0 = threadIdx.x, d1 = 1.5, d2 = 2.5, d3 = 3.5, d4 = 4.5, d5 = 5.5, d6 = 6.5, d7 = 7.5, b = 0.5;                \
    double sv = __builtin_bit_cast(double, (long)__builtin_amdgcn_readfirstlane(iters) | 0x3ff0000000000000L);            \
    for (int it = 0; it < iters; ++it) {                                                                               \
      if (NACC == 8)                                                                                                   \
        asm volatile(ASM(%0) ASM(%1) ASM(%2) ASM(%3) ASM(%4) ASM(%5) ASM(%6) ASM(%7) ASM(%0) ASM(%1) ASM(%2) ASM(%3) ASM(%4) ASM(%5) ASM(%6) ASM(%7) \
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "s"(sv), "v"(b));  \
      else                                                                                                             \
        asm volatile(ASM(%0) ASM(%1) ASM(%2) ASM(%3) ASM(%0) ASM(%1) ASM(%2) ASM(%3) ASM(%0) ASM(%1) ASM(%2) ASM(%3) ASM(%0) ASM(%1) ASM(%2) ASM(%3) \
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "s"(sv), "v"(b));  \
    }                                                                                                                  \
    if (d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 == 123.456) out[0] = d0;                                                 \
  }
#define A_FMAS(D) "v_fma_f64 " #D ", %8, %9, " #D "\n\t"
#define A_FMAV(D) "v_fma_f64 " #D ", %9, %9, " #D "\n\t"
KERNEL64S(k_fma_sgpr8, A_FMAS, 8)
KERNEL64S(k_fma_sgpr4, A_FMAS, 4)
KERNEL64S(k_fma_vacc8, A_FMAV, 8)
KERNEL64S(k_fma_vacc4, A_FMAV, 4)
KERNEL64(k_fma, A_FMA)
KERNEL64(k_mul, A_MUL)
KERNEL64(k_add, A_ADD)
KERNEL64(k_ldexp, A_LDEXP)
KERNEL64(k_rndne, A_RNDNE)
KERNEL64(k_mov64, A_MOV64)
KERNEL32(k_addu, A_ADDU)
KERNEL32(k_and, A_AND)
KERNEL32(k_lshl, A_LSHL)
KERNEL32(k_lshladd, A_LSHLADD)
KERNEL32(k_mullo, A_MULLO)
KERNEL32(k_bfe, A_BFE)
KERNEL32(k_cvti, A_CVTI)
KERNEL32(k_movdpp, A_MOVDPP)
KERNEL32(k_mov32, A_MOV32)
KERNEL32(k_cndmask, A_CNDMASK64A)

int main() {
  double* out; hipMalloc(&out, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct { const char* name; void (*k)(int, double*); } ks[] = {
      {"v_fma_f64", k_fma}, {"v_fma_f64 acc += s * v (8 accumulators)", k_fma_sgpr8}, {"v_fma_f64 acc += s * v (4 accumulators)", k_fma_sgpr4},
      {"v_fma_f64 acc += v * v (8 accumulators)", k_fma_vacc8}, {"v_fma_f64 acc += v * v (4 accumulators)", k_fma_vacc4}, {"v_mul_f64", k_mul}, {"v_add_f64", k_add}, {"v_ldexp_f64", k_ldexp}, {"v_rndne_f64", k_rndne}, {"v_mov_b64", k_mov64},
      {"v_cvt_i32_f64", k_cvti}, {"v_add_u32", k_addu}, {"v_and_b32", k_and}, {"v_lshlrev_b32", k_lshl}, {"v_lshl_add_u32", k_lshladd}, {"v_mul_lo_u32", k_mullo},
      {"v_bfe_u32", k_bfe}, {"v_cndmask_b32", k_cndmask}, {"v_mov_b32", k_mov32}, {"v_mov_b32_dpp quad_perm", k_movdpp}};
  const int iters = 20000;
  // clock: v_fma_f64 is known at 4.3 cycles per instruction when the part holds its clock; report everything relative to the wall clock at 2.4 GHz nominal
  for (auto& k : ks)
    for (int wps : {1, 2, 4, 8}) {
      const int blocks = 256 * wps;
      hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, 100, out);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, iters, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%-42s %d wave(s) per SIMD: %6.2f cycles per instruction per SIMD at 2.4 GHz\n", k.name, wps, ms * 1e-3 * 2.4e9 / (16.0 * iters * wps));
    }
  return 0;
}
