#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#define ITERS 4096
// 8 independent chains x 4 = 32 instrs per iteration
#define BODY(INSTR) \
  for (int it = 0; it < ITERS; ++it) { \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) { \
      asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7) \
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c)); } }

#define K(name, INSTR) __global__ __launch_bounds__(256) void name(unsigned* out, unsigned b, unsigned c) { \
  unsigned a[8]; for (int i=0;i<8;++i) a[i]=threadIdx.x*i+b; BODY(INSTR) unsigned s=0; for(int i=0;i<8;++i) s^=a[i]; out[blockIdx.x*256+threadIdx.x]=s; }

#define I_PKADD(n) "v_pk_add_i16 %" #n ", %" #n ", %8\n"
#define I_PKMAX(n) "v_pk_max_i16 %" #n ", %" #n ", %8\n"
#define I_PKSUBC(n) "v_pk_sub_u16 %" #n ", %" #n ", %8 clamp\n"
#define I_ADD32(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define I_MAX32(n) "v_max_i32 %" #n ", %" #n ", %8\n"
#define I_MAX3(n) "v_max3_i32 %" #n ", %" #n ", %8, %9\n"
#define I_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define I_PKADDF16(n) "v_pk_add_f16 %" #n ", %" #n ", %8\n"
#define I_PKMAXF16(n) "v_pk_max_f16 %" #n ", %" #n ", %8\n"
#define I_ADDF32(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define I_MAXF32(n) "v_max_f32 %" #n ", %" #n ", %8\n"
#define I_MAX3F32(n) "v_max3_f32 %" #n ", %" #n ", %8, %9\n"
#define I_FMAF32(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_DPP(n) "v_mov_b32_dpp %" #n ", %" #n " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_MAD24(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %9\n"
#define I_MAXU16(n) "v_max_u16 %" #n ", %" #n ", %8\n"
#define I_ADDU16(n) "v_add_u16 %" #n ", %" #n ", %8\n"
#define I_SUBU16C(n) "v_sub_u16 %" #n ", %" #n ", %8 clamp\n"
#define I_PKMAD(n) "v_pk_mad_i16 %" #n ", %" #n ", %8, %9\n"
#define I_PKMINU16(n) "v_pk_min_u16 %" #n ", %" #n ", %8\n"
#define I_PKFMAF16(n) "v_pk_fma_f16 %" #n ", %" #n ", %8, %9\n"
#define I_MAX3I16(n) "v_max3_i16 %" #n ", %" #n ", %8, %9\n"
#define I_MAX3U16(n) "v_max3_u16 %" #n ", %" #n ", %8, %9\n"
#define I_MED3(n) "v_med3_i32 %" #n ", %" #n ", %8, %9\n"
#define I_ADDSDWA(n) "v_add_u32_sdwa %" #n ", %" #n ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define I_SADU8(n) "v_sad_u8 %" #n ", %" #n ", %8, %9\n"
#define I_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define I_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define I_SUBI32C(n) "v_sub_i32 %" #n ", %" #n ", %8 clamp\n"
#define I_SUBU32(n) "v_sub_u32 %" #n ", %" #n ", %8\n"
#define I_MAXI16(n) "v_max_i16 %" #n ", %" #n ", %8\n"
#define I_PKMAX3F16(n) "v_pk_maximum3_f16 %" #n ", %" #n ", %8, %9\n"
#define I_MAXIMUM3F32(n) "v_maximum3_f32 %" #n ", %" #n ", %8, %9\n"
#define I_PKMINF16(n) "v_pk_min_f16 %" #n ", %" #n ", %8\n"

K(k_pkadd, I_PKADD) K(k_pkmax, I_PKMAX) K(k_pksubc, I_PKSUBC) K(k_add32, I_ADD32) K(k_max32, I_MAX32) K(k_max3, I_MAX3)
K(k_add3, I_ADD3) K(k_pkaddf16, I_PKADDF16) K(k_pkmaxf16, I_PKMAXF16) K(k_addf32, I_ADDF32) K(k_maxf32, I_MAXF32)
K(k_max3f32, I_MAX3F32) K(k_fmaf32, I_FMAF32) K(k_dpp, I_DPP) K(k_mad24, I_MAD24) K(k_maxu16, I_MAXU16) K(k_addu16, I_ADDU16)
K(k_subu16c, I_SUBU16C) K(k_pkmad, I_PKMAD) K(k_pkminu16, I_PKMINU16) K(k_pkfmaf16, I_PKFMAF16) K(k_max3i16, I_MAX3I16) K(k_max3u16, I_MAX3U16)
K(k_med3, I_MED3) K(k_addsdwa, I_ADDSDWA) K(k_sadu8, I_SADU8) K(k_perm, I_PERM) K(k_cndmask, I_CNDMASK) K(k_xor, I_XOR) K(k_subi32c, I_SUBI32C) K(k_subu32, I_SUBU32) K(k_maxi16, I_MAXI16) K(k_pkmax3f16, I_PKMAX3F16) K(k_maximum3f32, I_MAXIMUM3F32) K(k_pkminf16, I_PKMINF16)

typedef void (*kfn)(unsigned*, unsigned, unsigned);
int main() {
  unsigned* out; hipMalloc(&out, 256*8*256*4*4);
  struct E { const char* n; kfn f; } es[] = {
    {"v_pk_add_i16", k_pkadd}, {"v_pk_max_i16", k_pkmax}, {"v_pk_sub_u16 clamp", k_pksubc}, {"v_pk_min_u16", k_pkminu16}, {"v_pk_mad_i16", k_pkmad},
    {"v_add_u32", k_add32}, {"v_sub_u32", k_subu32}, {"v_sub_i32 clamp", k_subi32c}, {"v_max_i32", k_max32}, {"v_max3_i32", k_max3}, {"v_med3_i32", k_med3}, {"v_add3_u32", k_add3},
    {"v_pk_add_f16", k_pkaddf16}, {"v_pk_max_f16", k_pkmaxf16}, {"v_pk_fma_f16", k_pkfmaf16},
    {"v_add_f32", k_addf32}, {"v_max_f32", k_maxf32}, {"v_max3_f32", k_max3f32}, {"v_fma_f32", k_fmaf32},
    {"v_mov_b32_dpp row_shr:1", k_dpp}, {"v_mad_u32_u24", k_mad24}, {"v_max_u16", k_maxu16}, {"v_max_i16", k_maxi16}, {"v_add_u16", k_addu16}, {"v_sub_u16 clamp", k_subu16c},
    {"v_max3_i16", k_max3i16}, {"v_max3_u16", k_max3u16}, {"v_add_u32_sdwa", k_addsdwa}, {"v_sad_u8", k_sadu8}, {"v_perm_b32", k_perm}, {"v_cndmask_b32", k_cndmask}, {"v_xor_b32", k_xor},
    {"v_pk_maximum3_f16", k_pkmax3f16}, {"v_maximum3_f32", k_maximum3f32}, {"v_pk_min_f16", k_pkminf16}};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps : {1, 2, 4, 8}) {
    printf("== %d waves/SIMD (blocks/CU=%d)\n", wps, wps);
    for (auto& e : es) {
      dim3 grid(256 * wps), block(256);
      hipLaunchKernelGGL(e.f, grid, block, 0, 0, out, 3u, 5u);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(e.f, grid, block, 0, 0, out, 3u, 5u);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double instr_per_simd = (double)ITERS * 32 * wps;   // wave-instrs per SIMD
      double cyc = ms * 1e-3 * 2.4e9 / instr_per_simd;
      printf("%-26s %8.3f ms  %.2f cycles/wave-instr/SIMD @2.4GHz\n", e.n, ms, cyc);
    }
  }
  return 0;
}
