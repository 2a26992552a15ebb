// f32_cell_rate.hip — what bounds the float32 score cell of a lone long query (config 5)?
// (1) issue rate of the candidate instructions of the cell (8 independent chains per wave, like valu_rate.hip);
// (2) the cell loop itself, R rows per lane in registers with the dependency pattern of sw_score_kernel
//     (x_r = clamp(diag_r + p_r); h_r = max3(x_r, Hg_r, ng); Hg_r = ng = h_r - g), cycles per cell and lane at
//     1 / 2 / 4 wavefronts per SIMD, in three forms: add clamp (VOP3) + max3 + sub, v_fma_mix_f32 with a float16
//     score operand + max3 + sub, and the running maximum every 4th step on top.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096
#define BODY(INSTR) \
  for (int it = 0; it < ITERS; ++it) { \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) { \
      asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7) \
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c), "s"(sg)); } }
#define K(name, INSTR) __global__ __launch_bounds__(256) void name(unsigned* out, unsigned b, unsigned c, unsigned sg) { \
  unsigned a[8]; for (int i=0;i<8;++i) a[i]=threadIdx.x*i+b; BODY(INSTR) unsigned s=0; for(int i=0;i<8;++i) s^=a[i]; out[blockIdx.x*256+threadIdx.x]=s; }

#define I_ADDF32(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define I_ADDF32C(n) "v_add_f32_e64 %" #n ", %" #n ", %8 clamp\n"
#define I_SUBS(n) "v_subrev_f32 %" #n ", %10, %" #n "\n"
#define I_ADDS(n) "v_add_f32 %" #n ", %10, %" #n "\n"
#define I_MAX3F32(n) "v_max3_f32 %" #n ", %" #n ", %8, %9\n"
#define I_MAXIMUM3(n) "v_maximum3_f32 %" #n ", %" #n ", %8, %9\n"
#define I_FMAMIX(n) "v_fma_mix_f32 %" #n ", %8, %10, %" #n " op_sel_hi:[1,0,0]\n"
#define I_FMAMIXC(n) "v_fma_mix_f32 %" #n ", %8, %10, %" #n " op_sel_hi:[1,0,0] clamp\n"
#define I_FMAMIXH(n) "v_fma_mix_f32 %" #n ", %8, %10, %" #n " op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp\n"
#define I_MAXF32(n) "v_max_f32 %" #n ", %" #n ", %8\n"
#define I_PKMAX3F16(n) "v_pk_maximum3_f16 %" #n ", %" #n ", %8, %9\n"
#define I_CVT(n) "v_cvt_f32_f16 %" #n ", %" #n "\n"
K(k_addf32, I_ADDF32) K(k_addf32c, I_ADDF32C) K(k_subs, I_SUBS) K(k_adds, I_ADDS) K(k_max3, I_MAX3F32) K(k_maximum3, I_MAXIMUM3)
K(k_fmamix, I_FMAMIX) K(k_fmamixc, I_FMAMIXC) K(k_fmamixh, I_FMAMIXH) K(k_maxf32, I_MAXF32) K(k_pkmax3f16, I_PKMAX3F16) K(k_cvt, I_CVT)

// packed float32 add: 64-bit register pairs
__global__ __launch_bounds__(256) void k_pkaddf32(unsigned* out, unsigned b, unsigned c, unsigned sg) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a[8]; f2 bb = {(float)b, (float)c};
  for (int i = 0; i < 8; ++i) a[i] = f2{(float)(threadIdx.x * i), (float)b};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      asm volatile("v_pk_add_f32 %0, %0, %8\nv_pk_add_f32 %1, %1, %8\nv_pk_add_f32 %2, %2, %8\nv_pk_add_f32 %3, %3, %8\n"
                   "v_pk_add_f32 %4, %4, %8\nv_pk_add_f32 %5, %5, %8\nv_pk_add_f32 %6, %6, %8\nv_pk_add_f32 %7, %7, %8\n"
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(bb));
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = __float_as_uint(s);
}

// ---- the cell loop ---------------------------------------------------------------------------
// FORM 0: v_add_f32_e64 clamp, v_max3_f32, v_subrev_f32 (gap in an SGPR) — sw_score_kernel's float32 cell
// FORM 1: v_fma_mix_f32 (float16 score x scale + diag) clamp instead of the add — float16 profile entries
// FORM 2: FORM 0 with plain VOP2 v_add_f32 (no clamp; not a valid cell — isolates what the VOP3 clamp form costs)
// MK: running maximum every MK-th step (one max3 per two rows)
template <int R, int FORM, int MK>
__global__ __launch_bounds__(256) void k_cell(unsigned* out, const float* pin, float gapv, float scale, int steps) {
  float H[R], Hg[R], p[R];
  for (int r = 0; r < R; ++r) { H[r] = 0.f; Hg[r] = -gapv; p[r] = pin[(threadIdx.x * R + r) & 1023]; }
  float up_prev = 0.f, mx = 0.f;
  for (int t = 0; t < steps; t += 4) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float up = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(H[R - 1]), 0x138, 0xf, 0xf, true));
      float diag = up_prev;
      up_prev = up;
      float ng;
      asm volatile("v_subrev_f32 %0, %1, %2" : "=v"(ng) : "s"(gapv), "v"(up));
      float tp = 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float w = H[r];
        float x, h;
        if (FORM == 0) asm volatile("v_add_f32_e64 %0, %1, %2 clamp" : "=v"(x) : "v"(diag), "v"(p[r]));
        else if (FORM == 1) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0] clamp" : "=v"(x) : "v"(p[r]), "s"(scale), "v"(diag));
        else asm volatile("v_add_f32 %0, %1, %2" : "=v"(x) : "v"(diag), "v"(p[r]));
        asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(h) : "v"(x), "v"(Hg[r]), "v"(ng));
        if (MK == 1 || k == MK - 1) {
          if (r & 1) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(tp), "v"(h));
          else tp = h;
        }
        diag = w;
        H[r] = h;
        asm volatile("v_subrev_f32 %0, %1, %2" : "=v"(ng) : "s"(gapv), "v"(h));
        Hg[r] = ng;
      }
    }
  }
  float s = mx;
  for (int r = 0; r < R; ++r) s += H[r];
  out[blockIdx.x * 256 + threadIdx.x] = __float_as_uint(s);
}

// float32 cell with the gap penalty in a VGPR (an SGPR operand halves the issue rate of v_add / v_sub, see the table above)
template <int R, bool MIX, int MK>
__global__ __launch_bounds__(256) void k_cell_vg(unsigned* out, const float* pin, float gapv, float scale, int steps) {
  float H[R], Hg[R], p[R];
  for (int r = 0; r < R; ++r) { H[r] = 0.f; Hg[r] = -gapv; p[r] = pin[(threadIdx.x * R + r) & 1023]; }
  float up_prev = 0.f, mx = 0.f;
  float gv = gapv; asm volatile("" : "+v"(gv));
  float sv = scale; asm volatile("" : "+v"(sv));
  for (int t = 0; t < steps; t += 4) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float up = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(H[R - 1]), 0x138, 0xf, 0xf, true));
      float diag = up_prev;
      up_prev = up;
      float ng;
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(ng) : "v"(up), "v"(gv));
      float tp = 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float w = H[r];
        float x, h;
        if (!MIX) asm volatile("v_add_f32_e64 %0, %1, %2 clamp" : "=v"(x) : "v"(diag), "v"(p[r]));
        else if (r & 1) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp" : "=v"(x) : "v"(p[r >> 1]), "v"(sv), "v"(diag));
        else asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0] clamp" : "=v"(x) : "v"(p[r >> 1]), "v"(sv), "v"(diag));
        asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(h) : "v"(x), "v"(Hg[r]), "v"(ng));
        if (MK == 1 || k == MK - 1) {
          if (r & 1) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(tp), "v"(h));
          else tp = h;
        }
        diag = w;
        H[r] = h;
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(ng) : "v"(h), "v"(gv));
        Hg[r] = ng;
      }
    }
  }
  float s = mx;
  for (int r = 0; r < R; ++r) s += H[r];
  out[blockIdx.x * 256 + threadIdx.x] = __float_as_uint(s);
}

// 16-bit integer cell on UNPACKED registers (one cell per VGPR, low half): v_add_u16 / v_max_i16 / v_sub_u16 clamp / v_max_i16
// — each of them issues at the double rate; the 16-bit profile entry of odd rows takes one shift more (v_add_u16 has no op_sel on gfx9)
template <int R, int MK>
__global__ __launch_bounds__(256) void k_cell_u16(unsigned* out, const float* pin, float, float, int steps) {
  unsigned H[R], p[R / 2];
  for (int r = 0; r < R; ++r) H[r] = 0u;
  for (int r = 0; r < R / 2; ++r) p[r] = ((threadIdx.x * 7 + r) & 3) == 0 ? 0x0003fffdu : 0xfffd0003u;
  unsigned up_prev = 0u, mx = 0u;
  unsigned gv = 2u; asm volatile("" : "+v"(gv));
  for (int t = 0; t < steps; t += 4) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned up = (unsigned)__builtin_amdgcn_update_dpp(0, (int)H[R - 1], 0x138, 0xf, 0xf, true);
      unsigned diag = up_prev, north = up;
      up_prev = up;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const unsigned w = H[r];
        unsigned x, tt, y, h;
        if (r & 1) { unsigned hi; asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(hi) : "v"(p[r >> 1])); asm volatile("v_add_u16 %0, %1, %2" : "=v"(x) : "v"(diag), "v"(hi)); }
        else asm volatile("v_add_u16 %0, %1, %2" : "=v"(x) : "v"(diag), "v"(p[r >> 1]));
        asm volatile("v_max_i16 %0, %1, %2" : "=v"(tt) : "v"(w), "v"(north));
        if ((MK == 1 || k == MK - 1) && (r & 1)) asm volatile("v_max_i16 %0, %0, %1" : "+v"(mx) : "v"(tt));
        asm volatile("v_sub_u16 %0, %1, %2 clamp" : "=v"(y) : "v"(tt), "v"(gv));
        asm volatile("v_max_i16 %0, %1, %2" : "=v"(h) : "v"(x), "v"(y));
        diag = w;
        H[r] = h;
        north = h;
      }
    }
  }
  unsigned s = mx;
  for (int r = 0; r < R; ++r) s += H[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

typedef void (*kfn)(unsigned*, unsigned, unsigned, unsigned);
typedef void (*cfn)(unsigned*, const float*, float, float, int);
int main() {
  unsigned* out; hipMalloc(&out, 256 * 8 * 256 * 4 * 4);
  float* pin; hipMalloc(&pin, 4096);
  float hp[1024]; for (int i = 0; i < 1024; ++i) hp[i] = ((i * 7) % 4 == 0 ? 3.f : -3.f) / 65536.f;
  hipMemcpy(pin, hp, 4096, hipMemcpyHostToDevice);
  struct E { const char* n; kfn f; } es[] = {
    {"v_add_f32 (VOP2)", k_addf32}, {"v_add_f32_e64 clamp", k_addf32c}, {"v_subrev_f32 sgpr", k_subs}, {"v_add_f32 sgpr", k_adds},
    {"v_max_f32", k_maxf32}, {"v_max3_f32", k_max3}, {"v_maximum3_f32", k_maximum3}, {"v_fma_mix_f32 (f16 src0)", k_fmamix},
    {"v_fma_mix_f32 clamp", k_fmamixc}, {"v_fma_mix_f32 hi clamp", k_fmamixh}, {"v_pk_maximum3_f16", k_pkmax3f16}, {"v_cvt_f32_f16", k_cvt},
    {"v_pk_add_f32 (2 cells)", k_pkaddf32}};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps : {1, 2, 4}) {
    printf("== %d waves/SIMD\n", wps);
    for (auto& e : es) {
      dim3 grid(256 * wps), block(256);
      hipLaunchKernelGGL(e.f, grid, block, 0, 0, out, 3u, 5u, 0x3c003c00u);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(e.f, grid, block, 0, 0, out, 3u, 5u, 0x3c003c00u);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%-28s %8.3f ms  %.2f cycles/wave-instr/SIMD @2.4GHz\n", e.n, ms, ms * 1e-3 * 2.4e9 / ((double)ITERS * 32 * wps));
    }
  }
  struct C { const char* n; cfn f; int R; } cs[] = {
    {"cell R=20 add-clamp/max3/sub, max every step", k_cell<20, 0, 1>, 20}, {"cell R=20 add-clamp/max3/sub, max every 4th", k_cell<20, 0, 4>, 20},
    {"cell R=20 fma_mix/max3/sub, max every 4th", k_cell<20, 1, 4>, 20}, {"cell R=20 add(VOP2)/max3/sub, max every 4th", k_cell<20, 2, 4>, 20},
    {"cell R=10 add-clamp/max3/sub, max every 4th", k_cell<10, 0, 4>, 10}, {"cell R=10 fma_mix/max3/sub, max every 4th", k_cell<10, 1, 4>, 10},
    {"cell R=32 add-clamp/max3/sub, max every 4th", k_cell<32, 0, 4>, 32}, {"cell R=32 fma_mix/max3/sub, max every 4th", k_cell<32, 1, 4>, 32},
    {"cell R=20 add-clamp/max3/sub(vgpr gap), 4th", k_cell_vg<20, false, 4>, 20}, {"cell R=32 add-clamp/max3/sub(vgpr gap), 4th", k_cell_vg<32, false, 4>, 32},
    {"cell R=10 add-clamp/max3/sub(vgpr gap), 4th", k_cell_vg<10, false, 4>, 10},
    {"cell R=20 fma_mix(f16 pairs)/max3/sub(vgpr), 4th", k_cell_vg<20, true, 4>, 20}, {"cell R=10 fma_mix(f16 pairs)/max3/sub(vgpr), 4th", k_cell_vg<10, true, 4>, 10},
    {"cell R=20 u16 add/max/subc/max, max every 4th", k_cell_u16<20, 4>, 20}, {"cell R=10 u16 add/max/subc/max, max every 4th", k_cell_u16<10, 4>, 10},
    {"cell R=32 u16 add/max/subc/max, max every 4th", k_cell_u16<32, 4>, 32}};
  const int steps = 16384;
  for (int wps : {1, 2, 4}) {
    printf("== cell loop, %d waves/SIMD\n", wps);
    for (auto& c : cs) {
      dim3 grid(256 * wps), block(256);
      hipLaunchKernelGGL(c.f, grid, block, 0, 0, out, pin, 2.f / 65536.f, 1.f / 65536.f, steps);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(c.f, grid, block, 0, 0, out, pin, 2.f / 65536.f, 1.f / 65536.f, steps);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double cyc_cell = ms * 1e-3 * 2.4e9 / ((double)steps * c.R * wps);
      printf("%-50s %8.3f ms  %.2f cycles/cell/lane  -> %.1f TCUPS chip ceiling\n", c.n, ms, cyc_cell, 1024 * 64 * 2.4e9 / cyc_cell * 1e-12);
    }
  }
  return 0;
}
