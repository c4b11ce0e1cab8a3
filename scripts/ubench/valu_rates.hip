// valu_rates.hip -- issue cost of the vector / scalar instructions the traversal and shading kernels are made of, on gfx950, at 1 and 8 waves per SIMD.
// Measurement tool (not part of libmi355pt.so): hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_rates.hip -o scripts/ubench/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#define REP 64
#define STR2(x) #x
#define STR(x) STR2(x)
// 8 independent destination registers v8..v15, sources v0..v7 (initialised from the lane id so nothing is constant-folded: all inline asm)
#define BODY8(I0, I1, I2, I3, I4, I5, I6, I7) I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7 "\n"
#define KERNEL(NAME, ...) \
__global__ __launch_bounds__(256) void NAME(float *out, int iters) { \
    float a = threadIdx.x * 0.001f + 1.0f; \
    asm volatile("v_mov_b32 v0, %0\nv_mov_b32 v1, %0\nv_mov_b32 v2, %0\nv_mov_b32 v3, %0\nv_mov_b32 v4, %0\nv_mov_b32 v5, %0\nv_mov_b32 v6, %0\nv_mov_b32 v7, %0\n" \
                 "v_mov_b32 v8, %0\nv_mov_b32 v9, %0\nv_mov_b32 v10, %0\nv_mov_b32 v11, %0\nv_mov_b32 v12, %0\nv_mov_b32 v13, %0\nv_mov_b32 v14, %0\nv_mov_b32 v15, %0\nv_mov_b32 v16, %0\nv_mov_b32 v17, %0\nv_mov_b32 v18, %0\nv_mov_b32 v19, %0\nv_mov_b32 v20, %0\nv_mov_b32 v21, %0\nv_mov_b32 v22, %0\nv_mov_b32 v23, %0\n" \
                 "s_mov_b64 s[20:21], exec\ns_mov_b64 s[22:23], exec\n" :: "v"(a) : "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","s20","s21","s22","s23"); \
    for (int i = 0; i < iters; ++i) { \
        asm volatile(__VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ ::: "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","vcc","s20","s21","s22","s23","s24","s25","s26","s27"); \
    } \
    float r; asm volatile("v_add_f32 %0, v8, v9\nv_add_f32 %0, %0, v10\nv_add_f32 %0, %0, v16" : "=v"(r) :: "v8","v9","v10","v16"); \
    if (r == 12345.678f) out[threadIdx.x] = r; \
}
KERNEL(k_fma, BODY8("v_fma_f32 v8, v0, v1, v8", "v_fma_f32 v9, v1, v2, v9", "v_fma_f32 v10, v2, v3, v10", "v_fma_f32 v11, v3, v4, v11", "v_fma_f32 v12, v4, v5, v12", "v_fma_f32 v13, v5, v6, v13", "v_fma_f32 v14, v6, v7, v14", "v_fma_f32 v15, v7, v0, v15"))
KERNEL(k_add, BODY8("v_add_f32 v8, v0, v8", "v_add_f32 v9, v1, v9", "v_add_f32 v10, v2, v10", "v_add_f32 v11, v3, v11", "v_add_f32 v12, v4, v12", "v_add_f32 v13, v5, v13", "v_add_f32 v14, v6, v14", "v_add_f32 v15, v7, v15"))
KERNEL(k_max3, BODY8("v_max3_f32 v8, v0, v1, v8", "v_max3_f32 v9, v1, v2, v9", "v_max3_f32 v10, v2, v3, v10", "v_max3_f32 v11, v3, v4, v11", "v_max3_f32 v12, v4, v5, v12", "v_max3_f32 v13, v5, v6, v13", "v_max3_f32 v14, v6, v7, v14", "v_max3_f32 v15, v7, v0, v15"))
KERNEL(k_cvt_ubyte, BODY8("v_cvt_f32_ubyte0 v8, v0", "v_cvt_f32_ubyte1 v9, v1", "v_cvt_f32_ubyte2 v10, v2", "v_cvt_f32_ubyte3 v11, v3", "v_cvt_f32_ubyte0 v12, v4", "v_cvt_f32_ubyte1 v13, v5", "v_cvt_f32_ubyte2 v14, v6", "v_cvt_f32_ubyte3 v15, v7"))
KERNEL(k_cndmask_vcc, BODY8("v_cndmask_b32 v8, v0, v1, vcc", "v_cndmask_b32 v9, v1, v2, vcc", "v_cndmask_b32 v10, v2, v3, vcc", "v_cndmask_b32 v11, v3, v4, vcc", "v_cndmask_b32 v12, v4, v5, vcc", "v_cndmask_b32 v13, v5, v6, vcc", "v_cndmask_b32 v14, v6, v7, vcc", "v_cndmask_b32 v15, v7, v0, vcc"))
KERNEL(k_cndmask_sgpr, BODY8("v_cndmask_b32 v8, v0, v1, s[20:21]", "v_cndmask_b32 v9, v1, v2, s[22:23]", "v_cndmask_b32 v10, v2, v3, s[20:21]", "v_cndmask_b32 v11, v3, v4, s[22:23]", "v_cndmask_b32 v12, v4, v5, s[20:21]", "v_cndmask_b32 v13, v5, v6, s[22:23]", "v_cndmask_b32 v14, v6, v7, s[20:21]", "v_cndmask_b32 v15, v7, v0, s[22:23]"))
KERNEL(k_cmp_vcc, BODY8("v_cmp_lt_f32 vcc, v0, v1", "v_cmp_lt_f32 vcc, v1, v2", "v_cmp_lt_f32 vcc, v2, v3", "v_cmp_lt_f32 vcc, v3, v4", "v_cmp_lt_f32 vcc, v4, v5", "v_cmp_lt_f32 vcc, v5, v6", "v_cmp_lt_f32 vcc, v6, v7", "v_cmp_lt_f32 vcc, v7, v0"))
KERNEL(k_cmp_sgpr, BODY8("v_cmp_lt_f32 s[20:21], v0, v1", "v_cmp_lt_f32 s[22:23], v1, v2", "v_cmp_lt_f32 s[24:25], v2, v3", "v_cmp_lt_f32 s[26:27], v3, v4", "v_cmp_lt_f32 s[20:21], v4, v5", "v_cmp_lt_f32 s[22:23], v5, v6", "v_cmp_lt_f32 s[24:25], v6, v7", "v_cmp_lt_f32 s[26:27], v7, v0"))
KERNEL(k_cmp_cnd_pair, BODY8("v_cmp_lt_f32 vcc, v0, v1", "v_cndmask_b32 v8, v0, v1, vcc", "v_cmp_lt_f32 vcc, v2, v3", "v_cndmask_b32 v9, v2, v3, vcc", "v_cmp_lt_f32 vcc, v4, v5", "v_cndmask_b32 v10, v4, v5, vcc", "v_cmp_lt_f32 vcc, v6, v7", "v_cndmask_b32 v11, v6, v7, vcc"))
KERNEL(k_cmp_u64, BODY8("v_cmp_lt_u64 vcc, v[0:1], v[2:3]", "v_cmp_lt_u64 vcc, v[2:3], v[4:5]", "v_cmp_lt_u64 vcc, v[4:5], v[6:7]", "v_cmp_lt_u64 vcc, v[6:7], v[0:1]", "v_cmp_lt_u64 vcc, v[0:1], v[4:5]", "v_cmp_lt_u64 vcc, v[2:3], v[6:7]", "v_cmp_lt_u64 vcc, v[4:5], v[0:1]", "v_cmp_lt_u64 vcc, v[6:7], v[2:3]"))
KERNEL(k_minmax_u32, BODY8("v_min_u32 v8, v0, v8", "v_max_u32 v9, v1, v9", "v_min_u32 v10, v2, v10", "v_max_u32 v11, v3, v11", "v_min_u32 v12, v4, v12", "v_max_u32 v13, v5, v13", "v_min_u32 v14, v6, v14", "v_max_u32 v15, v7, v15"))
KERNEL(k_and_or, BODY8("v_and_or_b32 v8, v0, v1, v8", "v_and_or_b32 v9, v1, v2, v9", "v_and_or_b32 v10, v2, v3, v10", "v_and_or_b32 v11, v3, v4, v11", "v_and_or_b32 v12, v4, v5, v12", "v_and_or_b32 v13, v5, v6, v13", "v_and_or_b32 v14, v6, v7, v14", "v_and_or_b32 v15, v7, v0, v15"))
KERNEL(k_bfi, BODY8("v_bfi_b32 v8, v0, v1, v8", "v_bfi_b32 v9, v1, v2, v9", "v_bfi_b32 v10, v2, v3, v10", "v_bfi_b32 v11, v3, v4, v11", "v_bfi_b32 v12, v4, v5, v12", "v_bfi_b32 v13, v5, v6, v13", "v_bfi_b32 v14, v6, v7, v14", "v_bfi_b32 v15, v7, v0, v15"))
KERNEL(k_rcp, BODY8("v_rcp_f32 v8, v0", "v_rcp_f32 v9, v1", "v_rcp_f32 v10, v2", "v_rcp_f32 v11, v3", "v_rcp_f32 v12, v4", "v_rcp_f32 v13, v5", "v_rcp_f32 v14, v6", "v_rcp_f32 v15, v7"))
KERNEL(k_pk_fma, BODY8("v_pk_fma_f32 v[8:9], v[0:1], v[2:3], v[8:9]", "v_pk_fma_f32 v[10:11], v[2:3], v[4:5], v[10:11]", "v_pk_fma_f32 v[12:13], v[4:5], v[6:7], v[12:13]", "v_pk_fma_f32 v[14:15], v[6:7], v[0:1], v[14:15]", "v_pk_fma_f32 v[16:17], v[0:1], v[4:5], v[16:17]", "v_pk_fma_f32 v[18:19], v[2:3], v[6:7], v[18:19]", "v_pk_fma_f32 v[20:21], v[4:5], v[0:1], v[20:21]", "v_pk_fma_f32 v[22:23], v[6:7], v[2:3], v[22:23]"))
KERNEL(k_pk_mul, BODY8("v_pk_mul_f32 v[8:9], v[0:1], v[2:3]", "v_pk_mul_f32 v[10:11], v[2:3], v[4:5]", "v_pk_mul_f32 v[12:13], v[4:5], v[6:7]", "v_pk_mul_f32 v[14:15], v[6:7], v[0:1]", "v_pk_mul_f32 v[16:17], v[0:1], v[4:5]", "v_pk_mul_f32 v[18:19], v[2:3], v[6:7]", "v_pk_mul_f32 v[20:21], v[4:5], v[0:1]", "v_pk_mul_f32 v[22:23], v[6:7], v[2:3]"))
KERNEL(k_fmamk, BODY8("v_fmamk_f32 v8, v0, 0x3f800011, v1", "v_fmamk_f32 v9, v1, 0x3f800011, v2", "v_fmamk_f32 v10, v2, 0x3f800011, v3", "v_fmamk_f32 v11, v3, 0x3f800011, v4", "v_fmamk_f32 v12, v4, 0x3f800011, v5", "v_fmamk_f32 v13, v5, 0x3f800011, v6", "v_fmamk_f32 v14, v6, 0x3f800011, v7", "v_fmamk_f32 v15, v7, 0x3f800011, v0"))
KERNEL(k_salu, BODY8("s_and_b64 s[24:25], s[20:21], s[22:23]", "s_or_b64 s[26:27], s[20:21], s[22:23]", "s_and_b64 s[24:25], s[26:27], s[22:23]", "s_or_b64 s[26:27], s[24:25], s[22:23]", "s_and_b64 s[24:25], s[26:27], s[22:23]", "s_or_b64 s[26:27], s[20:21], s[24:25]", "s_and_b64 s[24:25], s[20:21], s[26:27]", "s_or_b64 s[26:27], s[24:25], s[22:23]"))
KERNEL(k_valu_salu_mix, BODY8("v_fma_f32 v8, v0, v1, v8", "s_and_b64 s[24:25], s[20:21], s[22:23]", "v_fma_f32 v9, v1, v2, v9", "s_or_b64 s[26:27], s[20:21], s[22:23]", "v_fma_f32 v10, v2, v3, v10", "s_and_b64 s[24:25], s[20:21], s[22:23]", "v_fma_f32 v11, v3, v4, v11", "s_or_b64 s[26:27], s[20:21], s[22:23]"))
KERNEL(k_saveexec, BODY8("s_and_saveexec_b64 s[24:25], s[20:21]", "v_fma_f32 v8, v0, v1, v8", "s_or_b64 exec, exec, s[24:25]", "v_fma_f32 v9, v1, v2, v9", "s_and_saveexec_b64 s[26:27], s[22:23]", "v_fma_f32 v10, v2, v3, v10", "s_or_b64 exec, exec, s[26:27]", "v_fma_f32 v11, v3, v4, v11"))
KERNEL(k_div_seq, BODY8("v_div_scale_f32 v8, vcc, v0, v1, v0", "v_rcp_f32 v9, v1", "v_fma_f32 v10, v1, v9, v2", "v_fma_f32 v11, v10, v9, v9", "v_mul_f32 v12, v8, v11", "v_fma_f32 v13, v1, v12, v8", "v_div_fmas_f32 v14, v13, v11, v12", "v_div_fixup_f32 v15, v14, v1, v0"))
KERNEL(k_dep_fma, BODY8("v_fma_f32 v8, v8, v1, v0", "v_fma_f32 v8, v8, v1, v0", "v_fma_f32 v8, v8, v1, v0", "v_fma_f32 v8, v8, v1, v0", "v_fma_f32 v8, v8, v1, v0", "v_fma_f32 v8, v8, v1, v0", "v_fma_f32 v8, v8, v1, v0", "v_fma_f32 v8, v8, v1, v0"))
KERNEL(k_fma_mix, BODY8("v_fma_mix_f32 v8, v0, v1, v8 op_sel_hi:[1,0,0]", "v_fma_mix_f32 v9, v1, v2, v9 op_sel:[1,0,0] op_sel_hi:[1,0,0]", "v_fma_mix_f32 v10, v2, v3, v10 op_sel_hi:[1,0,0]", "v_fma_mix_f32 v11, v3, v4, v11 op_sel:[1,0,0] op_sel_hi:[1,0,0]", "v_fma_mix_f32 v12, v4, v5, v12 op_sel_hi:[1,0,0]", "v_fma_mix_f32 v13, v5, v6, v13 op_sel:[1,0,0] op_sel_hi:[1,0,0]", "v_fma_mix_f32 v14, v6, v7, v14 op_sel_hi:[1,0,0]", "v_fma_mix_f32 v15, v7, v0, v15 op_sel:[1,0,0] op_sel_hi:[1,0,0]"))
KERNEL(k_max_f32, BODY8("v_max_f32 v8, v0, v8", "v_min_f32 v9, v1, v9", "v_max_f32 v10, v2, v10", "v_min_f32 v11, v3, v11", "v_max_f32 v12, v4, v12", "v_min_f32 v13, v5, v13", "v_max_f32 v14, v6, v14", "v_min_f32 v15, v7, v15"))
KERNEL(k_mul_f32, BODY8("v_mul_f32 v8, v0, v1", "v_mul_f32 v9, v1, v2", "v_mul_f32 v10, v2, v3", "v_mul_f32 v11, v3, v4", "v_mul_f32 v12, v4, v5", "v_mul_f32 v13, v5, v6", "v_mul_f32 v14, v6, v7", "v_mul_f32 v15, v7, v0"))
KERNEL(k_med3, BODY8("v_med3_f32 v8, v0, v1, v8", "v_med3_f32 v9, v1, v2, v9", "v_med3_f32 v10, v2, v3, v10", "v_med3_f32 v11, v3, v4, v11", "v_med3_f32 v12, v4, v5, v12", "v_med3_f32 v13, v5, v6, v13", "v_med3_f32 v14, v6, v7, v14", "v_med3_f32 v15, v7, v0, v15"))
KERNEL(k_perm, BODY8("v_perm_b32 v8, v0, v1, v2", "v_perm_b32 v9, v1, v2, v3", "v_perm_b32 v10, v2, v3, v4", "v_perm_b32 v11, v3, v4, v5", "v_perm_b32 v12, v4, v5, v6", "v_perm_b32 v13, v5, v6, v7", "v_perm_b32 v14, v6, v7, v0", "v_perm_b32 v15, v7, v0, v1"))
KERNEL(k_int_ops, BODY8("v_add_u32 v8, v0, v8", "v_lshl_add_u32 v9, v1, 2, v9", "v_and_b32 v10, v2, v10", "v_lshlrev_b32 v11, 3, v3", "v_bfe_u32 v12, v4, 8, 8", "v_sub_u32 v13, v5, v13", "v_or_b32 v14, v6, v14", "v_xor_b32 v15, v7, v15"))
KERNEL(k_mov, BODY8("v_mov_b32 v8, v0", "v_mov_b32 v9, v1", "v_mov_b32 v10, v2", "v_mov_b32 v11, v3", "v_mov_b32 v12, v4", "v_mov_b32 v13, v5", "v_mov_b32 v14, v6", "v_mov_b32 v15, v7"))
KERNEL(k_cvt_f16, BODY8("v_cvt_f32_f16 v8, v0", "v_cvt_f32_f16 v9, v1", "v_cvt_f32_f16 v10, v2", "v_cvt_f32_f16 v11, v3", "v_cvt_f32_f16 v12, v4", "v_cvt_f32_f16 v13, v5", "v_cvt_f32_f16 v14, v6", "v_cvt_f32_f16 v15, v7"))
KERNEL(k_pk_fma_f16, BODY8("v_pk_fma_f16 v8, v0, v1, v8", "v_pk_fma_f16 v9, v1, v2, v9", "v_pk_fma_f16 v10, v2, v3, v10", "v_pk_fma_f16 v11, v3, v4, v11", "v_pk_fma_f16 v12, v4, v5, v12", "v_pk_fma_f16 v13, v5, v6, v13", "v_pk_fma_f16 v14, v6, v7, v14", "v_pk_fma_f16 v15, v7, v0, v15"))
KERNEL(k_pk_max_f16, BODY8("v_pk_max_f16 v8, v0, v8", "v_pk_min_f16 v9, v1, v9", "v_pk_max_f16 v10, v2, v10", "v_pk_min_f16 v11, v3, v11", "v_pk_max_f16 v12, v4, v12", "v_pk_min_f16 v13, v5, v13", "v_pk_max_f16 v14, v6, v14", "v_pk_min_f16 v15, v7, v15"))
KERNEL(k_cmp_addc, BODY8("v_cmp_ne_u32 vcc, v0, v1", "v_addc_co_u32 v8, vcc, 0, v8, vcc", "v_cmp_ne_u32 vcc, v2, v3", "v_addc_co_u32 v9, vcc, 0, v9, vcc", "v_cmp_ne_u32 vcc, v4, v5", "v_addc_co_u32 v10, vcc, 0, v10, vcc", "v_cmp_ne_u32 vcc, v6, v7", "v_addc_co_u32 v11, vcc, 0, v11, vcc"))
KERNEL(k_fma_cvt_mix, BODY8("v_fma_f32 v8, v0, v1, v8", "v_cvt_f32_ubyte0 v16, v2", "v_fma_f32 v9, v1, v2, v9", "v_cvt_f32_ubyte1 v17, v3", "v_fma_f32 v10, v2, v3, v10", "v_cvt_f32_ubyte2 v18, v4", "v_fma_f32 v11, v3, v4, v11", "v_cvt_f32_ubyte3 v19, v5"))

typedef void (*KernelT)(float *, int);
int main() {
    struct E { const char *name; KernelT k; } tests[] = {{"v_fma_f32", k_fma}, {"v_add_f32 (VOP2)", k_add}, {"v_mul_f32", k_mul_f32}, {"v_max3_f32", k_max3}, {"v_max/min_f32", k_max_f32}, {"v_med3_f32", k_med3}, {"v_cvt_f32_ubyteN", k_cvt_ubyte}, {"v_cvt_f32_f16", k_cvt_f16}, {"v_fma_mix_f32 (f16 src0)", k_fma_mix}, {"fma + cvt_ubyte alternating (per instr)", k_fma_cvt_mix},
        {"v_cndmask sgpr pair", k_cndmask_sgpr},
        {"v_cmp -> vcc", k_cmp_vcc}, {"v_cmp -> sgpr pair", k_cmp_sgpr}, {"v_cmp + v_cndmask (dependent pair)", k_cmp_cnd_pair}, {"v_cmp + v_addc (dependent pair)", k_cmp_addc}, {"v_cmp_lt_u64", k_cmp_u64}, {"v_min/max_u32", k_minmax_u32}, {"v_and_or_b32", k_and_or}, {"v_bfi_b32", k_bfi}, {"v_perm_b32", k_perm}, {"int add/lshl_add/and/lshl/bfe/sub/or/xor", k_int_ops}, {"v_mov_b32", k_mov}, {"v_rcp_f32", k_rcp},
        {"v_pk_fma_f32 (2 fma)", k_pk_fma}, {"v_pk_mul_f32", k_pk_mul}, {"v_pk_fma_f16", k_pk_fma_f16}, {"v_pk_max/min_f16", k_pk_max_f16}, {"v_fmamk_f32 (literal)", k_fmamk}, {"s_and/or_b64", k_salu}, {"v_fma + s_and interleaved (per instr)", k_valu_salu_mix}, {"saveexec + fma + restore + fma (per instr)", k_saveexec}, {"IEEE divide sequence (8 instr)", k_div_seq}, {"dependent v_fma chain", k_dep_fma}};
    float *out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000; hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); const int cus = p.multiProcessorCount;
    printf("%d CUs, clock %d kHz (nominal); ns per wave-instruction per SIMD (and cycles at 2.4 GHz)\n", cus, p.clockRate);
    for (int wps : {1, 4, 8}) {
        printf("-- %d wave(s) per SIMD\n", wps);
        for (auto &t : tests) {
            const int blocks = cus * wps;      // 256 threads = 4 waves = one per SIMD
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 10);
            hipError_t err = hipDeviceSynchronize(); if (err != hipSuccess) { printf("   %-42s FAILED: %s\n", t.name, hipGetErrorString(err)); return 1; }
            hipEventRecord(e0); hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double instr = (double) iters * 64.0 * wps;      // per SIMD
            const double ns = ms * 1e6 / instr;
            printf("   %-42s %6.3f ns  = %5.2f cycles per instruction\n", t.name, ns, ns * 2.4);
        }
    }
    return 0;
}
