// uu3d_pk.h -- packed f32 VALU arithmetic written by name (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two floats per lane and
// instruction), for kernels whose elementwise work is VALU bound (the spatial stack; the softmax of attn_h3_kernel was tried and
// came out slower, docs/HISTORY.md E.11).
//
// The library is compiled with the packed-fp32-ops target feature OFF (docs/HISTORY.md E.12: a packed-f32 op whose op_sel reads
// the OTHER half of a register pair can lose that operand next to a busy matrix pipe, and hipcc chooses such forms on its own),
// which also makes the assembler refuse the instructions in inline asm.  A kernel that uses this header switches the feature
// back on for itself with UU3D_PK_TARGET; hipcc may then emit packed f32 in that kernel again, op_sel forms included --
// tests/test_isa_cpu.py checks that no packed instruction of those kernels carries op_sel / op_sel_hi, and that no other kernel
// has any.  None of the instructions below has one: both halves of every operand come from the same half of their pair.
//
// hipcc's hazard recognizer does not cover inline-asm READERS (section 12): results of MFMAs and of transcendental instructions
// must be fenced by hand before an op of this header reads them (pk::fence, pk::mfma_fence).
#pragma once
#include <hip/hip_runtime.h>

#if defined(__HIP_DEVICE_COMPILE__)
#define UU3D_PK_TARGET __attribute__((target("packed-fp32-ops")))
#else
#define UU3D_PK_TARGET
#endif

namespace uu3d {
typedef float f32x2 __attribute__((ext_vector_type(2)));
namespace pk {
__device__ __forceinline__ f32x2 add(const f32x2 a, const f32x2 b) { f32x2 d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f32x2 sub(const f32x2 a, const f32x2 b) { f32x2 d; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f32x2 mul(const f32x2 a, const f32x2 b) { f32x2 d; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f32x2 fma(const f32x2 a, const f32x2 b, const f32x2 c) { f32x2 d; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ f32x2 fnma(const f32x2 a, const f32x2 b, const f32x2 c) {          // c - a * b (one rounding)
    f32x2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d;
}
__device__ __forceinline__ f32x2 splat(const float v) { return (f32x2){v, v}; }
// A transcendental result (v_exp / v_rcp / v_rsq / v_sqrt) may not be read by the next non-transcendental VALU instruction
// (gfx940+: one wait state, software's to insert); hipcc covers its own instructions, not inline asm readers.  fence() sits
// between such results and the packed ops that read them.
__device__ __forceinline__ void fence(f32x2& a) { asm volatile("s_nop 0" : "+v"(a)); }
__device__ __forceinline__ void fence(f32x2& a, f32x2& b) { asm volatile("s_nop 0" : "+v"(a), "+v"(b)); }
// An MFMA result may not be read by a VALU instruction before the matrix pipe has written it back (8-pass MFMA: 11 wait states,
// 16-pass: 19); 20 wait states tied to the accumulators, once per batch of MFMAs (volatile asm statements keep their order, so
// "" : "+v"(x) statements placed after this one keep the readers of further accumulators behind it too).
template <class V> __device__ __forceinline__ void mfma_fence(V& a, V& b) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a), "+v"(b)); }
template <class V> __device__ __forceinline__ void behind_fence(V& a, V& b) { asm volatile("" : "+v"(a), "+v"(b)); }
}  // namespace pk
}  // namespace uu3d
