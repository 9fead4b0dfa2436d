// Device-side pieces shared by the fp32 NN kernel files (nn_kernels.hip, nn_gemm_dma.hip): vector types and the fused
// epilogue (activation / LearnableAffineBlock resolved once per kernel).
#pragma once
#include "nn.h"

namespace rt {
namespace nn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// `act` is uniform per launch.  Every case is a handful of VALU ops: no IEEE division (the
// x/6 of hardswish is a multiply, swish/sigmoid use v_exp + v_rcp), so the fused epilogues stay
// cheap next to the MFMA / load work.  Differences to the exact forms are <= 2 ulp.
__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == ACT_HSWISH) return v * fminf(fmaxf(v + 3.0f, 0.0f), 6.0f) * 0.16666667f;
  if (act == ACT_RELU) return fmaxf(v, 0.0f);
  if (act == ACT_NONE) return v;
  const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-v));
  return act == ACT_SWISH ? v * s : s;
}

// The epilogues resolve (activation, LAB) ONCE per kernel instead of per element: with the runtime
// `act` inside the element loops hipcc emits a branch ladder per output value (the 256 x 240 GEMM tile
// spent ~16 us of its ~80 us there).  act_dispatch calls f with compile-time tags for the combinations
// the networks use and with (-1, -1) = "decide per element" for anything else.  Same expressions, same
// rounding as act_apply.
template <int V> struct IntTag { static constexpr int value = V; };
template <int ACT, int LAB>
__device__ __forceinline__ float epi_val(float v, int act, int has_lab, float lab_a, float lab_c) {
  float t;
  if (ACT == ACT_HSWISH) t = v * fminf(fmaxf(v + 3.0f, 0.0f), 6.0f) * 0.16666667f;
  else if (ACT == ACT_RELU) t = fmaxf(v, 0.0f);
  else if (ACT == ACT_NONE) t = v;
  else if (ACT == ACT_SWISH) t = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
  else t = act_apply(v, act);
  if (LAB == 1 || (LAB < 0 && has_lab)) t = fmaf(t, lab_a, lab_c);
  return t;
}
template <class F>
__device__ __forceinline__ void act_dispatch(int act, int has_lab, bool has_res, F&& f) {
  // f(activation tag, LAB tag, residual tag); (-1, -1, 1) = everything decided per element
  if (has_res) f(IntTag<-1>{}, IntTag<-1>{}, IntTag<1>{});
  else if (act == ACT_HSWISH && has_lab) f(IntTag<ACT_HSWISH>{}, IntTag<1>{}, IntTag<0>{});
  else if (act == ACT_HSWISH) f(IntTag<ACT_HSWISH>{}, IntTag<0>{}, IntTag<0>{});
  else if (act == ACT_NONE && !has_lab) f(IntTag<ACT_NONE>{}, IntTag<0>{}, IntTag<0>{});
  else if (act == ACT_RELU && !has_lab) f(IntTag<ACT_RELU>{}, IntTag<0>{}, IntTag<0>{});
  else if (act == ACT_SWISH && !has_lab) f(IntTag<ACT_SWISH>{}, IntTag<0>{}, IntTag<0>{});
  else f(IntTag<-1>{}, IntTag<-1>{}, IntTag<0>{});
}

}  // namespace nn
}  // namespace rt
