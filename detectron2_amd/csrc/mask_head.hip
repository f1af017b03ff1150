// Mask-head glue either side of the mask ROIAlign / paste_masks_in_image.
//   replaces  detectron2/modeling/roi_heads/mask_head.py:31-113 (mask_rcnn_loss: advanced-index gather
//             pred_mask_logits[arange(B), gt_classes], bool -> fp32 copy of the targets, three comparison
//             passes each followed by a host-synchronising .item(), binary_cross_entropy_with_logits) and
//             mask_head.py:116-158 (mask_rcnn_inference: the same gather + sigmoid).
// Here: the class plane of every ROI is read in place (no gathered copy, no fp32 targets), the loss and the
// four training statistics come out of ONE pass (per-ROI partials, then a fixed-order final reduction:
// deterministic) and stay on the device (no .item()), and the backward writes the whole
// (B, C, M, M) gradient exactly once: zeros outside the class plane, (sigmoid(x) - t) g / (B M M) inside
// (the reference's autograd zero-fills and then index_put_s).
// Roofline: HBM.  Forward reads B*HW*(s + 1) bytes, backward writes B*C*HW*s bytes (32 MB for 256 x 80 x
// 28 x 28 bf16) and reads the class planes.
#include "common.h"

namespace d2amd {

struct MaskPartial { float loss; int incorrect, positive, false_pos, false_neg, bad; };

// log(sigmoid(x)) as ATen evaluates it: min(x, 0) - log1p(exp(-|x|))
__device__ __forceinline__ float log_sigmoid_f(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// MASKED: rows whose class is outside [0, C) are IGNORED (no loss, no statistics; `bad` then counts the ignored rows)
// instead of reported -- the fixed-shape lists of d2amd_label_and_sample_proposals carry background / padding rows
template <typename T, bool MASKED>
__global__ __launch_bounds__(256) void mask_loss_partial_kernel(const T* __restrict__ logits,
                                                               const int64_t* __restrict__ cls,
                                                               const uint8_t* __restrict__ gt, int C, int HW,
                                                               MaskPartial* __restrict__ part) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const long c = cls ? cls[b] : 0;
  const bool bad = c < 0 || c >= C;
  const T* x = logits + ((long)b * C + (bad ? 0 : c)) * HW;
  const uint8_t* t = gt + (long)b * HW;
  float loss = 0.f;
  int inc = 0, pos = 0, fp = 0, fn = 0;
  for (int i = tid; i < HW; i += 256) {
    const float xv = to_f32(x[i]);
    const bool tv = t[i] != 0;
    // binary_cross_entropy_with_logits: (1 - t) * x - log_sigmoid(x)
    loss += (tv ? 0.f : xv) - log_sigmoid_f(xv);
    const bool wrong = (xv > 0.f) != tv;  // mask_head.py:89
    inc += wrong;
    pos += tv;
    fp += wrong && !tv;
    fn += wrong && tv;
  }
  __shared__ float sl[256];
  __shared__ int si[4][256];
  sl[tid] = loss; si[0][tid] = inc; si[1][tid] = pos; si[2][tid] = fp; si[3][tid] = fn;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {  // fixed tree: deterministic
    if (tid < s) {
      sl[tid] += sl[tid + s];
#pragma unroll
      for (int q = 0; q < 4; q++) si[q][tid] += si[q][tid + s];
    }
    __syncthreads();
  }
  if (tid == 0) {
    if (MASKED && bad) part[b] = MaskPartial{0.f, 0, 0, 0, 0, 1};
    else part[b] = MaskPartial{bad ? 0.f : sl[0], si[0][0], si[1][0], si[2][0], si[3][0], bad ? 1 : 0};
  }
}

// MASKED: mean over the rows that count (B - ignored; 0 rows -> loss 0), stats_out[5] = that number of rows
template <bool MASKED>
__global__ __launch_bounds__(256) void mask_loss_final_kernel(const MaskPartial* __restrict__ part, int B, int HW,
                                                             float* __restrict__ loss_out,
                                                             int64_t* __restrict__ stats_out) {
  const int tid = threadIdx.x;
  double loss = 0.;
  long st[5] = {0, 0, 0, 0, 0};
  for (int b = tid; b < B; b += 256) {
    const MaskPartial p = part[b];
    loss += (double)p.loss;
    st[0] += p.incorrect; st[1] += p.positive; st[2] += p.false_pos; st[3] += p.false_neg; st[4] += p.bad;
  }
  __shared__ double sl[256];
  __shared__ long si[5][256];
  sl[tid] = loss;
#pragma unroll
  for (int q = 0; q < 5; q++) si[q][tid] = st[q];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      sl[tid] += sl[tid + s];
#pragma unroll
      for (int q = 0; q < 5; q++) si[q][tid] += si[q][tid + s];
    }
    __syncthreads();
  }
  if (tid == 0) {
    if (MASKED) {
      const long rows = (long)B - si[4][0];
      *loss_out = rows > 0 ? (float)(sl[0] / ((double)rows * (double)HW)) : 0.f;
      stats_out[5] = rows;
    } else {
      *loss_out = (float)(sl[0] / ((double)B * (double)HW));  // reduction="mean"
    }
#pragma unroll
    for (int q = 0; q < 5; q++) stats_out[q] = si[q][0];
  }
}

// one thread per VEC consecutive elements (16 B) of the gradient; HW % VEC == 0, so a vector lies in one plane
template <typename T, int VEC>
__global__ __launch_bounds__(256) void mask_loss_backward_kernel(const T* __restrict__ logits,
                                                                const int64_t* __restrict__ cls,
                                                                const uint8_t* __restrict__ gt,
                                                                const float* __restrict__ grad_loss, int C, int HW,
                                                                long nvec, float inv_n, T* __restrict__ grad) {
  const long v = (long)blockIdx.x * 256 + threadIdx.x;
  if (v >= nvec) return;
  const int per_plane = HW / VEC;
  const long plane = v / per_plane;
  const int i0 = (int)(v - plane * per_plane) * VEC;
  const long b = plane / C;
  const int c = (int)(plane - b * C);
  const long want = cls ? cls[b] : 0;
  struct __attribute__((aligned(VEC * sizeof(T)))) Pack { T e[VEC]; };
  Pack o;
  if (want == c) {
    const float g = *grad_loss * inv_n;
#pragma unroll
    for (int q = 0; q < VEC; q++) {
      const float xv = to_f32(logits[plane * HW + i0 + q]);
      const float tv = gt[b * HW + i0 + q] ? 1.f : 0.f;
      o.e[q] = from_f32<T>((sigmoid_f(xv) - tv) * g);
    }
  } else {
#pragma unroll
    for (int q = 0; q < VEC; q++) o.e[q] = from_f32<T>(0.f);
  }
  *reinterpret_cast<Pack*>(grad + plane * HW + i0) = o;
}

// the same with the mean taken over `*rows` rows, a number only the device knows (stats_out[5] of the masked forward);
// rows of an ignored class get zeros like any other plane that is not the row's class plane
template <typename T, int VEC>
__global__ __launch_bounds__(256) void mask_loss_backward_masked_kernel(const T* __restrict__ logits,
                                                                       const int64_t* __restrict__ cls,
                                                                       const uint8_t* __restrict__ gt,
                                                                       const float* __restrict__ grad_loss,
                                                                       const int64_t* __restrict__ rows, int C, int HW,
                                                                       long nvec, T* __restrict__ grad) {
  const long v = (long)blockIdx.x * 256 + threadIdx.x;
  if (v >= nvec) return;
  const int per_plane = HW / VEC;
  const long plane = v / per_plane;
  const int i0 = (int)(v - plane * per_plane) * VEC;
  const long b = plane / C;
  const int c = (int)(plane - b * C);
  const long want = cls ? cls[b] : 0;
  struct __attribute__((aligned(VEC * sizeof(T)))) Pack { T e[VEC]; };
  Pack o;
  if (want == c) {
    const long r = *rows;
    const float inv_n = r > 0 ? (float)(1.0 / ((double)r * (double)HW)) : 0.f;  // (the host's expression, on the device)
    const float g = *grad_loss * inv_n;
#pragma unroll
    for (int q = 0; q < VEC; q++) {
      const float xv = to_f32(logits[plane * HW + i0 + q]);
      const float tv = gt[b * HW + i0 + q] ? 1.f : 0.f;
      o.e[q] = from_f32<T>((sigmoid_f(xv) - tv) * g);
    }
  } else {
#pragma unroll
    for (int q = 0; q < VEC; q++) o.e[q] = from_f32<T>(0.f);
  }
  *reinterpret_cast<Pack*>(grad + plane * HW + i0) = o;
}

template <typename T>
__global__ __launch_bounds__(256) void mask_inference_kernel(const T* __restrict__ logits,
                                                            const int64_t* __restrict__ cls, int C, int HW, long n,
                                                            T* __restrict__ out) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const long b = e / HW;
  const int i = (int)(e - b * HW);
  const long c = cls ? cls[b] : 0;
  float r;
  if (c < 0 || c >= C) r = __uint_as_float(0x7fc00000u);  // the reference's gather raises; here: NaN, no sync
  else r = sigmoid_f(to_f32(logits[(b * C + c) * HW + i]));
  out[e] = from_f32<T>(r);
}

}  // namespace d2amd

using namespace d2amd;

extern "C" int d2amd_mask_rcnn_inference(const void* logits, const int64_t* classes, int B, int C, int HW, int dtype,
                                         void* out, void* stream) {
  D2_CHECK_ARG(B >= 0 && C > 0 && HW > 0, "mask_rcnn_inference: bad shape");
  D2_CHECK_ARG(classes != nullptr || C == 1, "mask_rcnn_inference: class-specific logits (C = %d) need classes", C);
  if (B == 0) return D2AMD_OK;
  D2_CHECK_ARG(logits && out, "mask_rcnn_inference: null pointer");
  const long n = (long)B * HW, nblk = (n + 255) / 256;
  D2_CHECK_ARG(nblk < (1l << 31), "mask_rcnn_inference: tensor too large");
  return D2_DISPATCH_DTYPE(dtype, [&]() -> int {
    hipLaunchKernelGGL(mask_inference_kernel<scalar_t>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                       (const scalar_t*)logits, classes, C, HW, n, (scalar_t*)out);
    D2_LAUNCH_OK();
    return D2AMD_OK;
  });
}

extern "C" size_t d2amd_mask_rcnn_loss_workspace_bytes(int B) { return (size_t)(B > 0 ? B : 1) * sizeof(MaskPartial); }

template <bool MASKED>
static int mask_loss_forward_impl(const void* logits, const int64_t* gt_classes, const uint8_t* gt_masks, int B, int C,
                                  int HW, int dtype, float* loss_out, int64_t* stats_out, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  D2_CHECK_ARG(B > 0 && C > 0 && HW > 0, "mask_rcnn_loss_forward: bad shape (the empty case is the caller's)");
  D2_CHECK_ARG(gt_classes != nullptr || C == 1, "mask_rcnn_loss_forward: class-specific logits need gt_classes");
  D2_CHECK_ARG(logits && gt_masks && loss_out && stats_out, "mask_rcnn_loss_forward: null pointer");
  if (!workspace || workspace_bytes < d2amd_mask_rcnn_loss_workspace_bytes(B)) {
    set_error("mask_rcnn_loss_forward: workspace too small");
    return D2AMD_EWORKSPACE;
  }
  MaskPartial* part = (MaskPartial*)workspace;
  hipStream_t s = (hipStream_t)stream;
  return D2_DISPATCH_DTYPE(dtype, [&]() -> int {
    hipLaunchKernelGGL((mask_loss_partial_kernel<scalar_t, MASKED>), dim3(B), dim3(256), 0, s, (const scalar_t*)logits,
                       gt_classes, gt_masks, C, HW, part);
    D2_LAUNCH_OK();
    hipLaunchKernelGGL(mask_loss_final_kernel<MASKED>, dim3(1), dim3(256), 0, s, part, B, HW, loss_out, stats_out);
    D2_LAUNCH_OK();
    return D2AMD_OK;
  });
}

extern "C" int d2amd_mask_rcnn_loss_forward(const void* logits, const int64_t* gt_classes, const uint8_t* gt_masks,
                                            int B, int C, int HW, int dtype, float* loss_out, int64_t* stats_out,
                                            void* workspace, size_t workspace_bytes, void* stream) {
  return mask_loss_forward_impl<false>(logits, gt_classes, gt_masks, B, C, HW, dtype, loss_out, stats_out, workspace,
                                       workspace_bytes, stream);
}

extern "C" int d2amd_mask_rcnn_loss_forward_masked(const void* logits, const int64_t* gt_classes,
                                                   const uint8_t* gt_masks, int B, int C, int HW, int dtype,
                                                   float* loss_out, int64_t* stats_out, void* workspace,
                                                   size_t workspace_bytes, void* stream) {
  D2_CHECK_ARG(gt_classes != nullptr, "mask_rcnn_loss_forward_masked: the classes carry the mask");
  return mask_loss_forward_impl<true>(logits, gt_classes, gt_masks, B, C, HW, dtype, loss_out, stats_out, workspace,
                                      workspace_bytes, stream);
}

extern "C" int d2amd_mask_rcnn_loss_backward(const void* logits, const int64_t* gt_classes, const uint8_t* gt_masks,
                                             const float* grad_loss, int B, int C, int HW, int dtype,
                                             void* grad_logits, void* stream) {
  D2_CHECK_ARG(B >= 0 && C > 0 && HW > 0, "mask_rcnn_loss_backward: bad shape");
  D2_CHECK_ARG(gt_classes != nullptr || C == 1, "mask_rcnn_loss_backward: class-specific logits need gt_classes");
  if (B == 0) return D2AMD_OK;
  D2_CHECK_ARG(logits && gt_masks && grad_loss && grad_logits, "mask_rcnn_loss_backward: null pointer");
  const float inv_n = (float)(1.0 / ((double)B * (double)HW));
  hipStream_t s = (hipStream_t)stream;
  return D2_DISPATCH_DTYPE(dtype, [&]() -> int {
    constexpr int VEC = 16 / (int)sizeof(scalar_t);
    const bool vec = HW % VEC == 0 && ((uintptr_t)grad_logits & 15) == 0;
    const long n = (long)B * C * HW, nvec = vec ? n / VEC : n;
    const long nblk = (nvec + 255) / 256;
    D2_CHECK_ARG(nblk < (1l << 31), "mask_rcnn_loss_backward: tensor too large");
    if (vec) {
      hipLaunchKernelGGL((mask_loss_backward_kernel<scalar_t, VEC>), dim3((unsigned)nblk), dim3(256), 0, s,
                         (const scalar_t*)logits, gt_classes, gt_masks, grad_loss, C, HW, nvec, inv_n,
                         (scalar_t*)grad_logits);
    } else {
      hipLaunchKernelGGL((mask_loss_backward_kernel<scalar_t, 1>), dim3((unsigned)nblk), dim3(256), 0, s,
                         (const scalar_t*)logits, gt_classes, gt_masks, grad_loss, C, HW, nvec, inv_n,
                         (scalar_t*)grad_logits);
    }
    D2_LAUNCH_OK();
    return D2AMD_OK;
  });
}

extern "C" int d2amd_mask_rcnn_loss_backward_masked(const void* logits, const int64_t* gt_classes,
                                                    const uint8_t* gt_masks, const float* grad_loss,
                                                    const int64_t* rows, int B, int C, int HW, int dtype,
                                                    void* grad_logits, void* stream) {
  D2_CHECK_ARG(B >= 0 && C > 0 && HW > 0, "mask_rcnn_loss_backward_masked: bad shape");
  if (B == 0) return D2AMD_OK;
  D2_CHECK_ARG(logits && gt_classes && gt_masks && grad_loss && rows && grad_logits,
               "mask_rcnn_loss_backward_masked: null pointer");
  hipStream_t s = (hipStream_t)stream;
  return D2_DISPATCH_DTYPE(dtype, [&]() -> int {
    constexpr int VEC = 16 / (int)sizeof(scalar_t);
    const bool vec = HW % VEC == 0 && ((uintptr_t)grad_logits & 15) == 0;
    const long n = (long)B * C * HW, nvec = vec ? n / VEC : n;
    const long nblk = (nvec + 255) / 256;
    D2_CHECK_ARG(nblk < (1l << 31), "mask_rcnn_loss_backward_masked: tensor too large");
    if (vec) {
      hipLaunchKernelGGL((mask_loss_backward_masked_kernel<scalar_t, VEC>), dim3((unsigned)nblk), dim3(256), 0, s,
                         (const scalar_t*)logits, gt_classes, gt_masks, grad_loss, rows, C, HW, nvec,
                         (scalar_t*)grad_logits);
    } else {
      hipLaunchKernelGGL((mask_loss_backward_masked_kernel<scalar_t, 1>), dim3((unsigned)nblk), dim3(256), 0, s,
                         (const scalar_t*)logits, gt_classes, gt_masks, grad_loss, rows, C, HW, nvec,
                         (scalar_t*)grad_logits);
    }
    D2_LAUNCH_OK();
    return D2AMD_OK;
  });
}
