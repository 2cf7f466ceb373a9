// Evaluation kernels: the confusion histogram behind mmseg IoUMetric.intersect_and_union (rein/dg_metrics.py:46-52 calls it per
// sample).  HBM-bound: 1 B prediction + 1 or 8 B label per pixel, read once with 16-byte loads; counts privatised per wave in LDS,
// one integer atomic per touched bin per block.
#include "common.h"

#define NC_MAX 64  // classes on the path: 19 (Cityscapes); rows = num_classes + 1 ("other" labels that are not ignore_index)

template <typename LT>
__global__ __launch_bounds__(256) void k_confusion_hist(const uint8_t* __restrict__ pred, const LT* __restrict__ label, long n, int nc,
                                                        int ignore, unsigned long long* __restrict__ hist) {
  // per-wave private copies: 4 waves x (nc+1) x nc 32-bit counters (19 classes: 4 x 380 x 4 B = 6 KiB)
  extern __shared__ unsigned int sh[];
  const int bins = (nc + 1) * nc;
  const int wave = threadIdx.x >> 6;
  unsigned int* mine = sh + wave * bins;
  for (int i = threadIdx.x; i < 4 * bins; i += 256) sh[i] = 0u;
  __syncthreads();
  // 16 pixels per thread-iteration: one 16-byte load of predictions (and the matching label loads)
  const long n16 = n >> 4;
  for (long v = blockIdx.x * 256L + threadIdx.x; v < n16; v += (long)gridDim.x * 256L) {
    const uint4 p4 = reinterpret_cast<const uint4*>(pred)[v];
    const uint32_t pw[4] = {p4.x, p4.y, p4.z, p4.w};
    int lab[16];
    if constexpr (sizeof(LT) == 1) {
      const uint4 l4 = reinterpret_cast<const uint4*>(label)[v];
      const uint32_t lw[4] = {l4.x, l4.y, l4.z, l4.w};
#pragma unroll
      for (int j = 0; j < 16; ++j) lab[j] = (lw[j >> 2] >> (8 * (j & 3))) & 0xff;
    } else {
      const longlong2* lp = reinterpret_cast<const longlong2*>(label + (v << 4));
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const longlong2 t = lp[j];
        lab[2 * j] = (int)t.x;
        lab[2 * j + 1] = (int)t.y;
      }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int p = (pw[j >> 2] >> (8 * (j & 3))) & 0xff;
      const int l = lab[j];
      if (l != ignore && p < nc) atomicAdd(&mine[((l >= 0 && l < nc) ? l : nc) * nc + p], 1u);
    }
  }
  // tail pixels (n % 16) by the first threads of block 0
  if (blockIdx.x == 0) {
    for (long i = (n16 << 4) + threadIdx.x; i < n; i += 256) {
      const int p = pred[i];
      const int l = (int)label[i];
      if (l != ignore && p < nc) atomicAdd(&mine[((l >= 0 && l < nc) ? l : nc) * nc + p], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < bins; i += 256) {
    const unsigned int c = sh[i] + sh[bins + i] + sh[2 * bins + i] + sh[3 * bins + i];
    if (c) atomicAdd(&hist[i], (unsigned long long)c);
  }
}

extern "C" int vfm_confusion_hist(const uint8_t* pred, const void* label, int label_dt, long n, int num_classes, int ignore_index,
                                  int64_t* hist, void* stream) {
  VFM_CHECK(hist && n >= 0, VFM_E_INVAL, "vfm_confusion_hist: null histogram / negative size");
  if (n == 0) return VFM_OK;  // an empty map (its pointers may be null) adds nothing
  VFM_CHECK(pred && label, VFM_E_INVAL, "vfm_confusion_hist: null pointer");
  VFM_CHECK(num_classes > 0 && num_classes <= NC_MAX, VFM_E_SHAPE, "vfm_confusion_hist: num_classes %d not in [1, %d]", num_classes, NC_MAX);
  VFM_CHECK(label_dt == VFM_I64 || label_dt == VFM_U8, VFM_E_UNSUPPORTED, "vfm_confusion_hist: labels must be int64 or uint8");
  VFM_CHECK(((uintptr_t)pred & 15) == 0 && ((uintptr_t)label & 15) == 0, VFM_E_ALIGN, "vfm_confusion_hist: 16-byte aligned maps required");
  const int bins = (num_classes + 1) * num_classes;
  const size_t lds = 4u * bins * sizeof(unsigned int);
  long blocks = (n / 16 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;  // 8 blocks per CU: enough loads in flight for the HBM stream, few flush atomics
  if (label_dt == VFM_U8)
    hipLaunchKernelGGL(k_confusion_hist<uint8_t>, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, pred,
                       (const uint8_t*)label, n, num_classes, ignore_index, (unsigned long long*)hist);
  else
    hipLaunchKernelGGL(k_confusion_hist<int64_t>, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, pred,
                       (const int64_t*)label, n, num_classes, ignore_index, (unsigned long long*)hist);
  VFM_LAUNCH_CHECK();
  return VFM_OK;
}
