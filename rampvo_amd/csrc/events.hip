// Device-side event stacking (the step before the encoder): reference utils/transformers.py:128-161,
// EventToStack_Numpy -- upstream a host-side np.add.at over 100k-500k events per step.
//   event i -> bin int32(float32(bins * i) / N)   (by index, not by time stamp)
//   grid[bin][y][x] += polarity (+-1), then cast to int8 (wraps modulo 256, like the numpy cast)
// Integer pixel coordinates only (the reference's uint16 path): the accumulation is an integer
// atomic add, so the result does not depend on the order the events arrive in.
#include "ramp_device.h"

__global__ void __launch_bounds__(256)
    event_scatter_kernel(const int32_t *__restrict__ x, const int32_t *__restrict__ y, const int8_t *__restrict__ p,
                         int32_t *__restrict__ grid, int N, int bins, int H, int W) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int xi = x[i], yi = y[i];
  if (xi < 0 || yi < 0 || xi >= W || yi >= H) return;
  const int b = (int)(((float)bins * (float)i) / (float)N);
  atomicAdd(&grid[((size_t)b * H + yi) * W + xi], (int)p[i]);
}

__global__ void __launch_bounds__(256)
    event_cast_kernel(const int32_t *__restrict__ grid, int8_t *__restrict__ out8, float *__restrict__ outf, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int8_t v = (int8_t)grid[i];              // wraps modulo 256
  if (out8) out8[i] = v;
  if (outf) outf[i] = (float)v;
}

extern "C" {

size_t ramp_event_stack_workspace_bytes(int bins, int H, int W) { return (size_t)bins * H * W * sizeof(int32_t); }

int ramp_event_stack(const int32_t *x, const int32_t *y, const int8_t *p, int N, int bins, int H, int W,
                     int8_t *out_i8, float *out_f32, void *ws, size_t ws_bytes, void *stream) {
  if (N < 0 || bins <= 0 || H <= 0 || W <= 0 || (!out_i8 && !out_f32) || !ws) return RAMP_EINVAL;
  if (ws_bytes < ramp_event_stack_workspace_bytes(bins, H, W)) return RAMP_EWORKSPACE;
  if (N > 0 && (!x || !y || !p)) return RAMP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long n = (long)bins * H * W;
  if (hipMemsetAsync(ws, 0, (size_t)n * sizeof(int32_t), st) != hipSuccess) return RAMP_ELAUNCH;
  if (N >= 2)     // fewer than 2 events: an empty grid (transformers.py:151-152)
    hipLaunchKernelGGL(event_scatter_kernel, dim3(ramp_cdiv(N, 256)), dim3(256), 0, st, x, y, p, (int32_t *)ws, N,
                       bins, H, W);
  hipLaunchKernelGGL(event_cast_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const int32_t *)ws,
                     out_i8, out_f32, n);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

}  // extern "C"
