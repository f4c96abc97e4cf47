// Register-allocation probe: can hipcc keep 256 VGPRs of resident MFMA operands + 128 accumulators in a 1-wave-per-SIMD kernel without scratch?
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

template <int NT>
__global__ __launch_bounds__(256, 1) void probe(const uint4* __restrict__ w, const uint4* __restrict__ act, float* __restrict__ out, int T) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint4 wr[4][16];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) wr[g][kb] = w[((blockIdx.x * 4 + wave) * 64 + g * 16 + kb) * 64 + lane];
  f32x4 acc[NT][4];
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[ni][g] = f32x4{0, 0, 0, 0};
    const uint4* a = act + (size_t)t * 128 * 136;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
      uint4 b[NT];
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) b[ni] = a[(ni * 16 + (lane & 15)) * 136 + (wave * 16 + kb) * 4 + (lane >> 4)];
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          acc[ni][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wr[g][kb]), __builtin_bit_cast(bf16x8_t, b[ni]), acc[ni][g], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g) s += acc[ni][g][0] + acc[ni][g][1] + acc[ni][g][2] + acc[ni][g][3];
    out[(size_t)(t * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = s;
  }
}
template __global__ void probe<8>(const uint4*, const uint4*, float*, int);
template __global__ void probe<4>(const uint4*, const uint4*, float*, int);
