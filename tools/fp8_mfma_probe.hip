// Issue rate of the three matrix instructions the e4m3 path can use on gfx950 (tools/, not part of the library), and a check of
// the scaled instruction's operand layout: v_mfma_f32_16x16x32_bf16, v_mfma_f32_16x16x32_fp8_fp8 (what csrc/gemm.hip's e4m3 kernels
// issue today) and v_mfma_scale_f32_16x16x128_f8f6f4 with unit scales (E8M0 127 in every scale byte).
//   hipcc -O3 --offload-arch=gfx950 tools/fp8_mfma_probe.hip -o tools/fp8_mfma_probe && tools/fp8_mfma_probe
// prints cycles per instruction and SIMD, FLOP per cycle and SIMD, and chip-wide TFLOP/s at the clock the loop ran at.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef int i8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t rnd(uint32_t x) {
  x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
  return x;
}

// KIND 0: bf16 K=32, 1: fp8 K=32, 2: scaled f8f6f4 K=128
template <int KIND>
__global__ __launch_bounds__(512) void rate(int iters, float* out, unsigned long long* ticks) {
  const int t = threadIdx.x;
  constexpr int NA = 12;  // independent accumulators: no read-after-write stall
  f32x4_t acc[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) acc[i] = f32x4_t{0, 0, 0, 0};
  i8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    // bf16: random sign / mantissa, exponent near 1; e4m3 bytes: random sign / mantissa, exponents 6..9 (values ~0.5..4)
    const uint32_t x = rnd(t * 8 + i + 1), y = rnd(t * 8 + i + 77777);
    a[i] = KIND == 0 ? (int)((x & 0x807f807fu) | 0x3f003f00u) : (int)((x & 0x87878787u) | 0x38383838u);
    b[i] = KIND == 0 ? (int)((y & 0x807f807fu) | 0x3f003f00u) : (int)((y & 0x87878787u) | 0x38383838u);
  }
  typedef int i4 __attribute__((ext_vector_type(4)));
  typedef long l1;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if constexpr (KIND == 0) {
        const i4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a4), __builtin_bit_cast(bf16x8_t, b4), acc[i], 0, 0, 0);
      } else if constexpr (KIND == 1) {
        const long al = ((long)(uint32_t)a[1] << 32) | (uint32_t)a[0], bl = ((long)(uint32_t)b[1] << 32) | (uint32_t)b[0];
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(al, bl, acc[i], 0, 0, 0);
      } else {
        acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int i = 0; i < NA; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.f) out[0] = s;
  if (t == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

// layout check: D = A B^T over K = 128 with one scaled instruction; lane l supplies row (l & 15) of its operand and the 32 bytes
// k = 32 (l >> 4) .. + 31 (the assumption csrc would build on); compared with the same sum in double on the host
__global__ void layout(const uint8_t* A, const uint8_t* B, float* D) {
  const int l = threadIdx.x;
  i8 a, b;
  const uint32_t* ap = (const uint32_t*)(A + (l & 15) * 128 + 32 * (l >> 4));
  const uint32_t* bp = (const uint32_t*)(B + (l & 15) * 128 + 32 * (l >> 4));
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (int)ap[i], b[i] = (int)bp[i];
  f32x4_t c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  // accumulator layout of the 16x16 shapes: lane (g = l >> 4, r = l & 15), register e -> D[4 g + e][r] with A as the row operand
#pragma unroll
  for (int e = 0; e < 4; ++e) D[(4 * (l >> 4) + e) * 16 + (l & 15)] = c[e];
}

static double e4m3(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  double x = e == 0 ? std::ldexp(m / 8.0, -6) : std::ldexp(1.0 + m / 8.0, e - 7);
  return s ? -x : x;
}

int main() {
  float* out;
  unsigned long long* ticks;
  CK(hipMalloc(&out, 64));
  CK(hipMalloc(&ticks, 64));
  int dev = 0, cus = 0, khz = 0;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, dev));
  const char* names[3] = {"v_mfma_f32_16x16x32_bf16", "v_mfma_f32_16x16x32_fp8_fp8", "v_mfma_scale_f32_16x16x128_f8f6f4"};
  const double flop[3] = {2.0 * 16 * 16 * 32, 2.0 * 16 * 16 * 32, 2.0 * 16 * 16 * 128};
  for (int wps = 1; wps <= 2; ++wps) {
    for (int kind = 0; kind < 3; ++kind) {
      const int iters = 20000, threads = 256 * wps;
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(cus), dim3(threads), 0, 0, iters, out, ticks);
        if (kind == 1) hipLaunchKernelGGL(rate<1>, dim3(cus), dim3(threads), 0, 0, iters, out, ticks);
        if (kind == 2) hipLaunchKernelGGL(rate<2>, dim3(cus), dim3(threads), 0, 0, iters, out, ticks);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
      }
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long tk = 0;
      CK(hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost));
      const double n_inst = (double)iters * 12 * wps;  // per SIMD
      const double total = flop[kind] * n_inst * 4 * cus;
      printf("%-36s %d wave(s)/SIMD: %6.1f ns per instruction and SIMD, %7.1f TFLOP/s chip-wide (%d CUs, %.3f ms)\n", names[kind], wps,
             ms * 1e6 / n_inst, total / (ms * 1e-3) / 1e12, cus, ms);
    }
  }
  // layout
  std::vector<uint8_t> hA(16 * 128), hB(16 * 128);
  for (int i = 0; i < 16 * 128; ++i) {
    uint32_t x = i * 2654435761u + 12345u;
    x ^= x >> 13;
    hA[i] = (uint8_t)(((x >> 3) & 0x87) | 0x30 | ((x >> 9) & 0x08));
    x = x * 1664525u + 1013904223u;
    hB[i] = (uint8_t)(((x >> 5) & 0x87) | 0x30 | ((x >> 11) & 0x08));
  }
  uint8_t *dA, *dB;
  float* dD;
  CK(hipMalloc(&dA, 2048));
  CK(hipMalloc(&dB, 2048));
  CK(hipMalloc(&dD, 1024));
  CK(hipMemcpy(dA, hA.data(), 2048, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, hB.data(), 2048, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  std::vector<float> hD(256);
  CK(hipMemcpy(hD.data(), dD, 1024, hipMemcpyDeviceToHost));
  double worst = 0, worst_t = 0;
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n) {
      double s = 0;
      for (int k = 0; k < 128; ++k) s += e4m3(hA[m * 128 + k]) * e4m3(hB[n * 128 + k]);
      // which operand indexes rows is what the check finds out: D[m][n] or D[n][m]
      worst = fmax(worst, fabs(hD[m * 16 + n] - s) / (fabs(s) + 1e-3));
      worst_t = fmax(worst_t, fabs(hD[n * 16 + m] - s) / (fabs(s) + 1e-3));
    }
  printf("scaled 16x16x128, unit scales, lane (g, r) = row r, bytes k = 32 g .. 32 g + 31: max rel. error vs double %.2e (as D[m][n]) / %.2e (as D[n][m])\n",
         worst, worst_t);
  return 0;
}
