// Feed-rate probe for the persistent GEMM (tools/, not part of the library): how many bytes per clock a CU can pull
// out of its XCD's L2 when a 192x128x64 bf16 stage (A 24 KB + B 16 KB) arrives
//   mode 0  all by LDS-DMA (global_load_lds_dwordx4), the current kernel's scheme
//   mode 1  B by LDS-DMA, A straight into registers in MFMA-fragment order (lane = row m, k-group kq; 16 rows x 64 B
//           per wave-instruction)
//   mode 2  as 1 with the contraction index permuted so that a lane's two k-steps are 32 contiguous bytes
//   mode 3  A into registers only          mode 4  B by LDS-DMA only
//   mode 5  A into registers with a DMA-shaped lane map (8 rows x 128 B per instruction; not usable as fragments:
//           the coalescing yardstick for mode 1 / 2)
// No MFMA, no LDS reads: the pure arrival rate with two stages in flight.  Build + run: tools/l2_feed_probe.sh
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const void* g, uint32_t lds) {
  lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds);  // wave-uniform by construction
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds) : "m0", "memory");
}
__device__ __forceinline__ void ld16(u32x4& d, const void* g) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(g) : "memory");
}

constexpr int STAGE_A = 192 * 128, STAGE_B = 128 * 128, SLOT = STAGE_A + STAGE_B;

// one stage's requests of this thread; a = this block's A panel (row stride lda bytes), b = its B panel (ldb)
template <int MODE>
__device__ __forceinline__ void issue(const char* a, const char* b, int64_t lda, int64_t ldb, int koff, uint32_t slot_lds,
                                      u32x4 (&r)[6], int t) {
  const int lane = t & 63, w = t >> 6;
  if (MODE == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {  // A: 8 rows x 128 B per wave-instruction
      const int row = 48 * w + 8 * i + (lane >> 3);
      dma16(a + row * lda + koff + 16 * (lane & 7), slot_lds + (48 * w + 8 * i) * 128);
    }
  }
  if (MODE == 1 || MODE == 2 || MODE == 3) {
    const int m = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int row = 48 * w + 16 * i + m;
        const int off = MODE == 2 ? 32 * kq + 16 * ks : 64 * ks + 16 * kq;
        ld16(r[2 * i + ks], a + row * lda + koff + off);
      }
  }
  if (MODE == 5) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int row = 48 * w + 8 * i + (lane >> 3);
      ld16(r[i], a + row * lda + koff + 16 * (lane & 7));
    }
  }
  if (MODE == 0 || MODE == 1 || MODE == 2 || MODE == 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // B: 8 rows x 128 B per wave-instruction
      const int row = 32 * w + 8 * i + (lane >> 3);
      dma16(b + row * ldb + koff + 16 * (lane & 7), slot_lds + STAGE_A + (32 * w + 8 * i) * 128);
    }
  }
}
template <int MODE> constexpr int per_stage() { return MODE == 0 ? 10 : MODE == 1 || MODE == 2 ? 10 : MODE == 4 ? 4 : 6; }

template <int MODE>
__global__ __launch_bounds__(256) void probe(const char* A, const char* B, int64_t lda, int64_t ldb, int ksteps, int tiles_m,
                                             int tiles_n, int reps, uint32_t* sink, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const int t = threadIdx.x;
  // XCD-aware order as in the GEMM: logical neighbours share an XCD; a block's tile = (tm, tn)
  const int nb = gridDim.x, q = nb >> 3, rr = nb & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int lid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
  u32x4 r0[6], r1[6], r2[6];
  uint32_t acc = 0;
  const unsigned long long c0 = __builtin_readcyclecounter();
  constexpr int P = per_stage<MODE>();
  for (int rep = 0; rep < reps; ++rep) {
    const int tile = (lid + rep * nb) % (tiles_m * tiles_n);
    const int tn = tile % tiles_n, tm = tile / tiles_n;
    const char* a = A + (int64_t)tm * 192 * lda;
    const char* b = B + (int64_t)tn * 128 * ldb;
    // ksteps is a multiple of 3: slots / register sets rotate with the unrolled body
    issue<MODE>(a, b, lda, ldb, 0, lds0, r0, t);
    issue<MODE>(a, b, lda, ldb, 128, lds0 + SLOT, r1, t);
    for (int ks = 0; ks < ksteps; ks += 3) {
#define STEP(RC, RN, K, SL)                                                                                       \
  if ((K) + 2 < ksteps) {                                                                                         \
    issue<MODE>(a, b, lda, ldb, ((K) + 2) * 128, lds0 + (SL) * SLOT, RN, t);                                      \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P) : "memory");                                                  \
  } else if ((K) + 1 < ksteps) {                                                                                  \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");                                                      \
  } else {                                                                                                        \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                              \
  }                                                                                                               \
  if (MODE != 0 && MODE != 4) {                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                                               \
      asm volatile("" : "+v"(RC[i]));                                                                             \
      acc ^= RC[i].x ^ RC[i].y ^ RC[i].z ^ RC[i].w;                                                               \
    }                                                                                                             \
  }                                                                                                               \
  __builtin_amdgcn_s_barrier();
      STEP(r0, r2, ks, 2)
      STEP(r1, r0, ks + 1, 0)
      STEP(r2, r1, ks + 2, 1)
#undef STEP
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  if (acc == 0x12345678u) sink[0] = acc;
  if (t == 0) cyc[blockIdx.x] = c1 - c0;
}

template <int MODE>
static void run(const char* name, const char* A, const char* B, int64_t lda, int64_t ldb, int ksteps, int tm, int tn, int grid,
                uint32_t* sink, unsigned long long* cyc, double bytes_per_stage) {
  const int reps = 40;
  CK(hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * SLOT));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int it = 0; it < 4; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 3 * SLOT, 0, A, B, lda, ldb, ksteps, tm, tn, reps, sink, cyc);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (it && ms < best) best = ms;
  }
  std::vector<unsigned long long> h(grid);
  CK(hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double mean = 0;
  for (auto v : h) mean += (double)v;
  mean /= grid;
  const double stages = (double)reps * ksteps;
  const double bytes = stages * bytes_per_stage;
  const double per_cu = bytes * (grid / 256.0) / (best * 1e-3) / 1e9;  // blocks per CU x bytes per block / time
  printf("%-48s grid %4d  %8.1f us  %7.1f GB/s per CU  %6.2f TB/s chip  %7.1f ns per stage  (memtime ticks/stage %.2f)\n", name,
         grid, best * 1e3, per_cu, per_cu * 256 / 1e3, best * 1e6 / stages, mean / stages);
}

int main(int argc, char** argv) {
  const int M = 12000, N = argc > 1 ? atoi(argv[1]) : 2048, K = argc > 2 ? atoi(argv[2]) : 512;
  const int tm = (M + 191) / 192, tn = N / 128, ksteps = K / 64 / 3 * 3 ? K / 64 / 3 * 3 : 3;
  const int64_t lda = (int64_t)K * 2, ldb = (int64_t)K * 2;
  char *A, *B;
  uint32_t* sink;
  unsigned long long* cyc;
  CK(hipMalloc(&A, (size_t)tm * 192 * lda + 4096));
  CK(hipMalloc(&B, (size_t)N * ldb + 4096));
  CK(hipMemset(A, 1, (size_t)tm * 192 * lda));
  CK(hipMemset(B, 1, (size_t)N * ldb));
  CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&cyc, 1024 * sizeof(unsigned long long)));
  printf("M %d N %d K %d: %d x %d tiles, %d k-steps per tile used\n", M, N, K, tm, tn, ksteps);
  for (int grid : {256, 512}) {
    if (grid == 512) continue;  // 3 slots = 120 KB: one block per CU
    run<0>("0 all LDS-DMA (A 24 KB + B 16 KB)", A, B, lda, ldb, ksteps, tm, tn, grid, sink, cyc, SLOT);
    run<1>("1 B LDS-DMA + A fragments to VGPR", A, B, lda, ldb, ksteps, tm, tn, grid, sink, cyc, SLOT);
    run<2>("2 B LDS-DMA + A fragments, 32 B per lane", A, B, lda, ldb, ksteps, tm, tn, grid, sink, cyc, SLOT);
    run<3>("3 A fragments to VGPR only (24 KB)", A, B, lda, ldb, ksteps, tm, tn, grid, sink, cyc, STAGE_A);
    run<4>("4 B LDS-DMA only (16 KB)", A, B, lda, ldb, ksteps, tm, tn, grid, sink, cyc, STAGE_B);
    run<5>("5 A to VGPR, 8 rows x 128 B per instr (24 KB)", A, B, lda, ldb, ksteps, tm, tn, grid, sink, cyc, STAGE_A);
  }
  return 0;
}
