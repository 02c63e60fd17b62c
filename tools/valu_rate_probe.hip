// Issue cost of the vector instructions the attention kernels' element loops are made of (tools/, not part of the library):
// clocks per wave-instruction with one and with two waves per SIMD, independent operands (8 chains per lane).
// build + run: hipcc -O2 --offload-arch=gfx950 tools/valu_rate_probe.hip -o tools/valu_rate_probe && tools/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define CHAIN8(OP)                                                                                  \
  asm volatile(OP(%0) "\n\t" OP(%1) "\n\t" OP(%2) "\n\t" OP(%3) "\n\t" OP(%4) "\n\t" OP(%5) "\n\t" OP(%6) "\n\t" OP(%7) \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])          \
               : "v"(c))

#define OP_MUL_LO(r) "v_mul_lo_u32 " #r ", " #r ", %8"
#define OP_MUL24(r) "v_mul_u32_u24 " #r ", " #r ", %8"
#define OP_MAD24(r) "v_mad_u32_u24 " #r ", " #r ", %8, %8"
#define OP_XOR(r) "v_xor_b32 " #r ", " #r ", %8"
#define OP_ADD(r) "v_add_u32 " #r ", " #r ", %8"
#define OP_FMA(r) "v_fma_f32 " #r ", " #r ", %8, %8"
#define OP_EXP(r) "v_exp_f32 " #r ", " #r
#define OP_LSHR(r) "v_lshrrev_b32 " #r ", 15, " #r
#define OP_MULF(r) "v_mul_f32 " #r ", " #r ", %8"
#define OP_PKMUL(r) "v_pk_mul_f32 " #r ", " #r ", " #r   /* placeholder: needs 64-bit regs, not timed */
#define OP_CVT(r) "v_cvt_pk_bf16_f32 " #r ", " #r ", %8"

template <int WHICH>
__global__ void probe(uint32_t* out, unsigned long long* ticks, uint32_t c, int iters) {
  uint32_t a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 8 + i + 1;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (WHICH == 0) { CHAIN8(OP_MUL_LO); CHAIN8(OP_MUL_LO); CHAIN8(OP_MUL_LO); CHAIN8(OP_MUL_LO); }
    if (WHICH == 1) { CHAIN8(OP_MUL24); CHAIN8(OP_MUL24); CHAIN8(OP_MUL24); CHAIN8(OP_MUL24); }
    if (WHICH == 2) { CHAIN8(OP_MAD24); CHAIN8(OP_MAD24); CHAIN8(OP_MAD24); CHAIN8(OP_MAD24); }
    if (WHICH == 3) { CHAIN8(OP_XOR); CHAIN8(OP_XOR); CHAIN8(OP_XOR); CHAIN8(OP_XOR); }
    if (WHICH == 4) { CHAIN8(OP_FMA); CHAIN8(OP_FMA); CHAIN8(OP_FMA); CHAIN8(OP_FMA); }
    if (WHICH == 5) { CHAIN8(OP_EXP); CHAIN8(OP_EXP); CHAIN8(OP_EXP); CHAIN8(OP_EXP); }
    if (WHICH == 6) { CHAIN8(OP_LSHR); CHAIN8(OP_LSHR); CHAIN8(OP_LSHR); CHAIN8(OP_LSHR); }
    if (WHICH == 7) { CHAIN8(OP_CVT); CHAIN8(OP_CVT); CHAIN8(OP_CVT); CHAIN8(OP_CVT); }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  uint32_t s = 0;
  for (int i = 0; i < 8; ++i) s ^= a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}

template <int WHICH>
void run(const char* name, int threads) {
  uint32_t* out;
  unsigned long long* ticks;
  CK(hipMalloc(&out, 256 * 1024 * 4));
  CK(hipMalloc(&ticks, 8));
  const int iters = 2000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  probe<WHICH><<<256, threads>>>(out, ticks, 0x846ca68bu, 10);
  CK(hipEventRecord(e0));
  probe<WHICH><<<256, threads>>>(out, ticks, 0x846ca68bu, iters);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long t;
  CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
  const double n = 32.0 * iters;  // instructions per wave
  printf("%-18s %4d threads/CU (%d wave(s) per SIMD): %6.2f ticks per wave-instruction, %7.1f us  -> %5.2f ns per instruction and SIMD\n", name, threads,
         threads / 256, (double)t / n, ms * 1e3, ms * 1e6 / (n * (threads / 256)));
  CK(hipFree(out));
  CK(hipFree(ticks));
}

int main() {
  for (int threads : {256, 512}) {
    run<0>("v_mul_lo_u32", threads);
    run<1>("v_mul_u32_u24", threads);
    run<2>("v_mad_u32_u24", threads);
    run<3>("v_xor_b32", threads);
    run<4>("v_fma_f32", threads);
    run<5>("v_exp_f32", threads);
    run<6>("v_lshrrev_b32", threads);
    run<7>("v_cvt_pk_bf16_f32", threads);
  }
  return 0;
}
