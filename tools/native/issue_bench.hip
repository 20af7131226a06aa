// Instruction-issue micro-benchmark for gfx950: what a SIMD sustains per wave64 instruction, by instruction class and by
// resident wavefronts per SIMD.  The scorer (csrc/score_qs.hip) and the assembly kernels are bound by instruction issue;
// rounds 1-5 priced every instruction at 4 cycles ("a wavefront issues one instruction per 4 cycles whatever its type").
// This program measures it instead:
//   hipcc --offload-arch=gfx950 -O2 -o issue_bench tools/native/issue_bench.hip && ./issue_bench
// Output: one line per (body, waves per SIMD): cycles per instruction per SIMD at the nominal 2.4 GHz.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

// every body: `iters` trips of an unrolled block of K instructions; results folded into out[] so nothing is dead
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)

__global__ void __launch_bounds__(256) k_valu_andor(int iters, uint32_t *out) {
  uint32_t a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, m = 0x10001u * (threadIdx.x & 15);
  for (int i = 0; i < iters; ++i) {
    REP8(asm volatile("v_and_or_b32 %0, %8, %9, %0\n\tv_and_or_b32 %1, %8, %9, %1\n\tv_and_or_b32 %2, %8, %9, %2\n\tv_and_or_b32 %3, %8, %9, %3\n\t"
                      "v_and_or_b32 %4, %8, %9, %4\n\tv_and_or_b32 %5, %8, %9, %5\n\tv_and_or_b32 %6, %8, %9, %6\n\tv_and_or_b32 %7, %8, %9, %7"
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "s"(i));)
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
constexpr int K_VALU_ANDOR = 64;

__global__ void __launch_bounds__(256) k_valu_pk16(int iters, uint32_t *out) {
  uint32_t a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
  for (int i = 0; i < iters; ++i) {
    REP8(asm volatile("v_pk_sub_i16 %0, %8, %0\n\tv_pk_sub_i16 %1, %8, %1\n\tv_pk_sub_i16 %2, %8, %2\n\tv_pk_sub_i16 %3, %8, %3\n\t"
                      "v_pk_ashrrev_i16 %4, 15, %4\n\tv_pk_ashrrev_i16 %5, 15, %5\n\tv_pk_ashrrev_i16 %6, 15, %6\n\tv_pk_ashrrev_i16 %7, 15, %7"
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(i));)
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
constexpr int K_VALU_PK16 = 64;

__global__ void __launch_bounds__(256) k_valu_addf64(int iters, double *out) {
  double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, m = 1e-9 * threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    REP8(asm volatile("v_add_f64 %0, %0, %8\n\tv_add_f64 %1, %1, %8\n\tv_add_f64 %2, %2, %8\n\tv_add_f64 %3, %3, %8\n\t"
                      "v_add_f64 %4, %4, %8\n\tv_add_f64 %5, %5, %8\n\tv_add_f64 %6, %6, %8\n\tv_add_f64 %7, %7, %8"
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
constexpr int K_VALU_ADDF64 = 64;

// one dependent chain of f64 adds (the leaf sum of one row)
__global__ void __launch_bounds__(256) k_valu_addf64_chain(int iters, double *out) {
  double a0 = threadIdx.x, m = 1e-9 * threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    REP16(asm volatile("v_add_f64 %0, %0, %1\n\tv_add_f64 %0, %0, %1\n\tv_add_f64 %0, %0, %1\n\tv_add_f64 %0, %0, %1" : "+v"(a0) : "v"(m));)
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0;
}
constexpr int K_VALU_ADDF64_CHAIN = 64;

__global__ void __launch_bounds__(256) k_salu(int iters, uint32_t *out) {
  uint32_t s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
  for (int i = 0; i < iters; ++i) {
    REP16(asm volatile("s_lshr_b32 %0, %0, 1\n\ts_pack_ll_b32_b16 %1, %1, %0\n\ts_lshr_b32 %2, %2, 1\n\ts_pack_ll_b32_b16 %3, %3, %2"
                       : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");)
  }
  out[blockIdx.x * 256 + threadIdx.x] = s0 ^ s1 ^ s2 ^ s3;
}
constexpr int K_SALU = 64;

// VALU and SALU side by side in ONE wavefront's stream (1 : 1)
__global__ void __launch_bounds__(256) k_valu_salu(int iters, uint32_t *out) {
  uint32_t a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, m = 0x10001u * (threadIdx.x & 15);
  uint32_t s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
  for (int i = 0; i < iters; ++i) {
    REP8(asm volatile("v_and_or_b32 %0, %8, %9, %0\n\ts_lshr_b32 %4, %4, 1\n\tv_and_or_b32 %1, %8, %9, %1\n\ts_pack_ll_b32_b16 %5, %5, %4\n\t"
                      "v_and_or_b32 %2, %8, %9, %2\n\ts_lshr_b32 %6, %6, 1\n\tv_and_or_b32 %3, %8, %9, %3\n\ts_pack_ll_b32_b16 %7, %7, %6"
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(m), "s"(i) : "scc");)
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ s0 ^ s1 ^ s2 ^ s3;
}
constexpr int K_VALU_SALU = 64;  // 32 VALU + 32 SALU

// the scorer's tree step as it is today: per node an M0 write, a mask replication, one ds_read_addtid, three VALU;
// 15 nodes + 13 VALU of leaf bookkeeping stand-ins = 58 VALU + 30 SALU + 15 LDS per "tree"
__global__ void __launch_bounds__(256) k_tree_now(int iters, uint32_t *out) {
  __shared__ uint32_t slab[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 256) slab[i] = i * 2654435761u;
  __syncthreads();
  uint32_t acc_a = 0, acc_b = 0, kk = 0x00400040u;
  uint32_t mv = (uint32_t)(blockIdx.x & 7) << 24 | 0x5a5au;
  for (int i = 0; i < iters; ++i) {
    uint32_t c[15], mm[15];
#pragma unroll
    for (int s = 0; s < 15; ++s) {
      asm volatile("s_lshr_b32 m0, %2, 16\n\ts_pack_ll_b32_b16 %1, %2, %2\n\tds_read_addtid_b32 %0" : "=v"(c[s]), "=s"(mm[s]) : "s"(mv + (s << 24)) : "memory", "scc");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < 15; ++s) asm volatile("v_pk_sub_i16 %0, %1, %0" : "+v"(c[s]) : "s"(kk));
#pragma unroll
    for (int s = 0; s < 15; ++s) asm volatile("v_pk_ashrrev_i16 %0, 15, %0" : "+v"(c[s]));
#pragma unroll
    for (int s = 0; s < 15; s += 2) {
      asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(acc_a) : "v"(c[s]), "s"(mm[s]));
      if (s + 1 < 15) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(acc_b) : "v"(c[s + 1]), "s"(mm[s + 1]));
    }
    REP8(asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc_a) : "v"(acc_b));)
    REP4(asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc_b) : "v"(acc_a));)
    asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc_b) : "v"(acc_a));
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc_a ^ acc_b;
}
constexpr int K_TREE_NOW = 58 + 30 + 15;

// the same with NO scalar work per node: the replicated mask and the LDS byte offset come ready-made (scalar
// registers loaded once), the read is a plain ds_read_b32 with a VGPR address formed by one v_add per node
__global__ void __launch_bounds__(256) k_tree_valu_addr(int iters, uint32_t *out) {
  __shared__ uint32_t slab[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 256) slab[i] = i * 2654435761u;
  __syncthreads();
  uint32_t acc_a = 0, acc_b = 0, kk = 0x00400040u, mm = 0x5a5a5a5au;
  const uint32_t lane_off = (threadIdx.x & 63) * 4, voff = (blockIdx.x & 7) << 8;
  for (int i = 0; i < iters; ++i) {
    uint32_t c[15];
#pragma unroll
    for (int s = 0; s < 15; ++s) {
      uint32_t addr;
      asm volatile("v_add_u32 %0, %1, %2" : "=v"(addr) : "s"(voff + (s << 8)), "v"(lane_off));
      asm volatile("ds_read_b32 %0, %1" : "=v"(c[s]) : "v"(addr) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < 15; ++s) asm volatile("v_pk_sub_i16 %0, %1, %0" : "+v"(c[s]) : "s"(kk));
#pragma unroll
    for (int s = 0; s < 15; ++s) asm volatile("v_pk_ashrrev_i16 %0, 15, %0" : "+v"(c[s]));
#pragma unroll
    for (int s = 0; s < 15; s += 2) {
      asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(acc_a) : "v"(c[s]), "s"(mm));
      if (s + 1 < 15) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(acc_b) : "v"(c[s + 1]), "s"(mm));
    }
    REP8(asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc_a) : "v"(acc_b));)
    REP4(asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc_b) : "v"(acc_a));)
    asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc_b) : "v"(acc_a));
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc_a ^ acc_b;
}
constexpr int K_TREE_VALU_ADDR = 58 + 15 + 15;

// M0 written by s_mov from a ready-made offset register, mask ready-made: one SALU per node instead of two
__global__ void __launch_bounds__(256) k_tree_m0_only(int iters, uint32_t *out) {
  __shared__ uint32_t slab[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 256) slab[i] = i * 2654435761u;
  __syncthreads();
  uint32_t acc_a = 0, acc_b = 0, kk = 0x00400040u, mm = 0x5a5a5a5au;
  const uint32_t voff = (blockIdx.x & 7) << 8;
  for (int i = 0; i < iters; ++i) {
    uint32_t c[15];
#pragma unroll
    for (int s = 0; s < 15; ++s)
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_read_addtid_b32 %0" : "=v"(c[s]) : "s"(voff + (s << 8)) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < 15; ++s) asm volatile("v_pk_sub_i16 %0, %1, %0" : "+v"(c[s]) : "s"(kk));
#pragma unroll
    for (int s = 0; s < 15; ++s) asm volatile("v_pk_ashrrev_i16 %0, 15, %0" : "+v"(c[s]));
#pragma unroll
    for (int s = 0; s < 15; s += 2) {
      asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(acc_a) : "v"(c[s]), "s"(mm));
      if (s + 1 < 15) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(acc_b) : "v"(c[s + 1]), "s"(mm));
    }
    REP8(asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc_a) : "v"(acc_b));)
    REP4(asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc_b) : "v"(acc_a));)
    asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc_b) : "v"(acc_a));
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc_a ^ acc_b;
}
constexpr int K_TREE_M0_ONLY = 58 + 30 + 15;  // (s_nop counted as an issue slot)

// dependent LDS search chain: 8 dependent ds_read_b32 + compare/select (the binning search of the assembly kernel)
__global__ void __launch_bounds__(256) k_lds_chain(int iters, uint32_t *out) {
  __shared__ uint32_t slab[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 256) slab[i] = (i * 2654435761u) & 0x3ffcu;
  __syncthreads();
  uint32_t a = threadIdx.x * 4;
  for (int i = 0; i < iters; ++i) {
    REP8(asm volatile("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a) : : "memory");)
  }
  out[blockIdx.x * 256 + threadIdx.x] = a;
}
constexpr int K_LDS_CHAIN = 8;

template <typename T, typename K>
static void run(const char *name, K kern, int k_per_iter, int n_cus, T *d_out) {
  const int iters = 4096;
  fprintf(stderr, "%s\n", name);
  for (int w : {1, 2, 4, 8}) {  // 256-thread workgroups per CU = wavefronts per SIMD
    const int grid = n_cus * w;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, iters / 8, d_out);  // warm
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, iters, d_out);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    // every SIMD runs w wavefronts, each issuing iters * k instructions
    const double instr_per_simd = (double)iters * k_per_iter * w;
    const double cyc = best * 1e-3 * 2.4e9 / instr_per_simd;
    printf("%-22s waves/SIMD %d  %8.3f ms  %6.2f cycles/instruction/SIMD (2.4 GHz)   %7.1f cycles per trip per wavefront\n", name, w, best, cyc,
           best * 1e-3 * 2.4e9 / iters);
  }
}

int main() {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  fprintf(stderr, "start\n");
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int n_cus = p.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz\n", p.gcnArchName, n_cus, p.clockRate);
  void *d;
  CK(hipMalloc(&d, (size_t)n_cus * 8 * 256 * 8));
  run("valu v_and_or_b32", k_valu_andor, K_VALU_ANDOR, n_cus, (uint32_t *)d);
  run("valu v_pk_*_i16", k_valu_pk16, K_VALU_PK16, n_cus, (uint32_t *)d);
  run("valu v_add_f64 indep", k_valu_addf64, K_VALU_ADDF64, n_cus, (double *)d);
  run("valu v_add_f64 chain", k_valu_addf64_chain, K_VALU_ADDF64_CHAIN, n_cus, (double *)d);
  run("salu", k_salu, K_SALU, n_cus, (uint32_t *)d);
  run("valu+salu 1:1", k_valu_salu, K_VALU_SALU, n_cus, (uint32_t *)d);
  run("tree step (today)", k_tree_now, K_TREE_NOW, n_cus, (uint32_t *)d);
  run("tree step (m0 only)", k_tree_m0_only, K_TREE_M0_ONLY, n_cus, (uint32_t *)d);
  run("tree step (valu addr)", k_tree_valu_addr, K_TREE_VALU_ADDR, n_cus, (uint32_t *)d);
  run("lds dependent chain", k_lds_chain, K_LDS_CHAIN, n_cus, (uint32_t *)d);
  CK(hipFree(d));
  return 0;
}
