// Micro-benchmark: issue cost of the vector-ALU instructions the fused warp is built from, on gfx950.
// Every SIMD runs `waves` waves that each issue `iters` x 16 independent instructions of one kind; the figure printed is
// SIMD cycles per wave-instruction (clock taken from s_memtime inside the kernel, so DVFS does not enter).
//   hipcc -O3 --offload-arch=gfx950 tools/valu_microbench.hip -o tools/valu_microbench && tools/valu_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define REP16(X) X X X X X X X X X X X X X X X X

// BODY uses registers %0..%7 as accumulators / sources (two 64-bit pairs p0,p1 for packed forms)
#define KERNEL(NAME, BODY)                                                                                                    \
    __global__ __launch_bounds__(256) void NAME(int iters, uint32_t *sink, unsigned long long *clk)                           \
    {                                                                                                                         \
        uint32_t a = threadIdx.x * 2654435761u + 12345u, b = a ^ 0x5bd1e995u, c = a + 77u, d = b + 99u;                      \
        uint32_t e = a * 3u, f = b * 5u, g = c * 7u, h = d * 9u;                                                             \
        uint64_t p0 = ((uint64_t)a << 32) | b, p1 = ((uint64_t)c << 32) | d, p2 = ((uint64_t)e << 32) | f, p3 = ((uint64_t)g << 32) | h; \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                          \
        for (int i = 0; i < iters; ++i) { REP16(BODY) }                                                                        \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                          \
        if ((a ^ b ^ c ^ d ^ e ^ f ^ g ^ h ^ (uint32_t)p0 ^ (uint32_t)p1 ^ (uint32_t)p2 ^ (uint32_t)p3) == 0x1234567u) sink[0] = a;                          \
        if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                                      \
    }

#define A4(OP) asm volatile(OP " %0, %0, %4\n" OP " %1, %1, %5\n" OP " %2, %2, %6\n" OP " %3, %3, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));
#define A4_3(OP) asm volatile(OP " %0, %0, %4, %5\n" OP " %1, %1, %5, %6\n" OP " %2, %2, %6, %7\n" OP " %3, %3, %7, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));
#define A4_1(OP) asm volatile(OP " %0, %4\n" OP " %1, %5\n" OP " %2, %6\n" OP " %3, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));
#define P4_3(OP) asm volatile(OP " %0, %0, %2, %3\n" OP " %1, %1, %3, %2\n" OP " %0, %0, %3, %2\n" OP " %1, %1, %2, %3" : "+v"(p0), "+v"(p1) : "v"(p2), "v"(p3));
#define P4_2(OP) asm volatile(OP " %0, %0, %2\n" OP " %1, %1, %3\n" OP " %0, %0, %3\n" OP " %1, %1, %2" : "+v"(p0), "+v"(p1) : "v"(p2), "v"(p3));

KERNEL(k_add_u32, A4("v_add_u32"))
KERNEL(k_fma_f32, A4_3("v_fma_f32"))
KERNEL(k_mul_f32, A4("v_mul_f32"))
KERNEL(k_pk_fma_f32, P4_3("v_pk_fma_f32"))
KERNEL(k_pk_mul_f32, P4_2("v_pk_mul_f32"))
KERNEL(k_pk_add_f32, P4_2("v_pk_add_f32"))
KERNEL(k_rcp_f32, A4_1("v_rcp_f32"))
KERNEL(k_rndne_f32, A4_1("v_rndne_f32"))
KERNEL(k_cvt_i32_f32, A4_1("v_cvt_i32_f32"))
KERNEL(k_dot4_u32_u8, A4_3("v_dot4_u32_u8"))
KERNEL(k_dot2_u32_u16, A4_3("v_dot2_u32_u16"))
KERNEL(k_perm_b32, A4_3("v_perm_b32"))
KERNEL(k_alignbyte, A4_3("v_alignbyte_b32"))
KERNEL(k_alignbit, A4_3("v_alignbit_b32"))
KERNEL(k_mul_u32_u24, A4("v_mul_u32_u24"))
KERNEL(k_mad_u32_u24, A4_3("v_mad_u32_u24"))
KERNEL(k_mul_lo_u32, A4("v_mul_lo_u32"))
KERNEL(k_mad_u64_u32, asm volatile("v_mad_u64_u32 %0, s[20:21], %2, %3, %0\nv_mad_u64_u32 %1, s[20:21], %3, %2, %1\nv_mad_u64_u32 %0, s[20:21], %3, %2, %0\nv_mad_u64_u32 %1, s[20:21], %2, %3, %1" : "+v"(p0), "+v"(p1) : "v"(e), "v"(f) : "s20", "s21");)
KERNEL(k_mul_hi_u32, A4("v_mul_hi_u32"))
KERNEL(k_lshl_or, A4_3("v_lshl_or_b32"))
KERNEL(k_lshl_add, A4_3("v_lshl_add_u32"))
KERNEL(k_and_or, A4_3("v_and_or_b32"))
KERNEL(k_or3, A4_3("v_or3_b32"))
KERNEL(k_bfe_u32, A4_3("v_bfe_u32"))
KERNEL(k_xor, A4("v_xor_b32"))
KERNEL(k_pk_mad_u16, A4_3("v_pk_mad_u16"))
KERNEL(k_pk_mul_lo_u16, A4("v_pk_mul_lo_u16"))
KERNEL(k_pk_lshrrev_b16, A4("v_pk_lshrrev_b16"))
KERNEL(k_pk_sub_u16, A4("v_pk_sub_u16"))
KERNEL(k_min3_u32, A4_3("v_min3_u32"))
KERNEL(k_max_u32, A4("v_max_u32"))
KERNEL(k_cmp_lt_u32, asm volatile("v_cmp_lt_u32 vcc, %0, %4\nv_cmp_lt_u32 vcc, %1, %5\nv_cmp_lt_u32 vcc, %2, %6\nv_cmp_lt_u32 vcc, %3, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h) : "vcc");)
KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %4, vcc\nv_cndmask_b32 %1, %1, %5, vcc\nv_cndmask_b32 %2, %2, %6, vcc\nv_cndmask_b32 %3, %3, %7, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h) : "vcc");)
KERNEL(k_mov_dpp, asm volatile("v_mov_b32_dpp %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %5 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %2, %6 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %7 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));)

KERNEL(k_and_b32, A4("v_and_b32"))
KERNEL(k_or_b32, A4("v_or_b32"))
KERNEL(k_lshlrev_b32, A4("v_lshlrev_b32"))
KERNEL(k_lshrrev_b32, A4("v_lshrrev_b32"))
KERNEL(k_ashrrev_i32, A4("v_ashrrev_i32"))
KERNEL(k_sub_u32, A4("v_sub_u32"))
KERNEL(k_add_f32, A4("v_add_f32"))
KERNEL(k_sub_f32, A4("v_sub_f32"))
KERNEL(k_fmac_f32, A4("v_fmac_f32"))
KERNEL(k_min_f32, A4("v_min_f32"))
KERNEL(k_min_u32, A4("v_min_u32"))
KERNEL(k_mov_b32, A4_1("v_mov_b32"))
KERNEL(k_cvt_f32_ubyte0, A4_1("v_cvt_f32_ubyte0"))
KERNEL(k_cvt_f32_ubyte3, A4_1("v_cvt_f32_ubyte3"))
KERNEL(k_cvt_f32_u32, A4_1("v_cvt_f32_u32"))
KERNEL(k_cvt_u32_f32, A4_1("v_cvt_u32_f32"))
KERNEL(k_cvt_pk_u8_f32, A4_3("v_cvt_pk_u8_f32"))
KERNEL(k_add3_u32, A4_3("v_add3_u32"))
KERNEL(k_add_lshl_u32, A4_3("v_add_lshl_u32"))
KERNEL(k_xad_u32, A4_3("v_xad_u32"))
KERNEL(k_bfi_b32, A4_3("v_bfi_b32"))
KERNEL(k_med3_i32, A4_3("v_med3_i32"))
KERNEL(k_sad_u8, A4_3("v_sad_u8"))
KERNEL(k_mad_u16, A4_3("v_mad_u16"))
KERNEL(k_mad_mix_f32, A4_3("v_fma_mix_f32"))
KERNEL(k_pk_add_u16, A4("v_pk_add_u16"))
KERNEL(k_pk_min_u16, A4("v_pk_min_u16"))
KERNEL(k_cmp_cnd, asm volatile("v_cmp_lt_u32 vcc, %0, %4\nv_cndmask_b32 %1, %1, %5, vcc\nv_cmp_lt_u32 vcc, %2, %6\nv_cndmask_b32 %3, %3, %7, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h) : "vcc");)
KERNEL(k_cmp_e64, asm volatile("v_cmp_lt_u32 s[20:21], %0, %4\nv_cmp_lt_u32 s[22:23], %1, %5\nv_cmp_lt_u32 s[24:25], %2, %6\nv_cmp_lt_u32 s[26:27], %3, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h) : "s20","s21","s22","s23","s24","s25","s26","s27");)
KERNEL(k_mix_fma_perm, asm volatile("v_mul_f32 %0, %0, %4\nv_perm_b32 %1, %1, %5, %6\nv_mul_f32 %2, %2, %6\nv_perm_b32 %3, %3, %7, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));)
KERNEL(k_readlane, asm volatile("v_readfirstlane_b32 s20, %0\nv_readfirstlane_b32 s21, %1\nv_readfirstlane_b32 s22, %2\nv_readfirstlane_b32 s23, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h) : "s20","s21","s22","s23");)

// ---- dependent chains and phase mixes (what a real kernel looks like) -------------------------------------------------------------
#define REP8(X) X X X X X X X X
KERNEL(k_fma_chain1, asm volatile("v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %0, %0, %5, %6\nv_fma_f32 %0, %0, %6, %7\nv_fma_f32 %0, %0, %7, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));)
KERNEL(k_fma_chain2, asm volatile("v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %5, %6\nv_fma_f32 %0, %0, %6, %7\nv_fma_f32 %1, %1, %7, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));)
KERNEL(k_perm_chain1, asm volatile("v_perm_b32 %0, %0, %4, %5\nv_perm_b32 %0, %0, %5, %6\nv_perm_b32 %0, %0, %6, %7\nv_perm_b32 %0, %0, %7, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));)
// 32 A-class then 32 B-class per body repetition (64 instructions like the other bodies): waves drift apart, so a SIMD sees both kinds at once
#define PH_A asm volatile("v_fma_f32 %0, %0, %4, %5\nv_mul_f32 %1, %1, %5\nv_add_f32 %2, %2, %6\nv_fma_f32 %3, %3, %7, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));
#define PH_B asm volatile("v_perm_b32 %0, %0, %4, %5\nv_dot4_u32_u8 %1, %1, %5, %6\nv_alignbyte_b32 %2, %2, %6, %7\nv_mad_u32_u24 %3, %3, %7, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));
__global__ __launch_bounds__(256) void k_phase_ab(int iters, uint32_t *sink, unsigned long long *clk)
{
    uint32_t a = threadIdx.x * 2654435761u + 12345u, b = a ^ 0x5bd1e995u, c = a + 77u, d = b + 99u;
    uint32_t e = a * 3u, f = b * 5u, g = c * 7u, h = d * 9u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x & 1) { REP8(PH_A) }              // odd blocks start half a period later
    for (int i = 0; i < iters; ++i) { REP8(PH_A) REP8(PH_B) REP8(PH_A) REP8(PH_B) REP8(PH_A) REP8(PH_B) REP8(PH_A) REP8(PH_B) }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((a ^ b ^ c ^ d) == 0x1234567u) sink[0] = a;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
__global__ __launch_bounds__(256) void k_phase_aa(int iters, uint32_t *sink, unsigned long long *clk)
{
    uint32_t a = threadIdx.x * 2654435761u + 12345u, b = a ^ 0x5bd1e995u, c = a + 77u, d = b + 99u;
    uint32_t e = a * 3u, f = b * 5u, g = c * 7u, h = d * 9u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { REP8(PH_A) REP8(PH_A) REP8(PH_A) REP8(PH_A) REP8(PH_A) REP8(PH_A) REP8(PH_A) REP8(PH_A) }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((a ^ b ^ c ^ d) == 0x1234567u) sink[0] = a;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
__global__ __launch_bounds__(256) void k_phase_bb(int iters, uint32_t *sink, unsigned long long *clk)
{
    uint32_t a = threadIdx.x * 2654435761u + 12345u, b = a ^ 0x5bd1e995u, c = a + 77u, d = b + 99u;
    uint32_t e = a * 3u, f = b * 5u, g = c * 7u, h = d * 9u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { REP8(PH_B) REP8(PH_B) REP8(PH_B) REP8(PH_B) REP8(PH_B) REP8(PH_B) REP8(PH_B) REP8(PH_B) }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((a ^ b ^ c ^ d) == 0x1234567u) sink[0] = a;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// ---- same-wave interleaving of the two instruction kinds (4 instructions per asm block like the others) ---------------------------------
KERNEL(k_alt_abab_dep, asm volatile("v_fma_f32 %0, %0, %4, %5\nv_perm_b32 %1, %1, %5, %6\nv_fma_f32 %0, %0, %6, %7\nv_perm_b32 %1, %1, %7, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));)
KERNEL(k_alt_aabb, asm volatile("v_fma_f32 %0, %0, %4, %5\nv_mul_f32 %2, %2, %6\nv_perm_b32 %1, %1, %5, %6\nv_dot4_u32_u8 %3, %3, %7, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));)
KERNEL(k_alt_aaab, asm volatile("v_fma_f32 %0, %0, %4, %5\nv_mul_f32 %2, %2, %6\nv_add_f32 %1, %1, %5\nv_dot4_u32_u8 %3, %3, %7, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));)
KERNEL(k_alt_abbb, asm volatile("v_fma_f32 %0, %0, %4, %5\nv_perm_b32 %2, %2, %6, %7\nv_alignbyte_b32 %1, %1, %5, %6\nv_dot4_u32_u8 %3, %3, %7, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));)
KERNEL(k_alt_sbsb, asm volatile("s_add_u32 s20, s20, 1\nv_perm_b32 %1, %1, %5, %6\ns_add_u32 s21, s21, 1\nv_dot4_u32_u8 %3, %3, %7, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h) : "s20", "s21", "scc");)
KERNEL(k_alt_pkb, asm volatile("v_pk_fma_f32 %0, %0, %2, %3\nv_perm_b32 %4, %4, %5, %6\nv_pk_fma_f32 %1, %1, %3, %2\nv_perm_b32 %5, %5, %6, %4" : "+v"(p0), "+v"(p1) : "v"(p2), "v"(p3), "v"(a), "v"(b), "v"(c));)

typedef void (*kern_t)(int, uint32_t *, unsigned long long *);

static void run(const char *name, kern_t k, int waves_per_simd, int ops_per_inst, uint32_t *sink, unsigned long long *clk)
{
    const int iters = 2000, blocks = 256 * waves_per_simd;      // one 256-thread block = one wave per SIMD of a CU
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, iters, sink, clk);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, iters, sink, clk);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];          // s_memtime ticks at 100 MHz on gfx9: convert with the wall clock below
    const double insts = (double)iters * 64.0;           // per wave
    // wall-clock view: all waves of a SIMD share it, so SIMD-cycles per wave-instruction = ms * f / (insts * waves_per_simd)
    printf("%-18s waves/SIMD %d : %6.2f ns per wave-instr per SIMD (wall)   wave-time median %8.0f ticks  -> %5.2f ticks/instr/wave  x%d ops\n", name, waves_per_simd,
           ms * 1e6 / (insts * waves_per_simd), med, med / insts, ops_per_inst);
}

int main()
{
    uint32_t *sink; unsigned long long *clk;
    hipMalloc(&sink, 256); hipMalloc(&clk, sizeof(unsigned long long) * 256 * 8 * 4);
#define RUN(K, OPS) run(#K, K, 1, OPS, sink, clk); run(#K, K, 4, OPS, sink, clk); run(#K, K, 8, OPS, sink, clk);
    RUN(k_add_u32, 1) RUN(k_fma_f32, 1) RUN(k_mul_f32, 1) RUN(k_pk_fma_f32, 2) RUN(k_pk_mul_f32, 2) RUN(k_pk_add_f32, 2) RUN(k_rcp_f32, 1) RUN(k_rndne_f32, 1)
    RUN(k_cvt_i32_f32, 1) RUN(k_dot4_u32_u8, 4) RUN(k_dot2_u32_u16, 2) RUN(k_perm_b32, 1) RUN(k_alignbyte, 1) RUN(k_alignbit, 1) RUN(k_mul_u32_u24, 1)
    RUN(k_mad_u32_u24, 1) RUN(k_mul_lo_u32, 1) RUN(k_mad_u64_u32, 1) RUN(k_mul_hi_u32, 1) RUN(k_lshl_or, 1) RUN(k_lshl_add, 1) RUN(k_and_or, 1) RUN(k_or3, 1) RUN(k_bfe_u32, 1) RUN(k_xor, 1)
    RUN(k_pk_mad_u16, 2) RUN(k_pk_mul_lo_u16, 2) RUN(k_pk_lshrrev_b16, 2) RUN(k_pk_sub_u16, 2) RUN(k_min3_u32, 1) RUN(k_max_u32, 1)
    RUN(k_cmp_lt_u32, 1) RUN(k_cndmask, 1) RUN(k_mov_dpp, 1)
    RUN(k_and_b32, 1) RUN(k_or_b32, 1) RUN(k_lshlrev_b32, 1) RUN(k_lshrrev_b32, 1) RUN(k_ashrrev_i32, 1) RUN(k_sub_u32, 1) RUN(k_add_f32, 1) RUN(k_sub_f32, 1) RUN(k_fmac_f32, 1) RUN(k_min_f32, 1) RUN(k_min_u32, 1) RUN(k_mov_b32, 1) RUN(k_cvt_f32_ubyte0, 1) RUN(k_cvt_f32_ubyte3, 1) RUN(k_cvt_f32_u32, 1) RUN(k_cvt_u32_f32, 1) RUN(k_cvt_pk_u8_f32, 1) RUN(k_add3_u32, 1) RUN(k_add_lshl_u32, 1) RUN(k_xad_u32, 1) RUN(k_bfi_b32, 1) RUN(k_med3_i32, 1) RUN(k_sad_u8, 1) RUN(k_mad_u16, 1) RUN(k_mad_mix_f32, 1) RUN(k_pk_add_u16, 1) RUN(k_pk_min_u16, 1) RUN(k_cmp_cnd, 1) RUN(k_cmp_e64, 1) RUN(k_mix_fma_perm, 1) RUN(k_readlane, 1)
    RUN(k_alt_abab_dep, 1) RUN(k_alt_aabb, 1) RUN(k_alt_aaab, 1) RUN(k_alt_abbb, 1) RUN(k_alt_sbsb, 1) RUN(k_alt_pkb, 1)
    RUN(k_fma_chain1, 1) RUN(k_fma_chain2, 1) RUN(k_perm_chain1, 1)
    // phase kernels issue 4x the instructions of the others per iteration (256): divide their figures by 4
    RUN(k_phase_aa, 4) RUN(k_phase_bb, 4) RUN(k_phase_ab, 4)
    return 0;
}
