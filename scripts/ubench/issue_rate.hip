// Micro-benchmark: cycles per wave64 instruction on one SIMD for the instruction kinds the render
// kernel is made of, at 1, 2, 4 and 8 resident waves per SIMD.  Each kernel runs ITER iterations of a
// hand-written block of 32 independent instructions (8 accumulator chains, four rounds), so the
// figure is issue throughput, not latency.  Every wave stamps the loop with s_memtime (shader clock)
// and s_memrealtime (100 MHz), so the table shows SIMD cycles of the clock the chip really held and
// that clock, not cycles of a nominal one.
//   hipcc --offload-arch=gfx950 -O3 -o build/issue_rate scripts/ubench/issue_rate.hip && build/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ITER = 2048;

struct Stamp { unsigned long long cycles, real; };

#define R4(x) x x x x
#define ROUND8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)

#define FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n\t"
#define FMAC(i) "v_fmac_f32_e32 %" #i ", %8, %9\n\t"
#define FMAS(i) "v_fma_f32 %" #i ", %" #i ", %10, %9\n\t"
#define ADD(i) "v_add_f32_e32 %" #i ", %" #i ", %8\n\t"
#define MUL(i) "v_mul_f32_e32 %" #i ", %" #i ", %8\n\t"
#define MULNEG(i) "v_mul_f32_e64 %" #i ", %" #i ", -%8\n\t"
#define SUBS(i) "v_subrev_f32_e32 %" #i ", %10, %" #i "\n\t"
#define RCP(i) "v_rcp_f32_e32 %" #i ", %" #i "\n\t"
#define SQRT(i) "v_sqrt_f32_e32 %" #i ", %" #i "\n\t"
#define MOV(i) "v_mov_b32_e32 %" #i ", %8\n\t"
#define CMP(i) "v_cmp_lt_f32_e32 vcc, %" #i ", %8\n\t"
#define CMPS(i) "v_cmp_lt_f32_e64 s[20:21], %" #i ", %8\n\t"
#define CMPX(i) "v_cmpx_lt_f32_e32 %8, %" #i "\n\t"
#define CMPXABS(i) "v_cmpx_le_f32_e64 exec, 1.0, |%" #i "|\n\t"
#define CNDMASK(i) "v_cmp_lt_f32_e32 vcc, %" #i ", %8\n\tv_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n\t"
#define MIN(i) "v_min_f32_e32 %" #i ", %" #i ", %8\n\t"
#define MAX3(i) "v_max3_f32 %" #i ", %" #i ", %8, %9\n\t"
#define MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n\t"
#define XOR(i) "v_xor_b32_e32 %" #i ", %" #i ", %8\n\t"
#define LSHL(i) "v_lshlrev_b32_e32 %" #i ", 13, %" #i "\n\t"
#define LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 13, %" #i "\n\t"
#define LSHLV(i) "v_lshlrev_b32_e32 %" #i ", %8, %" #i "\n\t"
#define ANDV(i) "v_and_b32_e32 %" #i ", %" #i ", %8\n\t"
#define ORV(i) "v_or_b32_e32 %" #i ", %" #i ", %8\n\t"
#define ADDU(i) "v_add_u32_e32 %" #i ", %" #i ", %8\n\t"
#define SUBV(i) "v_sub_f32_e32 %" #i ", %" #i ", %8\n\t"
#define ADDC(i) "v_add_f32_e32 %" #i ", 1.0, %" #i "\n\t"
#define MULC(i) "v_mul_f32_e32 %" #i ", 2.0, %" #i "\n\t"
#define FMAL(i) "v_fma_f32 %" #i ", %" #i ", %8, 1.0\n\t"
#define FMAABS(i) "v_fma_f32 %" #i ", |%" #i "|, %8, %9\n\t"
#define ADDE64(i) "v_add_f32_e64 %" #i ", %" #i ", %8\n\t"
#define MOVS(i) "v_mov_b32_e32 %" #i ", %10\n\t"
#define MOVC(i) "v_mov_b32_e32 %" #i ", 1.0\n\t"
#define BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 5\n\t"
#define MIN_ADD(i) "v_min_f32_e32 %" #i ", %" #i ", %8\n\tv_add_f32_e32 %" #i ", %" #i ", %9\n\t"
#define CMP_ADD(i) "v_cmp_lt_f32_e32 vcc, %" #i ", %8\n\tv_add_f32_e32 %" #i ", %" #i ", %9\n\t"
#define PK_ADD1(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n\t"
#define CVT(i) "v_cvt_f32_u32_e32 %" #i ", %" #i "\n\t"
#define SALU(i) "s_add_u32 s22, s22, 1\n\t"
#define SMOV(i) "s_mov_b64 s[22:23], s[20:21]\n\t"
#define FMA_SALU(i) FMA(i) SALU(i)
#define FMA_SALU2(i) FMA(i) SALU(i) SALU(i)
#define FMA_SMOV(i) FMA(i) SMOV(i)
#define PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n\t"
#define PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n\t"
#define PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %9\n\t"
#define PKFMAS(i) "v_pk_fma_f32 %" #i ", %" #i ", %10, %9\n\t"
#define PK_SALU(i) PKFMA(i) SALU(i)
#define PK_SALU2(i) PKFMA(i) SALU(i) SALU(i)

#define STAMP_BEGIN                                                                      \
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#define STAMP_END                                                                        \
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
    if ((threadIdx.x & 63) == 0) {                                                       \
        Stamp st; st.cycles = c1 - c0; st.real = r1 - r0;                                \
        stamps[blockIdx.x * 4 + threadIdx.x / 64] = st;                                  \
    }

#define SCALAR_KERNEL(name, OP, PRE, POST)                                                              \
    __global__ __launch_bounds__(256) void name(float* out, float a, float b, float s, Stamp* stamps) { \
        float x0 = threadIdx.x * 0.001f + 2.0f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4,      \
              x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;                                                     \
        STAMP_BEGIN                                                                                      \
        for (int it = 0; it < ITER; ++it)                                                                \
            asm volatile(PRE R4(ROUND8(OP)) POST                                                         \
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) \
                         : "v"(a), "v"(b), "s"(s)                                                        \
                         : "vcc", "scc", "s20", "s21", "s22", "s23");                                    \
        STAMP_END                                                                                        \
        out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                     \
    }

typedef float F2 __attribute__((ext_vector_type(2)));
#define PACKED_KERNEL(name, OP)                                                                         \
    __global__ __launch_bounds__(256) void name(float* out, float a, float b, float s, Stamp* stamps) { \
        F2 x0 = {threadIdx.x * 0.001f + 1.0f, 0.5f}, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, \
           x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;                                                        \
        F2 a2 = {a, a * 1.0001f}, b2 = {b, b * 0.999f};                                                  \
        F2 s2 = {s, s};                                                                                  \
        STAMP_BEGIN                                                                                      \
        for (int it = 0; it < ITER; ++it)                                                                \
            asm volatile(R4(ROUND8(OP))                                                                  \
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) \
                         : "v"(a2), "v"(b2), "s"(s2)                                                     \
                         : "vcc", "scc", "s22");                                                         \
        STAMP_END                                                                                        \
        F2 t = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                                                    \
        out[blockIdx.x * 256 + threadIdx.x] = t.x + t.y;                                                 \
    }

#define SAVE_EXEC "s_mov_b64 s[20:21], exec\n\t"
#define RESTORE_EXEC "s_mov_b64 exec, s[20:21]\n\t"

SCALAR_KERNEL(k_fma, FMA, "", "")
SCALAR_KERNEL(k_fmac, FMAC, "", "")
SCALAR_KERNEL(k_fma_sgpr, FMAS, "", "")
SCALAR_KERNEL(k_add, ADD, "", "")
SCALAR_KERNEL(k_mul, MUL, "", "")
SCALAR_KERNEL(k_mulneg, MULNEG, "", "")
SCALAR_KERNEL(k_sub_sgpr, SUBS, "", "")
SCALAR_KERNEL(k_rcp, RCP, "", "")
SCALAR_KERNEL(k_sqrt, SQRT, "", "")
SCALAR_KERNEL(k_mov, MOV, "", "")
SCALAR_KERNEL(k_cmp_vcc, CMP, "", "")
SCALAR_KERNEL(k_cmp_sgpr, CMPS, "", "")
SCALAR_KERNEL(k_cndmask, CNDMASK, "", "")
SCALAR_KERNEL(k_min, MIN, "", "")
SCALAR_KERNEL(k_max3, MAX3, "", "")
SCALAR_KERNEL(k_med3, MED3, "", "")
SCALAR_KERNEL(k_xor, XOR, "", "")
SCALAR_KERNEL(k_lshl, LSHL, "", "")
SCALAR_KERNEL(k_lshl_add, LSHLADD, "", "")
SCALAR_KERNEL(k_cvt, CVT, "", "")
SCALAR_KERNEL(k_lshlv, LSHLV, "", "")
SCALAR_KERNEL(k_andv, ANDV, "", "")
SCALAR_KERNEL(k_orv, ORV, "", "")
SCALAR_KERNEL(k_addu, ADDU, "", "")
SCALAR_KERNEL(k_subv, SUBV, "", "")
SCALAR_KERNEL(k_addc, ADDC, "", "")
SCALAR_KERNEL(k_mulc, MULC, "", "")
SCALAR_KERNEL(k_fmal, FMAL, "", "")
SCALAR_KERNEL(k_fmaabs, FMAABS, "", "")
SCALAR_KERNEL(k_adde64, ADDE64, "", "")
SCALAR_KERNEL(k_movs, MOVS, "", "")
SCALAR_KERNEL(k_movc, MOVC, "", "")
SCALAR_KERNEL(k_bfe, BFE, "", "")
SCALAR_KERNEL(k_min_add, MIN_ADD, "", "")
SCALAR_KERNEL(k_cmp_add, CMP_ADD, "", "")
SCALAR_KERNEL(k_fma_salu, FMA_SALU, "", "")
SCALAR_KERNEL(k_fma_salu2, FMA_SALU2, "", "")
SCALAR_KERNEL(k_fma_smov, FMA_SMOV, "", "")
SCALAR_KERNEL(k_cmpx, CMPX, SAVE_EXEC, RESTORE_EXEC)
SCALAR_KERNEL(k_cmpx_abs, CMPXABS, SAVE_EXEC, RESTORE_EXEC)
PACKED_KERNEL(k_pk_fma, PKFMA)
PACKED_KERNEL(k_pk_mul, PKMUL)
PACKED_KERNEL(k_pk_add, PKADD)
PACKED_KERNEL(k_pk_fma_sgpr, PKFMAS)
PACKED_KERNEL(k_pk_salu, PK_SALU)
PACKED_KERNEL(k_pk_salu2, PK_SALU2)

// One packed instruction (slow pipe) beside NF scalar fast-path instructions, on independent registers.
#define MIXED_KERNEL(name, PKOPS, FASTOPS)                                                              \
    __global__ __launch_bounds__(256) void name(float* out, float a, float b, float s, Stamp* stamps) { \
        F2 x0 = {threadIdx.x * 0.001f + 1.0f, 0.5f}, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;              \
        float y0 = threadIdx.x * 0.002f + 2.0f, y1 = y0 + 1, y2 = y0 + 2, y3 = y0 + 3;                   \
        F2 a2 = {a, a * 1.0001f}, b2 = {b, b * 0.999f};                                                  \
        STAMP_BEGIN                                                                                      \
        for (int it = 0; it < ITER; ++it)                                                                \
            asm volatile(R4(R4(PKOPS FASTOPS))                                                           \
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) \
                         : "v"(a2), "v"(b2), "v"(a), "v"(b)                                              \
                         : "vcc", "scc");                                                                \
        STAMP_END                                                                                        \
        F2 t = x0 + x1 + x2 + x3;                                                                        \
        out[blockIdx.x * 256 + threadIdx.x] = t.x + t.y + y0 + y1 + y2 + y3;                             \
    }
#define PK0 "v_pk_fma_f32 %0, %0, %8, %9\n\t"
#define PK1 "v_pk_fma_f32 %1, %1, %8, %9\n\t"
#define F4 "v_fma_f32 %4, %4, %10, %11\n\t"
#define F5 "v_fma_f32 %5, %5, %10, %11\n\t"
#define F6 "v_fma_f32 %6, %6, %10, %11\n\t"
#define F7 "v_fma_f32 %7, %7, %10, %11\n\t"
#define RCP4 "v_rcp_f32_e32 %4, %4\n\t"
#define CMP4 "v_cmp_lt_f32_e32 vcc, %4, %10\n\t"
#define A4 "v_add_f32_e32 %4, %4, %10\n\t"
#define A5 "v_add_f32_e32 %5, %5, %10\n\t"
#define M6 "v_min_f32_e32 %6, %6, %10\n\t"
#define M7 "v_min_f32_e32 %7, %7, %10\n\t"
#define C6 "v_cmp_lt_f32_e32 vcc, %6, %10\n\t"
#define PKM0 "v_pk_mul_f32 %0, %0, %8\n\t"
#define PKA0 "v_pk_add_f32 %0, %0, %9\n\t"
#define MUL5 "v_mul_f32_e32 %5, %5, %10\n\t"
MIXED_KERNEL(k_min_add_indep, M6, A4)
MIXED_KERNEL(k_min_fma_indep, M6, F4)
MIXED_KERNEL(k_cmp_fma_indep, C6, F4)
MIXED_KERNEL(k_pkfma_add, PK0, A4)
MIXED_KERNEL(k_pkmul_add, PKM0, A4)
MIXED_KERNEL(k_pkadd_add, PKA0, A4)
MIXED_KERNEL(k_pkadd_mul, PKA0, MUL5)
MIXED_KERNEL(k_fma_add, F4, A5)
MIXED_KERNEL(k_min_min, M6, M7)
MIXED_KERNEL(k_pk1_f1, PK0, F4)
MIXED_KERNEL(k_pk1_f2, PK0, F4 F5)
MIXED_KERNEL(k_pk1_f3, PK0, F4 F5 F6)
MIXED_KERNEL(k_pk2_f1, PK0 PK1, F4)
MIXED_KERNEL(k_rcp1_f3, RCP4, F5 F6 F7)
MIXED_KERNEL(k_rcp1_pk1_f2, RCP4 PK0, F5 F6)
MIXED_KERNEL(k_cmp1_pk1_f2, CMP4 PK0, F5 F6)

// Independent chains: 4 packed registers (%0-%3) and 4 scalar ones (%4-%7); BLOCK names each register once.
#define MIXED4_KERNEL(name, BLOCK)                                                                      \
    __global__ __launch_bounds__(256) void name(float* out, float a, float b, float s, Stamp* stamps) { \
        F2 x0 = {threadIdx.x * 0.001f + 1.0f, 0.5f}, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;              \
        float y0 = threadIdx.x * 0.002f + 2.0f, y1 = y0 + 1, y2 = y0 + 2, y3 = y0 + 3;                   \
        F2 a2 = {a, a * 1.0001f}, b2 = {b, b * 0.999f};                                                  \
        STAMP_BEGIN                                                                                      \
        for (int it = 0; it < ITER; ++it)                                                                \
            asm volatile(R4(BLOCK)                                                                       \
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) \
                         : "v"(a2), "v"(b2), "v"(a), "v"(b)                                              \
                         : "vcc", "scc");                                                                \
        STAMP_END                                                                                        \
        F2 t = x0 + x1 + x2 + x3;                                                                        \
        out[blockIdx.x * 256 + threadIdx.x] = t.x + t.y + y0 + y1 + y2 + y3;                             \
    }
#define PKF(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n\t"
#define SF(i) "v_fma_f32 %" #i ", %" #i ", %10, %11\n\t"
#define SA(i) "v_add_f32_e32 %" #i ", %" #i ", %10\n\t"
#define SMN(i) "v_min_f32_e32 %" #i ", %" #i ", %10\n\t"
#define SCM(i) "v_cmp_lt_f32_e32 vcc, %" #i ", %10\n\t"
#define SRC(i) "v_rcp_f32_e32 %" #i ", %" #i "\n\t"
MIXED4_KERNEL(k4_pk_f, PKF(0) SF(4) PKF(1) SF(5) PKF(2) SF(6) PKF(3) SF(7))
MIXED4_KERNEL(k4_pk_2f, PKF(0) SF(4) SF(5) PKF(1) SF(6) SF(7) PKF(2) SF(4) SF(5) PKF(3) SF(6) SF(7))
MIXED4_KERNEL(k4_f_a, SF(4) SA(5) SF(6) SA(7) SF(4) SA(5) SF(6) SA(7))
MIXED4_KERNEL(k4_mn_f, SMN(4) SF(5) SMN(6) SF(7) SMN(4) SF(5) SMN(6) SF(7))
MIXED4_KERNEL(k4_mn_2f, SMN(4) SF(5) SF(6) SMN(7) SF(4) SF(5) SMN(6) SF(7) SF(4) SMN(5) SF(6) SF(7))
MIXED4_KERNEL(k4_cm_f, SCM(4) SF(5) SCM(6) SF(7) SCM(4) SF(5) SCM(6) SF(7))
MIXED4_KERNEL(k4_rc_3f, SRC(4) SF(5) SF(6) SF(7) SRC(5) SF(6) SF(7) SF(4))
MIXED4_KERNEL(k4_pk_mn, PKF(0) SMN(4) PKF(1) SMN(5) PKF(2) SMN(6) PKF(3) SMN(7))
MIXED4_KERNEL(k4_f_only, SF(4) SF(5) SF(6) SF(7) SF(4) SF(5) SF(6) SF(7))

template <typename K>
void run(const char* name, K kernel, double valu_per_block, float* d, Stamp* d_stamps) {
    printf("%-12s", name);
    for (int blocks : {256, 512, 1024, 2048}) {   // 1, 2, 4, 8 waves per SIMD
        (void)hipMemset(d_stamps, 0, sizeof(Stamp) * 4 * 2048);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, 0.25f, d_stamps);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, 0.25f, d_stamps);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<Stamp> stamps(4 * blocks);
        (void)hipMemcpy(stamps.data(), d_stamps, sizeof(Stamp) * stamps.size(), hipMemcpyDeviceToHost);
        double cycles = 0, real = 0;
        for (const Stamp& st : stamps) { cycles += (double)st.cycles; real += (double)st.real; }
        const int waves_per_simd = blocks / 256;
        // wall time of the launch x the clock the waves measured = SIMD cycles; each SIMD ran waves_per_simd waves
        const double clock_ghz = cycles / real * 0.1;
        const double per_inst = ms * 1e-3 * clock_ghz * 1e9 / (valu_per_block * ITER * waves_per_simd);
        printf("  %dw %5.2f @%4.2f", waves_per_simd, per_inst, clock_ghz);
    }
    printf("\n");
    fflush(stdout);
}

int main() {
    float* d;
    Stamp* d_stamps;
    (void)hipMalloc(&d, 256 * 4096 * sizeof(float));
    (void)hipMalloc(&d_stamps, sizeof(Stamp) * 4 * 2048);
    printf("SIMD cycles per VALU wave-instruction @ clock held (GHz), by resident waves per SIMD\n");
    run("fma", k_fma, 32, d, d_stamps);
    run("fmac", k_fmac, 32, d, d_stamps);
    run("fma sgpr", k_fma_sgpr, 32, d, d_stamps);
    run("add", k_add, 32, d, d_stamps);
    run("mul", k_mul, 32, d, d_stamps);
    run("mul neg", k_mulneg, 32, d, d_stamps);
    run("sub sgpr", k_sub_sgpr, 32, d, d_stamps);
    run("pk_fma", k_pk_fma, 32, d, d_stamps);
    run("pk_fma sgpr", k_pk_fma_sgpr, 32, d, d_stamps);
    run("pk_mul", k_pk_mul, 32, d, d_stamps);
    run("pk_add", k_pk_add, 32, d, d_stamps);
    run("rcp", k_rcp, 32, d, d_stamps);
    run("sqrt", k_sqrt, 32, d, d_stamps);
    run("mov", k_mov, 32, d, d_stamps);
    run("cmp vcc", k_cmp_vcc, 32, d, d_stamps);
    run("cmp sgpr", k_cmp_sgpr, 32, d, d_stamps);
    run("cmpx", k_cmpx, 32, d, d_stamps);
    run("cmpx abs", k_cmpx_abs, 32, d, d_stamps);
    run("cmp+cndmask", k_cndmask, 64, d, d_stamps);
    run("min", k_min, 32, d, d_stamps);
    run("max3", k_max3, 32, d, d_stamps);
    run("med3", k_med3, 32, d, d_stamps);
    run("xor", k_xor, 32, d, d_stamps);
    run("lshl", k_lshl, 32, d, d_stamps);
    run("lshl_add", k_lshl_add, 32, d, d_stamps);
    run("cvt_f32_u32", k_cvt, 32, d, d_stamps);
    run("lshl vgpr", k_lshlv, 32, d, d_stamps);
    run("and", k_andv, 32, d, d_stamps);
    run("or", k_orv, 32, d, d_stamps);
    run("add_u32", k_addu, 32, d, d_stamps);
    run("sub", k_subv, 32, d, d_stamps);
    run("add 1.0", k_addc, 32, d, d_stamps);
    run("mul 2.0", k_mulc, 32, d, d_stamps);
    run("fma c=1.0", k_fmal, 32, d, d_stamps);
    run("fma |a|", k_fmaabs, 32, d, d_stamps);
    run("add e64", k_adde64, 32, d, d_stamps);
    run("mov sgpr", k_movs, 32, d, d_stamps);
    run("mov 1.0", k_movc, 32, d, d_stamps);
    run("bfe", k_bfe, 32, d, d_stamps);
    run("min+add", k_min_add, 64, d, d_stamps);
    run("cmp+add", k_cmp_add, 64, d, d_stamps);
    run("min,add ind", k_min_add_indep, 32, d, d_stamps);
    run("min,fma ind", k_min_fma_indep, 32, d, d_stamps);
    run("cmp,fma ind", k_cmp_fma_indep, 32, d, d_stamps);
    run("pkfma,add", k_pkfma_add, 32, d, d_stamps);
    run("pkmul,add", k_pkmul_add, 32, d, d_stamps);
    run("pkadd,add", k_pkadd_add, 32, d, d_stamps);
    run("pkadd,mul", k_pkadd_mul, 32, d, d_stamps);
    run("fma,add", k_fma_add, 32, d, d_stamps);
    run("min,min", k_min_min, 32, d, d_stamps);
    run("4: pk,f", k4_pk_f, 32, d, d_stamps);
    run("4: pk,f,f", k4_pk_2f, 48, d, d_stamps);
    run("4: f,add", k4_f_a, 32, d, d_stamps);
    run("4: min,f", k4_mn_f, 32, d, d_stamps);
    run("4: min,f,f", k4_mn_2f, 48, d, d_stamps);
    run("4: cmp,f", k4_cm_f, 32, d, d_stamps);
    run("4: rcp,f,f,f", k4_rc_3f, 32, d, d_stamps);
    run("4: pk,min", k4_pk_mn, 32, d, d_stamps);
    run("4: f only", k4_f_only, 32, d, d_stamps);
    run("pk + 1 fma", k_pk1_f1, 32, d, d_stamps);
    run("pk + 2 fma", k_pk1_f2, 48, d, d_stamps);
    run("pk + 3 fma", k_pk1_f3, 64, d, d_stamps);
    run("2 pk + 1 fma", k_pk2_f1, 48, d, d_stamps);
    run("rcp + 3 fma", k_rcp1_f3, 64, d, d_stamps);
    run("rcp+pk+2fma", k_rcp1_pk1_f2, 64, d, d_stamps);
    run("cmp+pk+2fma", k_cmp1_pk1_f2, 64, d, d_stamps);
    run("fma+1salu", k_fma_salu, 32, d, d_stamps);
    run("fma+2salu", k_fma_salu2, 32, d, d_stamps);
    run("fma+smov64", k_fma_smov, 32, d, d_stamps);
    run("pkfma+1salu", k_pk_salu, 32, d, d_stamps);
    run("pkfma+2salu", k_pk_salu2, 32, d, d_stamps);
    return 0;
}
