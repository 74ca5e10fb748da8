// Issue cost of the vector instructions the scan and insert kernels are made of, measured on the device:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates tools/valu_rates.hip && /tmp/valu_rates
// Every kernel runs N_ITER x 16 copies of ONE instruction on 8 independent register chains per lane, 4 waves per SIMD on
// every SIMD; cycles per wave-instruction = time x clock / (instructions per wave x waves per SIMD).  Reported next to
// v_add_u32 (= 1.0) because the kernels' time is read as SQ_INSTS_VALU x 4 cycles: an instruction that costs more than one
// issue slot counts for more than one there.  (A tool: nothing in the library or the tests depends on it.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32;
typedef unsigned long long u64;

#define N_ITER 2048
#define REP16(X) X X X X X X X X X X X X X X X X

#define KERNEL32(NAME, ASM)                                                                             \
    __global__ void __launch_bounds__(256) NAME(u32* out, u32 seed) {                                   \
        u32 a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 ^ 13, a6 = a0 + 17, a7 = a0 ^ 19; \
        u32 b = seed | 3;                                                                               \
        for (int i = 0; i < N_ITER; i++) {                                                              \
            asm volatile(ASM("%0") ASM("%1") ASM("%2") ASM("%3") ASM("%4") ASM("%5") ASM("%6") ASM("%7") ASM("%0") ASM("%1") ASM("%2") ASM("%3") ASM("%4") ASM("%5") ASM("%6") ASM("%7") \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)  \
                         : "v"(b)                                                                       \
                         : "vcc", "s10", "s11");                                                                      \
        }                                                                                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;             \
    }
#define KERNEL64(NAME, ASM)                                                                             \
    __global__ void __launch_bounds__(256) NAME(u32* out, u32 seed) {                                   \
        u64 a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 ^ 13, a6 = a0 + 17, a7 = a0 ^ 19; \
        u64 b = seed | 3;                                                                               \
        for (int i = 0; i < N_ITER; i++) {                                                              \
            asm volatile(ASM("%0") ASM("%1") ASM("%2") ASM("%3") ASM("%4") ASM("%5") ASM("%6") ASM("%7") ASM("%0") ASM("%1") ASM("%2") ASM("%3") ASM("%4") ASM("%5") ASM("%6") ASM("%7") \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)  \
                         : "v"(b), "v"((u32)b)                                                            \
                         : "vcc", "s10", "s11");                                                                      \
        }                                                                                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);      \
    }

#define A_ADD(R) "v_add_u32 " R ", " R ", %8\n"
#define A_MULLO(R) "v_mul_lo_u32 " R ", " R ", %8\n"
#define A_MUL24(R) "v_mul_u32_u24 " R ", " R ", %8\n"
#define A_MAD24(R) "v_mad_u32_u24 " R ", " R ", %8, " R "\n"
#define A_ALIGN(R) "v_alignbit_b32 " R ", " R ", %8, 7\n"
#define A_BFE(R) "v_bfe_u32 " R ", " R ", 3, 29\n"
#define A_LSHLADD(R) "v_lshl_add_u32 " R ", " R ", 3, %8\n"
#define A_CNDMASK(R) "v_cndmask_b32 " R ", " R ", %8, vcc\n"
#define A_CNDMASK64(R) "v_cndmask_b32_e64 " R ", " R ", %8, s[10:11]\n"
#define A_CNDMASKC(R) "v_cndmask_b32_e64 " R ", 0, 1, s[10:11]\n"
#define A_ADDS(R) "v_add_u32 " R ", s10, " R "\n"
#define A_AND(R) "v_and_b32 " R ", " R ", %8\n"
#define A_OR3(R) "v_or3_b32 " R ", " R ", %8, " R "\n"
#define A_ADD3(R) "v_add3_u32 " R ", " R ", %8, " R "\n"
#define A_LSHL(R) "v_lshlrev_b32 " R ", 3, " R "\n"
#define A_LSHR(R) "v_lshrrev_b32 " R ", 3, " R "\n"
#define A_SUB(R) "v_sub_u32 " R ", " R ", %8\n"
#define A_MIN(R) "v_min_u32 " R ", " R ", %8\n"
#define A_CMP32(R) "v_cmp_lt_u32 vcc, " R ", %8\n"
#define A_CMP32S(R) "v_cmp_lt_u32_e64 s[10:11], " R ", %8\n"
#define A_BCNT(R) "v_bcnt_u32_b32 " R ", " R ", %8\n"
#define A_BFREV(R) "v_bfrev_b32 " R ", " R "\n"
#define A_MOV(R) "v_mov_b32 " R ", %8\n"
#define A_READLANE(R) "v_readlane_b32 s10, " R ", 3\n"
#define A_DPP(R) "v_min_u32_dpp " R ", " R ", " R " row_ror:4 row_mask:0xf bank_mask:0xf\n"
#define A_SDWA(R) "v_lshlrev_b32_sdwa " R ", %8, " R " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define A_BITOP3(R) "v_bitop3_b32 " R ", " R ", %8, " R " bitop3:0x6c\n"
#define A_XOR(R) "v_xor_b32 " R ", " R ", %8\n"
#define A_MAD64(R) "v_mad_u64_u32 " R ", vcc, %9, 21, " R "\n"
#define A_LSHLADD64(R) "v_lshl_add_u64 " R ", " R ", 3, %8\n"
#define A_ADD64(R) "v_lshl_add_u64 " R ", " R ", 0, %8\n"
#define A_LSHL64(R) "v_lshlrev_b64 " R ", 3, " R "\n"
#define A_LSHR64(R) "v_lshrrev_b64 " R ", 3, " R "\n"
#define A_ADDF64(R) "v_add_f64 " R ", " R ", %8\n"
#define A_FMAF64(R) "v_fma_f64 " R ", " R ", %8, " R "\n"
#define A_MULF64(R) "v_mul_f64 " R ", " R ", %8\n"
#define A_CMP64(R) "v_cmp_lt_u64 vcc, " R ", %8\n"
#define A_MOV64(R) "v_mov_b64 " R ", %8\n"
#define A_PKADD(R) "v_pk_add_u16 " R ", " R ", %8\n"

KERNEL32(k_add, A_ADD)
KERNEL32(k_mullo, A_MULLO)
KERNEL32(k_mul24, A_MUL24)
KERNEL32(k_mad24, A_MAD24)
KERNEL32(k_align, A_ALIGN)
KERNEL32(k_bfe, A_BFE)
KERNEL32(k_lshladd, A_LSHLADD)
KERNEL32(k_cndmask, A_CNDMASK)
KERNEL32(k_sdwa, A_SDWA)
KERNEL32(k_cndmask64, A_CNDMASK64)
KERNEL32(k_cndmaskc, A_CNDMASKC)
KERNEL32(k_adds, A_ADDS)
KERNEL32(k_and, A_AND)
KERNEL32(k_or3, A_OR3)
KERNEL32(k_add3, A_ADD3)
KERNEL32(k_lshl, A_LSHL)
KERNEL32(k_lshr, A_LSHR)
KERNEL32(k_sub, A_SUB)
KERNEL32(k_min, A_MIN)
KERNEL32(k_cmp32, A_CMP32)
KERNEL32(k_cmp32s, A_CMP32S)
KERNEL32(k_bcnt, A_BCNT)
KERNEL32(k_bfrev, A_BFREV)
KERNEL32(k_mov, A_MOV)
KERNEL32(k_readlane, A_READLANE)
KERNEL32(k_dpp, A_DPP)
KERNEL32(k_bitop3, A_BITOP3)
KERNEL32(k_xor, A_XOR)
KERNEL32(k_pkadd, A_PKADD)
KERNEL64(k_mad64, A_MAD64)
KERNEL64(k_lshladd64, A_LSHLADD64)
KERNEL64(k_add64, A_ADD64)
KERNEL64(k_lshl64, A_LSHL64)
KERNEL64(k_lshr64, A_LSHR64)
KERNEL64(k_addf64, A_ADDF64)
KERNEL64(k_fmaf64, A_FMAF64)
KERNEL64(k_mulf64, A_MULF64)
KERNEL64(k_cmp64, A_CMP64)
KERNEL64(k_mov64, A_MOV64)

// random 8-byte LDS look-ups as the scan's class tables see them: 6 tables of 256 u64, independent random indices per lane
__global__ void __launch_bounds__(256) k_ldsrand(u32* out, u32 seed) {
    __shared__ u64 tab[6 * 256];
    for (u32 i = threadIdx.x; i < 6 * 256; i += 256) tab[i] = i * 0x9E3779B97F4A7C15ull;
    __syncthreads();
    u32 x = threadIdx.x * 2654435761u + seed;
    u64 acc = 0;
    for (int i = 0; i < N_ITER; i++) {
#pragma unroll
        for (int c = 0; c < 16; c++) {
            acc += tab[(c % 6) * 256 + ((x >> (c & 15)) & 255)];
        }
        x = x * 1664525u + 1013904223u + (u32)acc;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)acc;
}
// the same look-ups from 4-byte tables
__global__ void __launch_bounds__(256) k_ldsrand32(u32* out, u32 seed) {
    __shared__ u32 tab[6 * 256];
    for (u32 i = threadIdx.x; i < 6 * 256; i += 256) tab[i] = i * 0x9E3779B9u;
    __syncthreads();
    u32 x = threadIdx.x * 2654435761u + seed;
    u32 acc = 0;
    for (int i = 0; i < N_ITER; i++) {
#pragma unroll
        for (int c = 0; c < 16; c++) {
            acc += tab[(c % 6) * 256 + ((x >> (c & 15)) & 255)];
        }
        x = x * 1664525u + 1013904223u + acc;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

typedef void (*kern_t)(u32*, u32);
static double run(kern_t k, u32* d_out, int blocks) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, 2u);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const int blocks = cus * 4;  // 4 blocks of 4 waves per CU: 4 waves per SIMD
    u32* d_out;
    CHECK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
    printf("%s: %d CUs, clock %d kHz, %d blocks of 256\n", p.gcnArchName, cus, p.clockRate, blocks);
    struct { const char* name; kern_t k; } ks[] = {
        {"v_add_u32", k_add}, {"v_xor_b32", k_xor}, {"v_mul_lo_u32", k_mullo}, {"v_mul_u32_u24", k_mul24}, {"v_mad_u32_u24", k_mad24},
        {"v_alignbit_b32", k_align}, {"v_bfe_u32", k_bfe}, {"v_lshl_add_u32", k_lshladd}, {"v_cndmask_b32", k_cndmask},
        {"v_lshlrev_b32_sdwa", k_sdwa}, {"v_cndmask_b32_e64 sgpr mask", k_cndmask64}, {"v_cndmask_b32_e64 0,1", k_cndmaskc}, {"v_add_u32 sgpr src", k_adds},
        {"v_and_b32", k_and}, {"v_or3_b32", k_or3}, {"v_add3_u32", k_add3}, {"v_lshlrev_b32", k_lshl}, {"v_lshrrev_b32", k_lshr}, {"v_sub_u32", k_sub}, {"v_min_u32", k_min},
        {"v_cmp_lt_u32 vcc", k_cmp32}, {"v_cmp_lt_u32_e64 sgpr", k_cmp32s}, {"v_bcnt_u32_b32", k_bcnt}, {"v_bfrev_b32", k_bfrev}, {"v_mov_b32", k_mov}, {"v_readlane_b32", k_readlane},
        {"v_min_u32_dpp row_ror", k_dpp}, {"v_bitop3_b32", k_bitop3}, {"v_pk_add_u16", k_pkadd}, {"v_mad_u64_u32", k_mad64}, {"v_lshl_add_u64 <<3", k_lshladd64},
        {"v_lshl_add_u64 <<0", k_add64}, {"v_lshlrev_b64", k_lshl64}, {"v_lshrrev_b64", k_lshr64}, {"v_add_f64", k_addf64},
        {"v_fma_f64", k_fmaf64}, {"v_mul_f64", k_mulf64}, {"v_cmp_lt_u64", k_cmp64}, {"v_mov_b64", k_mov64},
        {"ds_read_b64 random (+addr)", k_ldsrand}, {"ds_read_b32 random (+addr)", k_ldsrand32},
    };
    double base = 0;
    for (auto& e : ks) {
        double best = 1e30;
        for (int r = 0; r < 3; r++) best = std::min(best, run(e.k, d_out, blocks));
        if (!base) base = best;
        // cycles per wave-instruction at the reported clock: 4 waves per SIMD share it
        const double insts = (double)N_ITER * 16;
        const double cyc = best * 1e-3 * (double)p.clockRate * 1e3 / (insts * 4);
        printf("%-28s %8.3f ms  %6.2f cycles/wave-inst  x%.2f of v_add_u32\n", e.name, best, cyc, best / base);
    }
    return 0;
}
