// valu_rate.hip -- issue rate of the VALU instruction classes the hope kernels are made of, measured on gfx950.
//
// Why: every kernel of the step is VALU-issue / latency bound, so its roofline is "wave-instructions per second".  That
// ceiling depends on the instruction: float64 add / mul / fma issue a wave64 in 4 cycles on a SIMD, 32-bit ALU ops in 2
// (MI355X_MICROARCH.md "Per-instruction cycle constants"), transcendentals and DPP / readlane traffic differently again.
// This program measures the rate of each class directly instead of assuming one number for all of them.
//
// Method: one kernel per class; every wave runs `iters` trips over 16 independent chains of the instruction (inline asm, so
// the compiler cannot fold or re-schedule it), 256-thread workgroups (one wave per SIMD), `wps` workgroups per CU, i.e.
// `wps` waves per SIMD.  Rate = instructions issued / kernel time (HIP events, best of 5); cycles per instruction =
// SIMD-seconds x measured shader clock / instructions, the clock from s_memtime over the kernel's wall time.
//
// Build + run: tools/valu_rate.py (hipcc --offload-arch=gfx950).  Output: one JSON object on stdout.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int CHAINS = 16;

#define REP16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

// ---- one kernel per instruction class -------------------------------------------------------------------------------------
// each body: 16 independent accumulators x[0..15]; `ticks` = s_memtime delta of wave 0 of every block (clock estimate)
#define KERNEL_D(NAME, ASM)                                                                          \
    __global__ __launch_bounds__(256) void NAME(int iters, double seed, double* out, unsigned long long* ticks) { \
        double x[CHAINS];                                                                            \
        _Pragma("unroll") for (int i = 0; i < CHAINS; i++) x[i] = seed + 1e-3 * (threadIdx.x + 64 * i);          \
        const double a = 1.0000001, b = 1e-9;                                                        \
        const unsigned long long t0 = __builtin_readcyclecounter();                                  \
        for (int it = 0; it < iters; it++) {                                                         \
            _Pragma("unroll") for (int i = 0; i < CHAINS; i++) asm volatile(ASM : "+v"(x[i]) : "v"(a), "v"(b) : "vcc");  \
        }                                                                                            \
        const unsigned long long t1 = __builtin_readcyclecounter();                                  \
        double s = 0;                                                                                \
        _Pragma("unroll") for (int i = 0; i < CHAINS; i++) s += x[i];                                \
        if (s == 123.456) out[0] = s;                                                                \
        if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;                                           \
    }
#define KERNEL_F(NAME, ASM)                                                                          \
    __global__ __launch_bounds__(256) void NAME(int iters, double seed, double* out, unsigned long long* ticks) { \
        float x[CHAINS];                                                                             \
        _Pragma("unroll") for (int i = 0; i < CHAINS; i++) x[i] = (float)seed + 1e-3f * (threadIdx.x + 64 * i);  \
        const float a = 1.0000001f, b = 1e-9f;                                                       \
        const unsigned long long t0 = __builtin_readcyclecounter();                                  \
        for (int it = 0; it < iters; it++) {                                                         \
            _Pragma("unroll") for (int i = 0; i < CHAINS; i++) asm volatile(ASM : "+v"(x[i]) : "v"(a), "v"(b) : "vcc");  \
        }                                                                                            \
        const unsigned long long t1 = __builtin_readcyclecounter();                                  \
        float s = 0;                                                                                 \
        _Pragma("unroll") for (int i = 0; i < CHAINS; i++) s += x[i];                                \
        if (s == 123.456f) out[0] = s;                                                               \
        if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;                                           \
    }
#define KERNEL_I(NAME, ASM)                                                                          \
    __global__ __launch_bounds__(256) void NAME(int iters, double seed, double* out, unsigned long long* ticks) { \
        int x[CHAINS];                                                                               \
        _Pragma("unroll") for (int i = 0; i < CHAINS; i++) x[i] = (int)seed + threadIdx.x + 64 * i;  \
        const int a = 3, b = 0x55aa;                                                                 \
        const unsigned long long t0 = __builtin_readcyclecounter();                                  \
        for (int it = 0; it < iters; it++) {                                                         \
            _Pragma("unroll") for (int i = 0; i < CHAINS; i++) asm volatile(ASM : "+v"(x[i]) : "v"(a), "v"(b) : "vcc");  \
        }                                                                                            \
        const unsigned long long t1 = __builtin_readcyclecounter();                                  \
        int s = 0;                                                                                   \
        _Pragma("unroll") for (int i = 0; i < CHAINS; i++) s += x[i];                                \
        if (s == 123456789) out[0] = s;                                                              \
        if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;                                           \
    }

KERNEL_D(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
KERNEL_D(k_add_f64, "v_add_f64 %0, %0, %2")
KERNEL_D(k_mul_f64, "v_mul_f64 %0, %0, %1")
KERNEL_D(k_min_f64, "v_min_f64 %0, %0, %1")
KERNEL_D(k_rcp_f64, "v_rcp_f64 %0, %0")
KERNEL_D(k_sqrt_f64, "v_sqrt_f64 %0, %0")
// a compare + select on the low word: the pattern of every fmin / clip / branch-free select in the kernels (2 instructions)
__global__ __launch_bounds__(256) void k_cmp_cndmask_f64(int iters, double seed, double* out, unsigned long long* ticks) {
    int y[CHAINS];
    double x[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; i++) { x[i] = seed + 1e-3 * (threadIdx.x + 64 * i); y[i] = threadIdx.x + i; }
    const double a = 1.0000001;
    const int b = 77;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++)
            asm volatile("v_cmp_lt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(y[i]) : "v"(x[i]), "v"(a), "v"(b) : "vcc");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    int s = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; i++) s += y[i];
    if (s == 123456789) out[0] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
KERNEL_F(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL_F(k_add_f32, "v_add_f32 %0, %0, %2")
KERNEL_F(k_min_f32, "v_min_f32 %0, %0, %1")
KERNEL_F(k_rcp_f32, "v_rcp_f32 %0, %0")
KERNEL_F(k_cmp_cndmask_f32, "v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" )
KERNEL_I(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL_I(k_and_b32, "v_and_b32 %0, %0, %2")
KERNEL_I(k_lshl_b32, "v_lshlrev_b32 %0, 1, %0")
KERNEL_I(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL_I(k_mov_dpp, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL_I(k_mov_b32, "v_mov_b32 %0, %1")
KERNEL_I(k_cvt_f64_i32, "v_cvt_f32_i32 %0, %0")

// v_readlane_b32 writes an SGPR: keep the result alive through a scalar add
__global__ __launch_bounds__(256) void k_readlane(int iters, double seed, double* out, unsigned long long* ticks) {
    int x[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; i++) x[i] = (int)seed + threadIdx.x + 64 * i;
    int acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
            int s;
            asm volatile("v_readlane_b32 %0, %1, 7" : "=s"(s) : "v"(x[i]));
            acc ^= s;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (acc == 123456789) out[0] = acc;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

// LDS: ds_read_b64 per lane, conflict-free (lane-consecutive), the dominant LDS access of the kernels
__global__ __launch_bounds__(256) void k_ds_read_b64(int iters, double seed, double* out, unsigned long long* ticks) {
    __shared__ double buf[256 * 8];
    for (int i = threadIdx.x; i < 256 * 8; i += 256) buf[i] = seed + i;
    __syncthreads();
    double s = 0;
    const int addr = (int)(threadIdx.x * 8);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        double v[CHAINS];
#define DSR(i) asm volatile("ds_read_b64 %0, %1 offset:" #i "*512" : "=v"(v[i]) : "v"(addr));
        REP16(DSR)
#undef DSR
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < CHAINS; i++) asm volatile("" :: "v"(v[i]));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (s == 123.456) out[0] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

// ---- per-wave issue cadence: scalar ALU, VALU + SALU interleaved, and one DEPENDENT chain --------------------------------
__global__ __launch_bounds__(256) void k_salu(int iters, double seed, double* out, unsigned long long* ticks) {
    int x[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; i++) x[i] = __builtin_amdgcn_readfirstlane((int)seed + i);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) asm volatile("s_add_u32 %0, %0, 3" : "+s"(x[i]) : : "scc");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    int s = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; i++) s += x[i];
    if (s == 123456789) out[0] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
// 16 x (one VALU + one SALU): if a wave could issue them together the pair would cost one VALU slot
__global__ __launch_bounds__(256) void k_valu_salu(int iters, double seed, double* out, unsigned long long* ticks) {
    int x[CHAINS], y[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; i++) { x[i] = (int)seed + threadIdx.x + i; y[i] = __builtin_amdgcn_readfirstlane((int)seed + i); }
    const int a = 3;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) asm volatile("v_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, 3" : "+v"(x[i]), "+s"(y[i]) : "v"(a) : "scc");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    int s = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; i++) s += x[i] + y[i];
    if (s == 123456789) out[0] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
__global__ __launch_bounds__(256) void k_dep_fma_f64(int iters, double seed, double* out, unsigned long long* ticks) {
    double x = seed + 1e-3 * threadIdx.x;
    const double a = 1.0000001, b = 1e-9;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (x == 123.456) out[0] = x;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
__global__ __launch_bounds__(256) void k_dep_fma_f32(int iters, double seed, double* out, unsigned long long* ticks) {
    float x = (float)seed + 1e-3f * threadIdx.x;
    const float a = 1.0000001f, b = 1e-9f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (x == 123.456f) out[0] = x;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
typedef void (*kern_t)(int, double, double*, unsigned long long*);
struct Case { const char* name; kern_t fn; int insts_per_chain_op; const char* cls; };

int main(int argc, char** argv) {
    int iters = 8192;
    if (argc > 1) iters = atoi(argv[1]);
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double* out;
    unsigned long long* ticks;
    CHK(hipMalloc(&out, 64));
    CHK(hipMalloc(&ticks, sizeof(unsigned long long) * cus * 8));
    std::vector<unsigned long long> hticks(cus * 8);
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const Case cases[] = {
        {"v_fma_f64", k_fma_f64, 1, "f64"}, {"v_add_f64", k_add_f64, 1, "f64"}, {"v_mul_f64", k_mul_f64, 1, "f64"},
        {"v_min_f64", k_min_f64, 1, "f64_minmax"}, {"v_rcp_f64", k_rcp_f64, 1, "trans_f64"}, {"v_sqrt_f64", k_sqrt_f64, 1, "trans_f64"},
        {"v_cmp_lt_f64+v_cndmask_b32", k_cmp_cndmask_f64, 2, "cmp_f64"},
        {"v_fma_f32", k_fma_f32, 1, "b32"}, {"v_add_f32", k_add_f32, 1, "b32"}, {"v_min_f32", k_min_f32, 1, "b32"},
        {"v_rcp_f32", k_rcp_f32, 1, "trans_f32"}, {"v_cmp_lt_f32+v_cndmask_b32", k_cmp_cndmask_f32, 2, "b32"},
        {"v_add_u32", k_add_u32, 1, "b32"}, {"v_and_b32", k_and_b32, 1, "b32"}, {"v_lshlrev_b32", k_lshl_b32, 1, "b32"},
        {"v_mul_lo_u32", k_mul_lo_u32, 1, "mul_i32"}, {"v_mov_b32_dpp", k_mov_dpp, 1, "dpp"}, {"v_mov_b32", k_mov_b32, 1, "b32"},
        {"v_cvt_f32_i32", k_cvt_f64_i32, 1, "b32"}, {"v_readlane_b32", k_readlane, 1, "readlane"},
        {"ds_read_b64", k_ds_read_b64, 1, "lds"},
        {"s_add_u32", k_salu, 1, "salu"}, {"v_add_u32+s_add_u32", k_valu_salu, 2, "valu_salu_pair"},
        {"v_fma_f64 (one dependent chain)", k_dep_fma_f64, 1, "dep_f64"}, {"v_fma_f32 (one dependent chain)", k_dep_fma_f32, 1, "dep_f32"},
    };
    printf("{\"device\": \"%s\", \"cus\": %d, \"iters\": %d, \"chains\": %d, \"results\": [", prop.gcnArchName, cus, iters, CHAINS);
    bool first = true;
    for (const Case& c : cases) {
        for (int wps : {1, 2, 4, 8}) {
            const int grid = cus * wps;
            double best_ms = 1e30;
            double clk_ghz = 0;
            for (int rep = 0; rep < 6; rep++) {
                CHK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(c.fn, dim3(grid), dim3(256), 0, 0, iters, 1.0 + rep, out, ticks);
                CHK(hipEventRecord(e1, 0));
                CHK(hipEventSynchronize(e1));
                float ms = 0;
                CHK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best_ms) {
                    best_ms = ms;
                    CHK(hipMemcpy(hticks.data(), ticks, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost));
                    // the longest-running block spans (almost) the whole kernel: ticks / wall time = counter frequency
                    unsigned long long mx = 0;
                    for (int i = 0; i < grid; i++) mx = std::max(mx, hticks[i]);
                    clk_ghz = (double)mx / (ms * 1e-3) / 1e9;
                }
            }
            const double wave_insts = (double)grid * 4 * (double)iters * CHAINS * c.insts_per_chain_op;   // 4 waves per workgroup
            const double rate = wave_insts / (best_ms * 1e-3);
            // per-SIMD issue interval in ns and in cycles of a 2.4 GHz clock (the data-sheet maximum)
            const double simd_ns = best_ms * 1e6 / ((double)wps * iters * CHAINS * c.insts_per_chain_op);
            printf("%s\n  {\"inst\": \"%s\", \"class\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"wave_insts_per_s\": %.4e, "
                   "\"ns_per_inst_per_simd\": %.4f, \"cycles_at_2p4GHz\": %.3f, \"memtime_GHz\": %.4f}",
                   first ? "" : ",", c.name, c.cls, wps, best_ms, rate, simd_ns, simd_ns * 2.4, clk_ghz);
            first = false;
        }
    }
    printf("\n]}\n");
    return 0;
}
