// Micro-benchmark: what a SIMD of gfx950 overlaps with its matrix instructions.
//
// The conv kernel's ablations show that the staging transform, the epilogue and the MFMAs each cost
// their full time although two waves share every SIMD.  This program measures the issue model directly:
//   same-wave   one wave per SIMD runs [4 MFMAs + V other instructions] per iteration, V = 0..64
//   two-wave    two waves per SIMD, one runs only MFMAs, the other only the other instruction kind
// for fp32 (v_mfma_f32_32x32x2_f32) and bf16 (v_mfma_f32_32x32x16_bf16) MFMAs against v_fma_f32, v_pk_fma_f32,
// ds_read_b32 and v_sin_f32.  "overlap" means time = max(parts); "exclusive" means time = sum(parts).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_coissue.hip -o gpurun_out/ubench_coissue   (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { K_NONE = 0, K_FMA = 1, K_PKFMA = 2, K_LDS = 3, K_SIN = 4, K_MOV = 5, K_GLD = 6, K_SALU = 7, K_LDSW = 8 };
static const float* g_host_src = nullptr;

template <int KIND, int V> __device__ __forceinline__ void others(float (&c)[8], float a, float b, const float* lds, const float* g_src) {
#pragma unroll
    for (int i = 0; i < V; ++i) {
        if constexpr (KIND == K_FMA) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c[i & 7]) : "v"(a), "v"(b));
        if constexpr (KIND == K_PKFMA) {
            floatx2& r = *reinterpret_cast<floatx2*>(&c[(i & 3) * 2]);
            floatx2 aa = {a, a}, bb = {b, b};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r) : "v"(aa), "v"(bb));
        }
        if constexpr (KIND == K_LDS) asm volatile("ds_read_b32 %0, %1" : "+v"(c[i & 7]) : "v"((unsigned)(size_t)lds + 256u * (i & 7)));     // destination stays live (it is c[])
        if constexpr (KIND == K_GLD) asm volatile("global_load_dword %0, %1, %2" : "+v"(c[i & 7]) : "v"(4u * (threadIdx.x & 63) + 256u * (i & 7)), "s"(g_src));
        if constexpr (KIND == K_SALU) { int t; asm volatile("s_add_u32 %0, %1, 1" : "=s"(t) : "s"(i) : "scc"); asm volatile("" :: "s"(t)); }
        if constexpr (KIND == K_LDSW) asm volatile("ds_write_b32 %0, %1" :: "v"((unsigned)(size_t)lds + 256u * (i & 7)), "v"(a));
        if constexpr (KIND == K_SIN) asm volatile("v_sin_f32 %0, %0" : "+v"(c[i & 7]));
        if constexpr (KIND == K_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(c[i & 7]) : "v"(a));
    }
    if constexpr (KIND == K_LDS || KIND == K_LDSW) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (KIND == K_GLD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// MODE 0: same wave: per iteration NM MFMAs, with V others spread behind each MFMA (V / NM each)
// MODE 1: two waves per SIMD (block of 512): waves 0-3 only MFMAs, waves 4-7 only others (V per iteration)
// MODE 2: block of 512, all eight waves run [MFMAs + others] (the conv kernel's situation)
template <bool BF16, int KIND, int V, int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, int* simd_ids, const float* gsrc) {
    __shared__ float lds[64 * 16];
    lds[threadIdx.x & 1023] = 1.f;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    floatx16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = 0.001f * (threadIdx.x + i);
    const float a = 1.0001f, b = 1e-6f;
    bf16x8 ab;
#pragma unroll
    for (int i = 0; i < 8; ++i) ab[i] = (__bf16)1.0f;
    if (simd_ids && (threadIdx.x & 63) == 0 && blockIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        simd_ids[wave] = (int)id;
    }
    const bool do_m = MODE != 1 || wave < 4, do_o = MODE != 1 || wave >= 4;
    constexpr int NM = 4, PER = V / NM;
    if (MODE == 1 && !do_m) {
        for (int it = 0; it < iters; ++it) others<KIND, V>(c, a, b, lds + (threadIdx.x & 63), gsrc);
    } else if (MODE == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                if constexpr (BF16) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc[m], 0, 0, 0);
                else acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                if constexpr (BF16) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc[m], 0, 0, 0);
                else acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                others<KIND, PER>(c, a, b, lds + (threadIdx.x & 63), gsrc);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += c[i];
    if (s == 123.456f) out[threadIdx.x] = s;      // never true: keeps everything live
    (void)do_o;
}

template <bool BF16, int KIND, int V, int MODE>
static double run(int threads, int iters, float* out, int* ids = nullptr) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<BF16, KIND, V, MODE>), dim3(256), dim3(threads), 0, 0, out, 1000, ids, g_host_src);      // warm-up (clock ramp)
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<BF16, KIND, V, MODE>), dim3(256), dim3(threads), 0, 0, out, iters, ids, g_host_src);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return ms * 1e6 / iters;                       // ns per iteration
}

static const char* kname(int k) { return k == K_FMA ? "v_fma_f32" : k == K_PKFMA ? "v_pk_fma_f32" : k == K_LDS ? "ds_read_b32" : k == K_SIN ? "v_sin_f32" : k == K_MOV ? "v_mov_b32" : k == K_GLD ? "global_load" : k == K_SALU ? "s_add_u32" : k == K_LDSW ? "ds_write_b32" : "-"; }

template <bool BF16, int KIND>
static void sweep(float* out, int iters) {
    const char* mm = BF16 ? "bf16 32x32x16" : "fp32 32x32x2 ";
    const double m0 = run<BF16, K_NONE, 0, 0>(256, iters, out);
    printf("%s  4 MFMAs alone (1 wave/SIMD): %7.1f ns/iter\n", mm, m0);
    printf("%s + %-12s same wave, V per 4 MFMAs:", mm, kname(KIND));
    printf("  V=4 %6.1f", run<BF16, KIND, 4, 0>(256, iters, out));
    printf("  V=8 %6.1f", run<BF16, KIND, 8, 0>(256, iters, out));
    printf("  V=16 %6.1f", run<BF16, KIND, 16, 0>(256, iters, out));
    printf("  V=32 %6.1f", run<BF16, KIND, 32, 0>(256, iters, out));
    printf("  V=64 %6.1f", run<BF16, KIND, 64, 0>(256, iters, out));
    printf(" ns/iter\n");
    printf("%s | %-12s two waves/SIMD (one MFMA-only, one other-only), V per iter:", mm, kname(KIND));
    printf("  V=8 %6.1f", run<BF16, KIND, 8, 1>(512, iters, out));
    printf("  V=16 %6.1f", run<BF16, KIND, 16, 1>(512, iters, out));
    printf("  V=32 %6.1f", run<BF16, KIND, 32, 1>(512, iters, out));
    printf("  V=64 %6.1f", run<BF16, KIND, 64, 1>(512, iters, out));
    printf(" ns/iter\n");
    printf("%s + %-12s two waves/SIMD, both run MFMAs + V:", mm, kname(KIND));
    printf("  V=0 %6.1f", run<BF16, K_NONE, 0, 2>(512, iters, out));
    printf("  V=8 %6.1f", run<BF16, KIND, 8, 2>(512, iters, out));
    printf("  V=16 %6.1f", run<BF16, KIND, 16, 2>(512, iters, out));
    printf("  V=32 %6.1f", run<BF16, KIND, 32, 2>(512, iters, out));
    printf(" ns/iter (per wave pair: 8 MFMAs + 2V)\n");
}

// the other instruction kind alone, one wave per SIMD (no MFMAs): ns per instruction
template <int KIND, int V> __global__ __launch_bounds__(256) void k_alone(float* out, int iters, const float* gsrc) {
    __shared__ float lds[64 * 16];
    lds[threadIdx.x] = 1.f;
    __syncthreads();
    float c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = 0.001f * (threadIdx.x + i);
    for (int it = 0; it < iters; ++it) others<KIND, V>(c, 1.0001f, 1e-6f, lds + (threadIdx.x & 63), gsrc);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += c[i];
    if (s == 123.456f) out[threadIdx.x] = s;
}
template <int KIND> static void alone(float* out, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_alone<KIND, 32>), dim3(256), dim3(256), 0, 0, out, 1000, g_host_src);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_alone<KIND, 32>), dim3(256), dim3(256), 0, 0, out, iters, g_host_src);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-12s alone, 1 wave/SIMD: %6.2f ns per instruction\n", kname(KIND), ms * 1e6 / iters / 32);
}

int main() {
    float* out;
    int* ids;
    CHECK(hipMalloc(&out, 4096));
    CHECK(hipMalloc(&ids, 64));
    float* src;
    CHECK(hipMalloc(&src, 1 << 16));
    CHECK(hipMemset(src, 0, 1 << 16));
    g_host_src = src;
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const int iters = 200000;
    run<false, K_NONE, 0, 2>(512, 1000, out, ids);
    int h[8];
    CHECK(hipMemcpy(h, ids, 32, hipMemcpyDeviceToHost));
    printf("HW_ID of the 8 waves of block 0 (simd = bits 5:4):");
    for (int i = 0; i < 8; ++i) printf(" w%d:simd%d", i, (h[i] >> 4) & 3);
    printf("\n");
    alone<K_FMA>(out, iters);
    alone<K_PKFMA>(out, iters);
    alone<K_SIN>(out, iters);
    alone<K_MOV>(out, iters);
    alone<K_LDS>(out, iters);
    alone<K_LDSW>(out, iters);
    alone<K_GLD>(out, iters);
    sweep<false, K_FMA>(out, iters);
    sweep<false, K_PKFMA>(out, iters);
    sweep<false, K_SIN>(out, iters);
    sweep<false, K_LDS>(out, iters);
    sweep<false, K_LDSW>(out, iters);
    sweep<false, K_GLD>(out, iters);
    sweep<false, K_SALU>(out, iters);
    sweep<true, K_FMA>(out, iters);
    sweep<true, K_PKFMA>(out, iters);
    sweep<true, K_SIN>(out, iters);
    sweep<true, K_LDS>(out, iters);
    sweep<true, K_GLD>(out, iters);
    CHECK(hipFree(out));
    CHECK(hipFree(ids));
    return 0;
}
