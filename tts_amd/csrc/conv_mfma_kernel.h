// Implicit-GEMM Conv1d for gfx950 on the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32.
//
// This one kernel carries ~99 % of the model's FLOPs (SURVEY.md 8a rows a5, a12, a16, a17 and
// every Linear, which is a k=1 conv in the channel-first layout):
//     y[b][co][t] = sum_{tap,ci} w[tap][ci][co] * f(x[b][ci][t + tap*dil - pad]) + bias[co]
// GEMM view per sentence:  M = co (MFMA row i), N = t (MFMA column j = lane&31, so one
// accumulator register of a half-wave is 32 consecutive t of one output row: 128-B coalesced
// stores), K = (tap, ci).  No im2col: the x tile [KC ci][BN + halo] is staged ONCE per ci-tile
// into LDS with the producer's AdaIN affine and Snake/LeakyReLU already applied (so the
// normalised tensor is never written to HBM), and every tap re-reads it at a shifted column.
//
// Persistent workgroups (2 per CU) walk the output tiles, so the prologue of a tile (first x tile,
// first weights) and the store tail of the previous one overlap the MFMAs: with K = cin*taps as small
// as 384 every output tile is only 8 ci-tiles long and a one-tile-per-block launch spends half its
// time in un-overlapped prologues and epilogues.
// Pipeline: the x tile is double-buffered in LDS -- the raw global loads of ci-tile i+1 are issued
// into registers at the first tap of tile i and transformed + written to the other buffer right
// after that tap's MFMAs are issued; one barrier per ci-tile, or (two-set loop, fp32 / bf16x3) a third buffer and a
// step counter instead of it (LDS3 below).  The A operand (weights) goes
// global->VGPR directly (lane l reads w[tap][ci0+2kk+(l>>5)][m+(l&31)], two 128-B segments,
// L2-resident) and is prefetched one whole tap (16 registers) ahead of the MFMAs that consume it,
// across ci-tile boundaries too.  B operand: ds_read_b32 of
// xs[2kk+(l>>5)][n+(l&31)+tap*dil], 32 consecutive floats per half-wave = conflict-free.
// A wave owns a 64(co) x 64(t) tile = 2x2 MFMA tiles = 64 accumulator VGPRs; WM x WN waves per block.
//
// This header holds the kernel template and the launch helpers that pick its compile-time epilogue; the instantiations are
// split over two translation units so that they compile in parallel: conv_mfma.hip (fp32, bf16x3, the generic bf16 loop,
// launch_conv) and conv_ring.hip (the bf16 ring pipeline).
#pragma once
#include "kernels.h"
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <utility>

#ifndef K82_OCC_BF16
#define K82_OCC_BF16 3              // workgroups per CU (= waves per SIMD) the bf16-mode kernels are built for
#endif
#ifndef K82_OCC_RING
#define K82_OCC_RING 2
#endif
#ifndef K82_SMALL_GRID
#define K82_SMALL_GRID 512          // fp32 launches whose 128 x 128 tiles would not fill the persistent grid use 64 x 64 tiles
#endif
#ifndef K82_EPI_FAST
#define K82_EPI_FAST 1              // lean epilogue for interior tiles (0: the general path everywhere)
#endif
#ifndef K82_OCC_F32
#define K82_OCC_F32 2               // workgroups per CU of the fp32 kernels
#endif
#ifndef K82_TRACE
#define K82_TRACE 0                 // 1: phase time stamps (tools/trace_conv.py); never in the shipped library
#endif
#ifndef K82_RING_SMEM
#define K82_RING_SMEM 1             // ring pipeline: per-channel staging parameters through the scalar cache (0: vector loads, A/B only)
#endif
#ifndef K82_RING_WN
#define K82_RING_WN 2               // ring pipeline: 2 = large launches run 4 x 2 waves on 128 x 256 tiles (1: always 4 x 1)
#endif
#ifndef K82_F32_HWSIN
#define K82_F32_HWSIN 1             // fp32 Snake through v_sin_f32 (argument reduced with v_fract): +1.2 % on configs[1] once the waves were decoupled
                                    // (+0.6 % before), teacher-forced waveform error 2.03e-6 against 1.91e-6 with the polynomial (0: the polynomial)
#endif
#ifndef K82_LDS3
#define K82_LDS3 1                  // two-set loop: three x-tile buffers + a step counter instead of the per-ci-tile barrier (fp32 +0.5 %, bf16x3 +1.6 %)
#endif
#ifndef K82_F32_BM256
#define K82_F32_BM256 1             // fp32, >= 256 output channels, large launches: 256 x 128 workgroup tiles (+1.6 % on configs[1])
#endif
#ifndef K82_SPLIT_BM256
#define K82_SPLIT_BM256 1           // bf16x3, >= 256 output channels, large launches: 256 x 128 tiles, 8 x 1 waves (+2.5 % on configs[1])
#endif
#ifndef K82_RING_BM256
#define K82_RING_BM256 1            // bf16 ring, >= 256 output channels, large launches: 256 x 128 tiles, 8 x 1 waves (+1.5 % on configs[1])
#endif
#ifndef K82_N32_HTN
#define K82_N32_HTN 2               // MFMA tiles along t per wave in the 32-row fp32 shape (conv_post, the F0 / N heads): 2 = 256-column workgroup tiles
#endif
#ifndef K82_SMALL_OCC
#define K82_SMALL_OCC 2             // persistent workgroups per CU of the fp32 small-launch (64 x 64 tile) kernels
#endif
#ifndef K82_OCC_SPLIT
#define K82_OCC_SPLIT 2             // the bf16x3 kernels carry twice the A fragments: 3 per CU spills
#endif

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(<N - 1>) -- every index a constant expression inside f,
// whatever the optimiser's unroll thresholds (a loop it declines to unroll turns register arrays into scratch memory)
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

// (bf16(a), bf16(b)) packed into one dword, a in the low half: one v_cvt_pk_bf16_f32 (round-to-nearest-even)
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((floatx2){a, b}, bf16x2));
}

__device__ __forceinline__ float gelu_new(float x) {
    const float k = 0.7978845608028654f;   // sqrt(2/pi)
    return 0.5f * x * (1.0f + tanhf(k * (x + 0.044715f * x * x * x)));
}

// Snake1D: v + sin(alpha v)^2 / alpha.  sin^2 is even and pi-periodic, so only r = x - k*pi
// (two-constant Cody-Waite, k = rint(x/pi)) in [-pi/2, pi/2] is needed, no sign bookkeeping; the
// odd Taylor polynomial to r^11 is within 6e-8 of sin(r) there.  ~14 VALU instructions, no branches
// (ocml sinf costs ~60 and pulls a Payne-Hanek slow path into the staging loop).
__device__ __forceinline__ float snake(float v, float al, float ial) {
    const float x = al * v;
    const float k = rintf(x * 0.318309886183790672f);
    float r = fmaf(-k, 3.140625f, x);
    r = fmaf(-k, 9.67653589793e-4f, r);
    const float r2 = r * r;
    float pl = fmaf(r2, -2.50521083854e-8f, 2.75573192240e-6f);
    pl = fmaf(r2, pl, -1.98412698413e-4f);
    pl = fmaf(r2, pl, 8.33333333333e-3f);
    pl = fmaf(r2, pl, -1.66666666667e-1f);
    const float sn = fmaf(r * r2, pl, r);
    return fmaf(ial, sn * sn, v);
}

// EPI selects the epilogue at compile time (a load under a run-time flag, even a wave-uniform one,
// makes hipcc branch around it and drain vmcnt(0) per element -- the 16 us/tile the first versions
// of this kernel lost):
//   EPI_PLAIN  y = (f(acc + bias)) * out_scale                               float4 stores along t
//   EPI_RES    y = (f(acc + bias) + res) * out_scale                         + float4 residual loads
//   EPI_ACC    as EPI_RES, then accbuf = (acc_init ? 0 : accbuf) + y * acc_scale; y optional
//   EPI_SCALAR transposed-conv phase interleave (up_s) / x2 nearest residual (res_shift): dword path
//   EPI_PART   split-K: the raw sum of ONE K block goes to ConvP::part (no bias, no epilogue); splitk_reduce_kernel finishes
enum { EPI_PLAIN = 0, EPI_RES = 1, EPI_ACC = 2, EPI_SCALAR = 3, EPI_PART = 4 };
// sum over the 8 lanes of a row span (lane & ~7 ... | 7) on the VALU's DPP path (no LDS round trips): the two quad
// permutes (xor 1, xor 2), then row_half_mirror (lane i <-> 7 - i, the other quad of the eight)
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float sum8(float v) {
    v = dpp_add<0xB1>(v);      // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);      // quad_perm [2,3,0,1]
    return dpp_add<0x141>(v);  // row_half_mirror
}

// PREC = 1 (bf16): same staging, same epilogues, but the MFMAs are v_mfma_f32_32x32x16_bf16 (fp32 accumulate): the
// A operand is the pre-packed bf16 weight image (one 16-byte load per lane, row and k-half), the B operand is the
// staged x tile rounded to bf16 (round-to-nearest-even) when it is written to LDS.  One MFMA covers the whole
// 16-channel tile of a tap, so the matrix pipe needs 1/16 of the fp32 time and the kernel becomes a streaming
// (HBM / staging bound) kernel.
// bf16 mode only: sin through the hardware v_sin_f32 (argument in revolutions, ~1e-6 absolute error for the
// |alpha v| < 256 * 2 pi this model produces): 4 instructions instead of 14 -- in bf16 mode the staging VALU, not
// the matrix pipe, is what the Snake convolutions wait for.
// alrev = alpha / (2 pi), formed once per channel and ci-tile (one multiply per element less)
__device__ __forceinline__ float snake_fast(float v, float alrev, float ial) {
    const float sn = __builtin_amdgcn_sinf(alrev * v);
    return fmaf(ial, sn * sn, v);
}
constexpr float kInvTwoPi = 0.15915494309189535f;

//
// PREC = 2, "bf16x3": fp32-grade products from the bf16 matrix pipe.  Every operand is split into two bf16
// numbers, v = hi + lo + e with hi = rne_bf16(v), lo = rne_bf16(v - hi), |e| <= 2^-17 |v|, and
//     a * b  ~=  a_hi b_hi + a_hi b_lo + a_lo b_hi          (the dropped a_lo b_lo term is <= 2^-16 |a b|)
// is accumulated in fp32 by three v_mfma_f32_32x32x16_bf16 per 16-channel slice: 3/16 of the fp32 matrix time.
// The split of x happens ONCE, at staging.  In both bf16 modes the LDS tile is stored in MFMA-operand order: a dword
// holds the bf16 values of one column for the channel PAIR (2q, 2q+1), rows [q = 0..7][column] (plus the same 8 rows
// of lo parts in the split mode), so dword j of a lane's B fragment is one conflict-free ds_read_b32 -- no
// conversion or byte shuffling in the tap loop (VALU instructions there cost the SIMD ~10 cycles each that the
// partner wave's MFMAs do not cover).  A wave stages two channel pairs per ci-tile, packing with v_cvt_pk_bf16_f32.
// The weight image holds the (hi, lo) fragments of a lane side by side (32 bytes).
enum { PREC_F32 = 0, PREC_BF16 = 1, PREC_SPLIT = 2 };

//
// (Measured and not kept: a 512-thread workgroup per CU holding two wave groups that alternate, one barrier-separated
// slot out of step, between a compute slot (all taps of a ci-tile) and a stage slot (transform / split / LDS writes /
// epilogue), so that every SIMD always pairs a matrix-bound wave with a VALU-bound one.  Correct, but 5-10 % slower
// in the split mode and 20-35 % slower in the bf16 mode than two free-running 256-thread workgroups: a lone wave's
// tap step is bound by its own in-order issue -- weight loads, LDS reads, waits, 12 MFMAs -- not by the partner.)
// Two elements at a time on the packed-fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32: IEEE-identical to the
// scalar forms, half the issue slots -- every vector instruction of the staging step is one the partner wave's
// MFMAs have to share the SIMD's issue with).
__device__ __forceinline__ floatx2 fma2(floatx2 a, floatx2 b, floatx2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ floatx2 snake2(floatx2 v, float al, float ial) {          // == snake() per element
    const floatx2 x = v * al;
    floatx2 k = x * 0.318309886183790672f;
    k.x = rintf(k.x); k.y = rintf(k.y);
    floatx2 r = fma2(-k, (floatx2)(3.140625f), x);
    r = fma2(-k, (floatx2)(9.67653589793e-4f), r);
    const floatx2 r2 = r * r;
    floatx2 pl = fma2(r2, (floatx2)(-2.50521083854e-8f), (floatx2)(2.75573192240e-6f));
    pl = fma2(r2, pl, (floatx2)(-1.98412698413e-4f));
    pl = fma2(r2, pl, (floatx2)(8.33333333333e-3f));
    pl = fma2(r2, pl, (floatx2)(-1.66666666667e-1f));
    const floatx2 sn = fma2(r * r2, pl, r);
    return fma2((floatx2)(ial), sn * sn, v);
}
// HT: a wave owns HT x HT MFMA tiles (HT = 2: 64 x 64, the throughput shape; HT = 1: 32 x 32, four times as many
// output tiles for launches that would otherwise occupy a fraction of the CUs -- single sentences, token-rate
// layers; same K order per output element, so the results are bit-identical to the HT = 2 kernel's).
// KCT: input channels per staged tile.  16 everywhere except the small-launch kernel of 1-tap convolutions (every
// Linear of the token-rate layers), which takes 32: at 8 MFMAs per tile its K loop is a chain of staging latencies,
// and twice the channels per barrier halves the chain (1-tap only: with more taps a wider tile would change the
// order in which an output element's products are summed, and with it the bits).
// TAPS > 0 (bf16 modes): the tap count is a compile-time constant and the kernel runs the RING pipeline (below, "ring
// pipeline"); TAPS = 0: any tap count / tap mask at run time, two fragment sets one tap ahead.
// HTN: MFMA tiles per wave along t (default HT: square wave tiles).  The ring pipeline of the bf16 modes runs 4 x 1 waves
// of 32 (co) x 128 (t): a wave then loads only ITS 32 weight rows per tap -- with 2 x 2 waves of 64 x 64 the two waves
// that share rows both fetch them, and the L1 -> register return path (TD 88 % busy in the bf16 mode, 33 % MFMA) is what
// those launches wait for.
// KSPLIT (fp32, two-set loop): BLOCKED summation over K.  One accumulator per output element sums its K = cin * taps
// products as ONE sequential fp32 chain, so its rounding noise grows like sqrt(K): measured against a float64 result
// (tools/conv_noise.py) the kernel is as accurate as torch's CPU kernels up to K = 256 and 1.9x / 2.9x / 5.2x noisier at
// K = 768 / 2048 / 3270, because those sum in blocks of a few hundred products.  With KSPLIT the running accumulator is
// folded into a second one every ConvP::kflush 16-channel tiles (~128 products: engine.hip cp()) and cleared; the output is the sum of
// the block sums.  The engine asks for it on the token-rate and prosody layers (ALBERT, duration / F0 / N predictors, text
// encoder: < 3 % of the FLOPs), whose noise reaches the rounded durations and the F0 curve that the harmonic source
// integrates into a phase.  Costs 16 * HT * HTN more registers and two vector instructions per accumulator and block.
// The fold points are positions in K, the same for every tile shape: shapes stay bit-identical to each other.
// XB / YB (bf16 ring kernels): the input / output tensor lives in HBM as bf16 channel pairs (ConvP::x_b16 / y_b16).
template <int WM, int WN, int HT, int EPI, bool SNAKE, int PREC, int KCT = 16, int TAPS = 0, int HTN = HT, bool KSPLIT = false, bool XB = false, bool YB = false>
__global__ __launch_bounds__(64 * WM * WN, WM * WN > 4 ? 1 : TAPS > 0 ? (HT * HTN > 4 ? 1 : K82_OCC_RING) : PREC == 1 ? K82_OCC_BF16 : PREC == 2 ? K82_OCC_SPLIT : K82_OCC_F32) void conv_mfma_kernel(ConvP p) {
    constexpr bool BF16 = PREC != PREC_F32;        // bf16 matrix instruction, 16-byte A fragments
    constexpr bool SPLIT = PREC == PREC_SPLIT;
    constexpr int WT = 32 * HT, WTN = 32 * HTN;    // rows and columns of a wave's output tile
    constexpr int BM = WT * WM, BN = WTN * WN, KC = KCT, WP = BN + 68, NW = WM * WN;
    static_assert(KC == 16 || (KC == 32 && HT == 1 && !BF16), "32-channel tiles: small-launch fp32 kernel only");
    static_assert(!KSPLIT || (PREC == PREC_F32 && TAPS == 0), "blocked summation: the fp32 two-set loop only");
    static_assert(!(XB || YB) || (PREC == PREC_BF16 && TAPS > 0 && HT == 1 && HTN == 4 && SNAKE), "bf16 activations in HBM: the bf16 ring kernels of the Snake convs only");
    static_assert(!YB || EPI == EPI_PLAIN, "bf16 output: plain epilogue only");
    static_assert(HT == 2 || (HT == 1 && (!BF16 || HTN == 4)), "one MFMA tile row per wave: the fp32 small-launch shape and the 32 x 128 bf16 shapes");
    constexpr int CPW = KC / NW;               // channels staged per wave
    constexpr int NCOL = (WP / 4 + 63) / 64;   // float4 sweeps (64 lanes x 4 columns) per channel
    constexpr int PATCH = 32 * 36 + 64;        // per-wave epilogue patch [32][36] + the wave's 64 bias values
    static_assert(KC % NW == 0, "KC must be a multiple of the wave count");
    // ONE __shared__ object (a second one beside the staging array can cost a vmcnt(0) per k-step)
    // LDS3 (fp32 / bf16x3 two-set loop, 2 x 2 and 4 x 1 waves): THREE x-tile buffers and a counter instead of the barrier --
    // a wave signals "my share of tile g+1 is staged" right after its first tap and only waits, a whole ci-tile later,
    // for all NW signals of that step.  Safe with three buffers: a wave that stages tile g+2 (into the buffer of tile
    // g-1) has seen every wave's step-g signal, and a wave's LDS instructions execute in order, so every read of tile
    // g-1 is behind that signal.
    constexpr bool LDS3 = K82_LDS3 && TAPS == 0 && PREC != PREC_BF16 && WM >= 2;
    constexpr int NBUF = LDS3 ? 3 : 2;
    __shared__ __attribute__((aligned(16))) float lds[NBUF * KC * WP + NW * PATCH + 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // provably wave-uniform: Tile stays in SGPRs
    const int wm = wave / WN, wn = wave % WN;
    const int nw = wn * WTN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int ntiles = (p.cin_p + KC - 1) / KC, nt16 = p.cin_p / 16;      // staged tiles; 16-channel tiles of the weight images
    const int MT = p.cout_p / BM, NT = (p.Lmax_out + BN - 1) / BN;
    const int total = MT * NT * p.B, G = gridDim.x;
    const unsigned tapall = (p.taps >= 32) ? 0xffffffffu : ((1u << p.taps) - 1u);
    float* const patch = lds + NBUF * KC * WP + wave * PATCH;
    unsigned* const stepcnt = reinterpret_cast<unsigned*>(lds + NBUF * KC * WP + NW * PATCH);
    float* const bsh = patch + 32 * 36;
#if K82_TRACE
    // phase time stamps of every wave (tools/trace_conv.py reads them): event = id << 56 | s_memtime
    constexpr int TRCAP = 1024;
    unsigned long long* const trb = p.trace + (long)(blockIdx.x * NW + wave) * TRCAP;
    int trn = 0;
    auto TR = [&](unsigned long long id) {
        const unsigned long long t = __builtin_readcyclecounter();
        if (lane == 0 && trn < TRCAP) trb[trn] = (id << 56) | (t & 0x00ffffffffffffffull);
        ++trn;
    };
    { unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      if (lane == 0) trb[0] = ((unsigned long long)xcc << 32) | hw; trn = 1; }
#else
#define TR(id) ((void)0)
#endif

    // ---- persistent tile walk: block g handles output tiles g, g+G, g+2G, ... (m fastest, so the
    // blocks that share an x tile run side by side); everything in Tile is wave-uniform
    struct Tile { int b, m0, n0, Lin, Lout, tbase, shift, W, nact, tap0; unsigned mask; };
    auto decode = [&](int q, Tile& T) -> bool {
        const int mt = q % MT, r = q / MT;
        T.b = r / NT;
        T.n0 = (r % NT) * BN;
        T.Lin = p.lin.len(T.b);
        T.Lout = p.lout.len(T.b);
        if (T.n0 >= T.Lout) return false;
        T.m0 = mt * BM + wm * WT;
        // the LDS tile starts at the 16-byte aligned position tbase <= n0 - pad: every global load is
        // an aligned float4 (rows are 16-B aligned, pitch % 4 == 0), every LDS write a ds_write_b128
        T.tbase = ((T.n0 - p.pad) >> 2) << 2;
        T.shift = (T.n0 - p.pad) - T.tbase;
        T.W = BN + (p.taps - 1) * p.dil + T.shift;
        T.mask = (p.tapmask ? p.tapmask[T.m0 >> 6] : 0xffffffffu) & tapall;
        T.nact = __builtin_popcount(T.mask);
        T.tap0 = T.mask ? __builtin_ctz(T.mask) : 0;
        return T.nact > 0;
    };
    auto next_valid = [&](int q, Tile& T) -> int {      // first valid tile at or after q (stride G), or -1
        while (q < total) {
            if (decode(q, T)) return q;
            q += G;
        }
        return -1;
    };

    floatx16 acc[HT][HTN];
    floatx16 acc2[KSPLIT ? HT : 1][KSPLIT ? HTN : 1];      // KSPLIT: sum of the finished K blocks
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < HT; ++i)
#pragma unroll
            for (int j = 0; j < HTN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[i][j][r] = 0.f;
                    if constexpr (KSPLIT) acc2[i][j][r] = 0.f;
                }
    };
    auto fold_acc = [&](bool clear) {                      // acc2 += acc ; acc = clear ? 0 : acc2
        if constexpr (KSPLIT) {
#pragma unroll
            for (int i = 0; i < HT; ++i)
#pragma unroll
                for (int j = 0; j < HTN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        acc2[i][j][r] += acc[i][j][r];
                        acc[i][j][r] = clear ? 0.f : acc2[i][j][r];
                    }
        }
    };

    // channel (within the ci-tile) that staging slot cc of this wave carries: fp32 w, w+NW, ...; bf16 modes the
    // channel pairs (2w, 2w+1) and (8+2w, 8+2w+1), i.e. rows w and 4+w of the packed tile
    auto chan = [&](int cc) -> int { return BF16 ? 2 * (wave + NW * (cc >> 1)) + (cc & 1) : wave + cc * NW; };
    static_assert(!BF16 || (NW * CPW == 16 && CPW % 2 == 0), "the packed bf16 staging: every wave carries whole channel pairs of a 16-channel tile");
    float4 raw[CPW][NCOL], nmr[CPW];
    uint4 rawb[CPW / 2 > 0 ? CPW / 2 : 1][NCOL];            // XB: the loads of a channel PAIR row (4 columns x 2 channels per lane)
    auto xb_half = [](unsigned d, int h) -> float { return __uint_as_float(h ? (d & 0xffff0000u) : (d << 16)); };
    float alr[CPW], ialr[CPW];
    // Per-channel transform parameters always come from a readable address (the padded bias vector stands
    // in when there is no AdaIN / Snake), so they are prefetched unconditionally with the tile.
    const float4* const normp = (p.tr == TR_NORM) ? p.norm : reinterpret_cast<const float4*>(p.bias);
    const int nstride = (p.tr == TR_NORM) ? p.nstride : 0, nclamp = (p.tr == TR_NORM) ? p.cin - 1 : 0;
    const float4 nid = make_float4(0.f, p.tr == TR_SCALE ? p.in_scale : 1.f, 1.f, 0.f);
    const float slope = (p.act == ACT_LEAKY) ? p.slope : 1.f;
    const int wlast = (BN + (p.taps - 1) * p.dil + 2) & ~3;     // offset of the last float4 of an x tile that any tap reads
    // raw global loads of one ci-tile (no transform yet): wave w takes channels w, w+NW, ...
    // All loads are UNCONDITIONAL with clamped (always in-bounds) coordinates; validity is applied
    // later with selects.
    auto load_tile = [&](const Tile& T, int ci0, bool params = true) {      // params: a constant at every call site
        const float* xb = p.x + (long)T.b * p.xbs;
        // last aligned float4 that starts inside the row AND inside the columns the taps read (BN + halo + shift <= 3):
        // a 64-lane sweep covers 256 columns, a 128-column tile needs 131-181 of them -- the lanes beyond re-read the
        // tile's last float4 (same cache line as their neighbour's) instead of pulling the next tile's columns through
        // L2 and HBM (round 2 measured 2.2x the algorithmic read traffic on the Snake convs, profiles/r02_v6_traffic.json)
        const int tlast = min(((T.Lin - 1) >> 2) << 2, T.tbase + wlast);
        if constexpr (XB) {
#pragma unroll
            for (int pr = 0; pr < CPW / 2; ++pr) {
                const unsigned* xr = reinterpret_cast<const unsigned*>(xb) + (long)((ci0 >> 1) + wave + NW * pr) * p.xpitch;   // (cin % 16 == 0: every pair is real)
#pragma unroll
                for (int k = 0; k < NCOL; ++k) {
                    const int t = min(max(T.tbase + 4 * (lane + 64 * k), 0), tlast);
                    rawb[pr][k] = *reinterpret_cast<const uint4*>(xr + t);
                }
            }
        }
#pragma unroll
        for (int cc = 0; cc < CPW; ++cc) {
            const int ci = min(ci0 + chan(cc), p.cin - 1);
            const float* xr = xb + (long)ci * p.xpitch;
            if constexpr (!XB) {
#pragma unroll
            for (int k = 0; k < NCOL; ++k) {
                const int t = min(max(T.tbase + 4 * (lane + 64 * k), 0), tlast);
                raw[cc][k] = *reinterpret_cast<const float4*>(xr + t);
            }
            }
            if (params) {
                nmr[cc] = normp[(long)T.b * nstride + min(ci, nclamp)];
                if constexpr (SNAKE) { alr[cc] = p.alpha[ci]; ialr[cc] = p.inv_alpha[ci]; }
            }
        }
    };
    // transform (AdaIN affine, activation) and write to the LDS buffer; zero padding stays zero.
    // Branch-free per element: "no transform" is the affine (0, 1, 1, 0), TR_SCALE is (0, s, 1, 0),
    // "no activation" is LeakyReLU with slope 1 -- all exact identities in fp32 -- and Snake is a
    // template parameter.
    // VALU budget (the two waves of a SIMD share its vector issue, so every instruction here is paid twice
    // when they run in step): the affine is folded to one fma per element (scale = rstd * (1 + gamma),
    // offset = beta - mean * scale: within 2 ulp of the reference's (x - mean) * rstd * (1 + gamma) + beta),
    // LeakyReLU is max(u, slope * u) (slope in [0, 1], checked on the host), validity is one unsigned
    // compare against the row length (0 for padded channels).
    auto write_tile = [&](const Tile& T, int ci0, float* dst) {
        float out[CPW][NCOL][4];
        // (Measured and not kept: touching the per-channel parameters with an opaque asm here so that "scale = rstd * (1 +
        // gamma)" is computed behind the tap's MFMAs -- instruction selection places it right behind the loads, in front
        // of the tap, with an s_waitcnt for their L2 round trip.  The wait is real (tools/trace_conv.py: the first tap of
        // a ci-tile takes 1.2-2.1 us longer than the others) but the partner wave covers it: pinned 2044, unpinned 2063
        // audio-s/s on one box.)
#pragma unroll
        for (int cc = 0; cc < CPW; ++cc) {
            const unsigned lin = (ci0 + chan(cc) < p.cin) ? (unsigned)T.Lin : 0u;
            const float4 nm = (p.tr == TR_NORM) ? nmr[cc] : nid;
            const float sc = nm.y * nm.z, of = fmaf(-nm.x, sc, nm.w);
#pragma unroll
            for (int k = 0; k < NCOL; ++k) {
                const unsigned t = (unsigned)(T.tbase + 4 * (lane + 64 * k));
                if constexpr (!BF16) {
                    // fp32 mode: two elements per packed instruction (measured: Snake convs +2..10 %; in the bf16
                    // modes the register pairs get in the way of the split / pack step: -5..15 %, scalar form kept)
                    const floatx2 v2[2] = {{raw[cc][k].x, raw[cc][k].y}, {raw[cc][k].z, raw[cc][k].w}};
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        floatx2 u = fma2(v2[h], (floatx2)(sc), (floatx2)(of));
#if K82_F32_HWSIN
                        if constexpr (SNAKE) {      // hardware sine (argument in revolutions, reduced with v_fract)
                            const floatx2 xr = u * (alr[cc] * kInvTwoPi);
                            floatx2 sn;
                            sn.x = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(xr.x));
                            sn.y = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(xr.y));
                            u = fma2((floatx2)(ialr[cc]), sn * sn, u);
                        }
#else
                        if constexpr (SNAKE) u = snake2(u, alr[cc], ialr[cc]);   // (hw sin in fp32 mode: +1 %, not taken)
#endif
                        else { const floatx2 w = u * slope; u.x = fmaxf(u.x, w.x); u.y = fmaxf(u.y, w.y); }
                        out[cc][k][2 * h] = (t + 2 * h < lin) ? u.x : 0.f;      // t < 0 wraps to a huge unsigned: also invalid
                        out[cc][k][2 * h + 1] = (t + 2 * h + 1 < lin) ? u.y : 0.f;
                    }
                } else {
                    float v[4];
                    if constexpr (XB) {
                        const uint4 d = rawb[cc >> 1][k];
                        v[0] = xb_half(d.x, cc & 1); v[1] = xb_half(d.y, cc & 1); v[2] = xb_half(d.z, cc & 1); v[3] = xb_half(d.w, cc & 1);
                    } else {
                        v[0] = raw[cc][k].x; v[1] = raw[cc][k].y; v[2] = raw[cc][k].z; v[3] = raw[cc][k].w;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float u = fmaf(v[e], sc, of);
                        if constexpr (SNAKE) u = snake_fast(u, alr[cc] * kInvTwoPi, ialr[cc]);
                        else u = fmaxf(u, u * slope);
                        out[cc][k][e] = (t + e < lin) ? u : 0.f;
                    }
                }
            }
        }
        // lanes past the row end write to a dump slot in the wave's (idle) epilogue patch: no branch
        if constexpr (!BF16) {
#pragma unroll
            for (int cc = 0; cc < CPW; ++cc)
#pragma unroll
                for (int k = 0; k < NCOL; ++k) {
                    const int col = 4 * (lane + 64 * k);
                    float* d = (col < WP) ? dst + chan(cc) * WP + col : patch + 4 * lane;
                    *reinterpret_cast<float4*>(d) = make_float4(out[cc][k][0], out[cc][k][1], out[cc][k][2], out[cc][k][3]);
                }
        } else {
#pragma unroll
            for (int pr = 0; pr < CPW / 2; ++pr)
#pragma unroll
                for (int k = 0; k < NCOL; ++k) {
                    const int col = 4 * (lane + 64 * k);
                    unsigned hi[4], lo[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float u0 = out[2 * pr][k][e], u1 = out[2 * pr + 1][k][e];
                        hi[e] = pack_bf16(u0, u1);
                        if constexpr (SPLIT)
                            lo[e] = pack_bf16(u0 - __uint_as_float(hi[e] << 16), u1 - __uint_as_float(hi[e] & 0xffff0000u));
                    }
                    float* d = (col < WP) ? dst + (wave + NW * pr) * WP + col : patch + 4 * lane;
                    *reinterpret_cast<uint4*>(d) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
                    if constexpr (SPLIT) {
                        float* d2 = (col < WP) ? dst + (8 + wave + NW * pr) * WP + col : patch + 4 * lane;
                        *reinterpret_cast<uint4*>(d2) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
                    }
                }
        }
    };
    // (Measured and not kept: a second copy of this step without the validity selects for interior tiles -- 40 of ~170
    // vector instructions fewer per ci-tile, 1.5 % SLOWER on configs[1]; profiles/r02_ab_stage_fast.txt.)

    // Two sets of A fragments: the one being consumed and the one being loaded.  fp32: f[kk][mi] scalars;
    // bf16 modes: q[] are the lane's 16-byte fragments exactly as loaded (q[mi], or q[2 mi] = hi / q[2 mi + 1] = lo
    // in the split mode) -- kept as whole 4-register tuples so the load's destination IS the MFMA operand.
    struct ASet { float f[KC / 2][2]; uint4 q[4]; };
    ASet a0, a1;
    // Addresses are (wave-uniform 64-bit base in SGPRs) + (32-bit lane offset in one VGPR): the per-tap address
    // arithmetic runs on the scalar unit and the loads use the saddr form -- vector instructions in the tap loop
    // compete with the MFMAs for issue, scalar ones do not.
    const unsigned a_lane = BF16 ? (unsigned)((lhi * p.cout_p + l31) * (SPLIT ? 32 : 16))
                            : (WN == 2) ? (unsigned)((lhi * p.cout_p + l31) * 32)
                                                   : (unsigned)((lhi * p.cout_p + l31) * 4);
    auto load_a = [&](int m0, int ci0, int tap, ASet& a) {
        if constexpr (BF16) {
            constexpr int FB = SPLIT ? 32 : 16;                // bytes per (row, k-half): hi | lo fragments, or one
            const char* base = reinterpret_cast<const char*>(p.wb) +
                               ((long)((tap * nt16 + ci0 / 16) * 2) * p.cout_p + m0) * FB;
            const uint4* wp = reinterpret_cast<const uint4*>(base + a_lane);
            if constexpr (SPLIT) { a.q[0] = wp[0]; a.q[1] = wp[1]; if constexpr (HT == 2) { a.q[2] = wp[64]; a.q[3] = wp[65]; } }
            else { a.q[0] = wp[0]; if constexpr (HT == 2) a.q[1] = wp[32]; }
        } else if constexpr (WN == 2) {                         // 2 x 2 and 4 x 2 waves (both wave-tile sizes)
            // fragment image: the lane's 8 k-step values of a row are 32 contiguous bytes -- 4 x 16-byte loads per
            // tap instead of 16 dword loads
            const char* base = reinterpret_cast<const char*>(p.wf) +
                               ((long)((tap * nt16 + ci0 / 16) * 2) * p.cout_p + m0) * 32;
#pragma unroll
            for (int h = 0; h < KC / 16; ++h)                   // 16-channel sub-tiles (the image has one spare tile at its end)
#pragma unroll
                for (int mi = 0; mi < HT; ++mi) {
                    const float4* wq = reinterpret_cast<const float4*>(base + (long)h * p.cout_p * 64 + a_lane);
                    const float4 u = wq[64 * mi], v = wq[64 * mi + 1];
                    a.f[8 * h + 0][mi] = u.x; a.f[8 * h + 1][mi] = u.y; a.f[8 * h + 2][mi] = u.z; a.f[8 * h + 3][mi] = u.w;
                    a.f[8 * h + 4][mi] = v.x; a.f[8 * h + 5][mi] = v.y; a.f[8 * h + 6][mi] = v.z; a.f[8 * h + 7][mi] = v.w;
                }
        } else {
            const char* base = reinterpret_cast<const char*>(p.w) + ((long)(tap * p.cin_p + ci0) * p.cout_p + m0) * 4;
#pragma unroll
            for (int kk = 0; kk < KC / 2; ++kk) {
                const float* wp = reinterpret_cast<const float*>(base + (long)(2 * kk) * p.cout_p * 4 + a_lane);
                a.f[kk][0] = wp[0];
                if constexpr (HT == 2) a.f[kk][1] = wp[32];
            }
        }
    };
    // bf16 modes: B fragments of one 32-column half.  bpre carries the ni = 0 half of the NEXT step across steps:
    // it is read while the MFMAs of the current step's second half run, so the only LDS latency a wave exposes is
    // the first read after each barrier (the two waves of a SIMD tend to run in phase, so an exposed read stalls
    // the matrix pipe for both).
    struct BFrag { union { bf16x8 v; unsigned u[4]; } h, l; };
    BFrag bpre;
    auto read_b = [&](const float* xsb, int off, int ni, BFrag& f) {
        // lane: channel pairs 4*lhi .. +3 (rows of the packed tile), column l31.  One group per column half: the
        // compiler fence keeps the load merger from pairing ni = 0 with ni = 1 dwords, which belong to different
        // operand tuples and would have to be copied apart again
        const float* xsp = xsb + (4 * lhi) * WP + nw + l31 + off + 32 * ni;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f.h.u[j] = __float_as_uint(xsp[j * WP]);
            if constexpr (SPLIT) f.l.u[j] = __float_as_uint(xsp[(8 + j) * WP]);
        }
        asm volatile("" ::: "memory");
    };
    auto mma_half = [&](const ASet& a, int ni, const BFrag& f) {
        if constexpr (HT == 2 && HTN == 2)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            if constexpr (SPLIT) {                                     // small terms first
                const bf16x8 ah = __builtin_bit_cast(bf16x8, a.q[2 * mi]), al = __builtin_bit_cast(bf16x8, a.q[2 * mi + 1]);
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, f.h.v, acc[mi][ni], 0, 0, 0);
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, f.l.v, acc[mi][ni], 0, 0, 0);
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, f.h.v, acc[mi][ni], 0, 0, 0);
            } else {
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a.q[mi]), f.h.v, acc[mi][ni], 0, 0, 0);
            }
        }
    };
    auto mma_tile14 = [&](const ASet& a, int ni, const BFrag& f) {       // HT = 1, HTN = 4 (32 x 128 wave tile), generic loop
        if constexpr (SPLIT) {                                         // same three terms, same order as mma_half
            const bf16x8 ah = __builtin_bit_cast(bf16x8, a.q[0]), al = __builtin_bit_cast(bf16x8, a.q[1]);
            acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, f.h.v, acc[0][ni], 0, 0, 0);
            acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, f.l.v, acc[0][ni], 0, 0, 0);
            acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, f.h.v, acc[0][ni], 0, 0, 0);
        } else {
            acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a.q[0]), f.h.v, acc[0][ni], 0, 0, 0);
        }
    };
    // fp32: B fragments (k-step kk: rows 2 kk + lhi of the tile) as single ds_read_b32 with immediate offsets; the
    // compiler fences keep the load merger from pairing them into ds_read2_b32 (see mfma_tap)
    float bpf[HTN] = {};
    auto read_b32 = [&](const float* xsp, int kk, float (&d)[HTN]) {
#pragma unroll
        for (int ni = 0; ni < HTN; ++ni) {
            d[ni] = xsp[2 * kk * WP + 32 * ni];
            asm volatile("" ::: "memory");
        }
    };
    // offn: column offset of the next tap of the SAME ci-tile (any valid offset when there is none: the prefetched
    // fragments are then simply not used -- the first step after the barrier reads its own)
    auto mfma_tap = [&](const float* xsb, int off, int offn, const ASet& a) {
        if constexpr (BF16 && HT == 1 && HTN == 4) {
            // 32 x 128 wave tile in the two-set loop (the split mode: a wave fetches only ITS 32 weight rows, half the
            // bytes per MFMA through the L1 return path that bounds these kernels; B fragments one column block ahead)
            BFrag f[2];
            read_b(xsb, off, 1, f[1]);
            __builtin_amdgcn_sched_barrier(0);
            mma_tile14(a, 0, bpre);
            __builtin_amdgcn_sched_barrier(0);
            read_b(xsb, off, 2, f[0]);
            __builtin_amdgcn_sched_barrier(0);
            mma_tile14(a, 1, f[1]);
            __builtin_amdgcn_sched_barrier(0);
            read_b(xsb, off, 3, f[1]);
            __builtin_amdgcn_sched_barrier(0);
            mma_tile14(a, 2, f[0]);
            __builtin_amdgcn_sched_barrier(0);
            read_b(xsb, offn, 0, bpre);
            __builtin_amdgcn_sched_barrier(0);
            mma_tile14(a, 3, f[1]);
        } else if constexpr (BF16) {
            BFrag f1;
            read_b(xsb, off, 1, f1);
            __builtin_amdgcn_sched_barrier(0);
            mma_half(a, 0, bpre);
            __builtin_amdgcn_sched_barrier(0);
            read_b(xsb, offn, 0, bpre);
            __builtin_amdgcn_sched_barrier(0);
            mma_half(a, 1, f1);
        } else {
            // All B fragments of the tap are requested up front (k-step 0 came with the previous tap), one ds_read_b32
            // each off ONE address register: fp32 MFMAs share the SIMD's vector issue with every VALU instruction
            // (tools/ubench_coissue.hip: time adds, ~5 cycles per instruction, ~20 for one that sits alone between two
            // MFMAs), LDS instructions do not.  The compiler's own schedule was: read a pair (ds_read2_b32, whose 8-bit
            // offsets need a v_add_u32 per k-step), wait for it, four MFMAs, next pair.
            const float* xsp = xsb + lhi * WP + nw + l31 + off;
            const float* xnp = xsb + lhi * WP + nw + l31 + offn;
            float bb[KC / 2][HTN];
#pragma unroll
            for (int ni = 0; ni < HTN; ++ni) bb[0][ni] = bpf[ni];
#pragma unroll
            for (int kk = 1; kk < KC / 2; ++kk) read_b32(xsp, kk, bb[kk]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < KC / 2; ++kk) {
                if (kk == KC / 2 - 2) read_b32(xnp, 0, bpf);
#pragma unroll
                for (int mi = 0; mi < HT; ++mi)
#pragma unroll
                    for (int ni = 0; ni < HTN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.f[kk][mi], bb[kk][ni], acc[mi][ni], 0, 0, 0);
            }
        }
    };

    // (Tried and measured on one box, none kept: slicing the staging transform between the MFMA pairs of the
    // second tap so it issues in their shadow -- 0..2 % slower; sched_group_barrier patterns for the same --
    // ignored by the scheduler; non-temporal epilogue stores -- no change; the stores themselves cost < 5 %.)
    // ---- epilogue of one output tile.  D layout: column j = lane&31 (t), row i = (r&3) + 8*(r>>2)
    // + 4*(lane>>5) (co).  The wide epilogues send each 32x32 accumulator tile through the wave's LDS
    // patch and read it back as float4 along t: 4 dwordx4 stores of 8 rows x 128 B per tile, float4
    // residual / running-mean traffic; the residual loads of 32x32 tile i+1 are in flight while tile i
    // is finished.  Rows are padded to a multiple of 4 floats and nothing downstream reads the padding
    // (every consumer masks by length), so a float4 that starts inside the row is stored whole.
    // bias_l = bias[m0 + lane] was loaded at the start of the tile and reaches the lanes through LDS.
    // Epilogue VALU budget: on this chip every vector instruction of ANY wave of the SIMD takes its issue cycles away
    // from the fp32 MFMAs (tools/ubench_coissue.hip: the times add), so the common case -- a wave tile that lies wholly
    // inside the sentence and the real channels, no GELU -- runs a lean path: no clamps, no validity selects, no
    // exec-masked stores, row addressing on the scalar unit (per-lane 32-bit byte offsets, host-checked), packed fp32
    // arithmetic in the fp32 mode (plain in the bf16 modes: v_pk_* does not overlap the bf16 MFMAs, plain VALU does),
    // the 8-lane statistics sums as v_add_f32_dpp.  Same operations per element in the same order as the general
    // path: the two give the same bits (tiles of one tensor take different paths).
    // The wide epilogues read their operands (output / residual / running-mean / statistics pointers, pitches, scales)
    // from the kernel-argument segment when they start, through an opaque pointer: kept in SGPRs across the main loop
    // they are spilled to VGPR lanes (the kernel needs ~130 scalar values) and every use costs a v_readlane -- a
    // vector instruction; an s_load costs none.
    struct EpiArgs { float* y; const float* res; float* accbuf; float2* stat; long ybs, rbs, abs_, sbs; int ypitch, rpitch, apitch, snt, acc_init, out_act; float out_scale, acc_scale; };
    auto epi_args = [&]() -> EpiArgs {
        typedef const ConvP __attribute__((address_space(4))) * KArg;
        KArg k = (KArg)__builtin_amdgcn_kernarg_segment_ptr();      // ConvP is the kernel's only argument
        asm volatile("" : "+s"(k));
        typedef __attribute__((address_space(1))) float GF;
        typedef __attribute__((address_space(1))) float2 GF2;
        EpiArgs e;
        e.y = (float*)(GF*)k->y; e.res = (const float*)(const GF*)k->res; e.accbuf = (float*)(GF*)k->accbuf; e.stat = (float2*)(GF2*)k->stat;
        e.ybs = k->ybs; e.rbs = k->rbs; e.abs_ = k->abs_; e.sbs = k->sbs;
        e.ypitch = k->ypitch; e.rpitch = k->rpitch; e.apitch = k->apitch; e.snt = k->snt; e.acc_init = k->acc_init; e.out_act = k->out_act;
        e.out_scale = k->out_scale; e.acc_scale = k->acc_scale;
        return e;
    };
    auto dpp_sum8x2 = [&](float& s1, float& s2) {
        // s_nop: a VALU write followed by a DPP read of the same register needs two wait states, and the hazard
        // recogniser does not look inside an asm block
        asm volatile("s_nop 1\n\t"
                     "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                     "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                     "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                     "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1"
                     : "+v"(s1), "+v"(s2));
    };
    auto epilogue_fast = [&](const Tile& T, const EpiArgs& E) {
        typedef float floatx4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(1))) floatx4 G4;       // global address space spelled out: the addresses are integers here
        typedef __attribute__((address_space(1))) floatx2 G2;
        const int i0 = lane >> 3, c4 = (lane & 7) * 4;
        const int row = T.m0 + i0, col = T.n0 + nw + c4;
        // per-lane addresses of the tile's first row; the rows are walked by adding the (scalar) 8-row step.  pin(): an
        // opaque no-op on the running address -- otherwise all 16 row addresses are formed ahead of the loop, spilled
        // and fetched back (vector instructions again)
        auto pin = [](unsigned long& a) { asm volatile("" : "+v"(a)); };
        const unsigned long y0 = (unsigned long)(E.y + (long)T.b * E.ybs) + (unsigned)(row * E.ypitch + col) * 4u;
        const unsigned long r0 = EPI >= EPI_RES ? (unsigned long)(E.res + (long)T.b * E.rbs) + (unsigned)(row * E.rpitch + col) * 4u : 0ul;
        const unsigned long a0 = EPI == EPI_ACC ? (unsigned long)(E.accbuf + (long)T.b * E.abs_) + (unsigned)(row * E.apitch + col) * 4u : 0ul;
        const unsigned long s0 = (unsigned long)(E.stat + (long)T.b * E.sbs + ((T.n0 + nw) >> 5)) + (unsigned)(row * E.snt) * 8u;
        const long ystep = (long)E.ypitch * 32, rstep = (long)E.rpitch * 32, astep = (long)E.apitch * 32, sstep = (long)E.snt * 64;   // 8 rows, bytes
        const bool store_y = EPI != EPI_ACC || E.y;
        float4 r4[2][4], a4[2][4];
        auto issue = [&](int idx, float4 (&r)[4], float4 (&a)[4]) {
            const int mi = idx / HTN, ni = idx % HTN;
            unsigned long ra = r0 + 4 * mi * rstep, aa = a0 + 4 * mi * astep;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (EPI >= EPI_RES) { pin(ra); const floatx4 t = *reinterpret_cast<const G4*>(ra + ni * 128); r[q] = make_float4(t.x, t.y, t.z, t.w); ra += rstep; }
                if constexpr (EPI == EPI_ACC) { pin(aa); const floatx4 t = *reinterpret_cast<const G4*>(aa + ni * 128); a[q] = make_float4(t.x, t.y, t.z, t.w); aa += astep; }
            }
        };
        issue(0, r4[0], a4[0]);
#pragma unroll
        for (int idx = 0; idx < HT * HTN; ++idx) {
            const int mi = idx / HTN, ni = idx % HTN;
            if (idx + 1 < HT * HTN) issue(idx + 1, r4[(idx + 1) & 1], a4[(idx + 1) & 1]);
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lhi) * 36 + l31] = acc[mi][ni][r];
            unsigned long ya = y0 + 4 * mi * ystep, aa = a0 + 4 * mi * astep, sa = s0 + 4 * mi * sstep;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + 8 * q;
                const float4 v4 = *reinterpret_cast<const float4*>(patch + i * 36 + c4);
                const float bs = bsh[mi * 32 + i];
                float v[4], av[4] = {0.f, 0.f, 0.f, 0.f};
                if constexpr (!BF16) {
                    floatx2 x0 = {v4.x, v4.y}, x1 = {v4.z, v4.w};
                    x0 = x0 + bs; x1 = x1 + bs;
                    if constexpr (EPI >= EPI_RES) {
                        const float4 u = r4[idx & 1][q];
                        x0 = x0 + (floatx2){u.x, u.y}; x1 = x1 + (floatx2){u.z, u.w};
                    }
                    x0 = x0 * E.out_scale; x1 = x1 * E.out_scale;
                    if constexpr (EPI == EPI_ACC) {
                        const float4 u = a4[idx & 1][q];
                        floatx2 c0 = {u.x, u.y}, c1 = {u.z, u.w};
                        if (E.acc_init) { c0 = (floatx2)(0.f); c1 = (floatx2)(0.f); }
                        c0 = fma2(x0, (floatx2)(E.acc_scale), c0); c1 = fma2(x1, (floatx2)(E.acc_scale), c1);
                        av[0] = c0.x; av[1] = c0.y; av[2] = c1.x; av[3] = c1.y;
                    }
                    v[0] = x0.x; v[1] = x0.y; v[2] = x1.x; v[3] = x1.y;
                } else {
                    v[0] = v4.x; v[1] = v4.y; v[2] = v4.z; v[3] = v4.w;
                    float rs[4] = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (EPI >= EPI_RES) { const float4 u = r4[idx & 1][q]; rs[0] = u.x; rs[1] = u.y; rs[2] = u.z; rs[3] = u.w; }
                    if constexpr (EPI == EPI_ACC) {
                        const float4 u = a4[idx & 1][q];
                        av[0] = E.acc_init ? 0.f : u.x; av[1] = E.acc_init ? 0.f : u.y;
                        av[2] = E.acc_init ? 0.f : u.z; av[3] = E.acc_init ? 0.f : u.w;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float u = v[e] + bs;
                        if constexpr (EPI >= EPI_RES) u += rs[e];
                        u *= E.out_scale;
                        v[e] = u;
                        av[e] = fmaf(u, E.acc_scale, av[e]);
                    }
                }
                if (E.stat) {                                   // wave-uniform
                    float s1 = 0.f + v[0], s2 = v[0] * v[0];
#pragma unroll
                    for (int e = 1; e < 4; ++e) { s1 += v[e]; s2 = fmaf(v[e], v[e], s2); }
                    dpp_sum8x2(s1, s2);
                    int c4o = c4;
                    asm volatile("" : "+v"(c4o));               // (compared afresh: the hoisted lane mask would be another spilled SGPR pair)
                    pin(sa);
                    if (c4o == 0) *reinterpret_cast<G2*>(sa + ni * 8) = (floatx2){s1, s2};
                    sa += sstep;
                }
                if (store_y) { pin(ya); *reinterpret_cast<G4*>(ya + ni * 128) = (floatx4){v[0], v[1], v[2], v[3]}; ya += ystep; }
                if constexpr (EPI == EPI_ACC) { pin(aa); *reinterpret_cast<G4*>(aa + ni * 128) = (floatx4){av[0], av[1], av[2], av[3]}; aa += astep; }
            }
        }
    };
    // YB: the output goes to HBM as bf16 channel pairs (ConvP::y_b16).  Per 32 x 32 accumulator tile a lane takes the
    // PAIR of rows (2 pr, 2 pr + 1), pr = lane / 8 + 8 q, and four columns: two float4 reads of the patch, one 16-byte
    // store of four packed dwords -- half the store instructions and bytes of the fp32 epilogue.  The statistics partials
    // are those of the fp32 values, in the order the fp32 epilogues sum them (same bits as with an fp32 t1).
    auto epilogue_b16 = [&](const Tile& T, const EpiArgs& E) {
        const int p0 = lane >> 3, c4 = (lane & 7) * 4;
        unsigned* const yb = reinterpret_cast<unsigned*>(E.y + (long)T.b * E.ybs);
#pragma unroll
        for (int idx = 0; idx < HT * HTN; ++idx) {
            const int mi = idx / HTN, ni = idx % HTN;
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lhi) * 36 + l31] = acc[mi][ni][r];
            const int t = T.n0 + nw + ni * 32 + c4;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int pr = p0 + 8 * q;                      // pair row inside the 32-row tile
                float v[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = 2 * pr + h;
                    const float4 v4 = *reinterpret_cast<const float4*>(patch + i * 36 + c4);
                    const float bs = bsh[mi * 32 + i];
                    v[h][0] = (v4.x + bs) * E.out_scale; v[h][1] = (v4.y + bs) * E.out_scale;
                    v[h][2] = (v4.z + bs) * E.out_scale; v[h][3] = (v4.w + bs) * E.out_scale;
                    if (E.stat) {                               // wave-uniform
                        float s1 = 0.f, s2 = 0.f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float u = (t + e < T.Lout) ? v[h][e] : 0.f;
                            s1 += u;
                            s2 = fmaf(u, u, s2);
                        }
                        s1 = sum8(s1);
                        s2 = sum8(s2);
                        if (c4 == 0 && t < T.Lout)
                            E.stat[(long)T.b * E.sbs + (long)(T.m0 + mi * 32 + i) * E.snt + (t >> 5)] = make_float2(s1, s2);
                    }
                }
                if (t < T.Lout)                                 // (rows: cout % 32 == 0, host-checked)
                    *reinterpret_cast<uint4*>(yb + (long)(((T.m0 + mi * 32) >> 1) + pr) * E.ypitch + t) =
                        make_uint4(pack_bf16(v[0][0], v[1][0]), pack_bf16(v[0][1], v[1][1]), pack_bf16(v[0][2], v[1][2]), pack_bf16(v[0][3], v[1][3]));
            }
        }
    };
    auto epilogue = [&](const Tile& T, float bias_l) {
        if constexpr (EPI == EPI_PART) {                       // split-K: this workgroup's block sum, as it is
            (void)bias_l;
            const int z = (int)blockIdx.x % p.nblk;
            float* pb = p.part + ((long)(z * p.B + T.b) * p.cout_p) * p.ppitch;
#pragma unroll
            for (int mi = 0; mi < HT; ++mi)
#pragma unroll
                for (int ni = 0; ni < HTN; ++ni) {
                    const int t = T.n0 + nw + ni * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = T.m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                        if (t < T.Lout) pb[(long)m * p.ppitch + t] = acc[mi][ni][r];
                    }
                }
            return;
        }
        bsh[lane] = bias_l;
        if constexpr (YB) {
            epilogue_b16(T, epi_args());
            return;
        }
        if constexpr (EPI != EPI_SCALAR) {
            const EpiArgs E = epi_args();
            const long yb = (long)T.b * E.ybs, rb = (long)T.b * E.rbs, ab = (long)T.b * E.abs_;
#if K82_EPI_FAST
            if (E.out_act != OACT_GELU_NEW && T.n0 + nw + WTN <= T.Lout && T.m0 + WT <= p.cout) { epilogue_fast(T, E); return; }
#endif
            const int i0 = lane >> 3, c4 = (lane & 7) * 4;
            const int tl = ((T.Lout - 1) >> 2) << 2;
            float4 r4[2][4], a4[2][4];
            auto issue = [&](int idx, float4 (&r)[4], float4 (&a)[4]) {
                const int mi = idx / HTN, ni = idx % HTN;
                const int tc = min(T.n0 + nw + ni * 32 + c4, tl);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int mc = min(T.m0 + mi * 32 + i0 + 8 * q, p.cout - 1);
                    if constexpr (EPI >= EPI_RES) r[q] = *reinterpret_cast<const float4*>(E.res + rb + (long)mc * E.rpitch + tc);
                    if constexpr (EPI == EPI_ACC) a[q] = *reinterpret_cast<const float4*>(E.accbuf + ab + (long)mc * E.apitch + tc);
                }
            };
            issue(0, r4[0], a4[0]);
#pragma unroll
            for (int idx = 0; idx < HT * HTN; ++idx) {
                const int mi = idx / HTN, ni = idx % HTN;
                if (idx + 1 < HT * HTN) issue(idx + 1, r4[(idx + 1) & 1], a4[(idx + 1) & 1]);
#pragma unroll
                for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lhi) * 36 + l31] = acc[mi][ni][r];
                const int t = T.n0 + nw + ni * 32 + c4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = i0 + 8 * q;
                    const float4 v4 = *reinterpret_cast<const float4*>(patch + i * 36 + c4);
                    const float bs = bsh[mi * 32 + i];
                    const int m = T.m0 + mi * 32 + i;
                    float v[4] = {v4.x, v4.y, v4.z, v4.w};
                    float rs[4] = {0.f, 0.f, 0.f, 0.f}, av[4] = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (EPI >= EPI_RES) { const float4 u = r4[idx & 1][q]; rs[0] = u.x; rs[1] = u.y; rs[2] = u.z; rs[3] = u.w; }
                    if constexpr (EPI == EPI_ACC) {
                        const float4 u = a4[idx & 1][q];
                        av[0] = E.acc_init ? 0.f : u.x; av[1] = E.acc_init ? 0.f : u.y;
                        av[2] = E.acc_init ? 0.f : u.z; av[3] = E.acc_init ? 0.f : u.w;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float u = v[e] + bs;
                        if (E.out_act == OACT_GELU_NEW) u = gelu_new(u);
                        if constexpr (EPI >= EPI_RES) u += rs[e];
                        u *= E.out_scale;
                        v[e] = u;
                        av[e] = av[e] + u * E.acc_scale;
                    }
                    if (E.stat) {                               // wave-uniform: instance-norm partials of this row span
                        float s1 = 0.f, s2 = 0.f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float u = (t + e < T.Lout) ? v[e] : 0.f;
                            s1 += u;
                            s2 = fmaf(u, u, s2);
                        }
                        s1 = sum8(s1);
                        s2 = sum8(s2);
                        if (c4 == 0 && m < p.cout && t < T.Lout)
                            E.stat[(long)T.b * E.sbs + (long)m * E.snt + (t >> 5)] = make_float2(s1, s2);
                    }
                    if (m < p.cout && t < T.Lout) {             // stores only under the lane-dependent condition
                        if (EPI != EPI_ACC || E.y)
                            *reinterpret_cast<float4*>(E.y + yb + (long)m * E.ypitch + t) = make_float4(v[0], v[1], v[2], v[3]);
                        if constexpr (EPI == EPI_ACC)
                            *reinterpret_cast<float4*>(E.accbuf + ab + (long)m * E.apitch + t) = make_float4(av[0], av[1], av[2], av[3]);
                    }
                }
            }
        } else {
            const long yb = (long)T.b * p.ybs, rb = (long)T.b * p.rbs;
            const int upmask = (1 << p.up_cout_shift) - 1;
            const float* resp = p.res ? p.res + rb : p.bias;      // always a readable address: no branch per load
            const int rpitch = p.res ? p.rpitch : 0;
            const float rmul = p.res ? 1.f : 0.f;
#pragma unroll
            for (int mi = 0; mi < HT; ++mi)
#pragma unroll
                for (int ni = 0; ni < HTN; ++ni) {
                    const int t = T.n0 + nw + ni * 32 + l31;
                    const int tc = min(t, T.Lout - 1);
                    float rv[16];
                    int off[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int mc = min(T.m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi, p.cout - 1);
                        int row = mc, col = tc;
                        if (p.up_s) { row = mc & upmask; col = p.up_s * tc + (mc >> p.up_cout_shift) + p.up_off; }
                        off[r] = row * p.ypitch + col;
                        rv[r] = resp[p.res ? (long)row * rpitch + (col >> p.res_shift) : 0];
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int mrow = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                        float v = acc[mi][ni][r] + bsh[mrow];
                        if (p.out_act == OACT_GELU_NEW) v = gelu_new(v);
                        v = (v + rv[r] * rmul) * p.out_scale;
                        if (T.m0 + mrow < p.cout && t < T.Lout) p.y[yb + off[r]] = v;
                    }
                }
        }
    };

    // ---- the (output tile, ci-tile, active tap) step sequence.  Step s consumes the A fragments
    // prefetched during step s-1.  The first tap of every ci-tile also stages x: the raw loads of the
    // NEXT ci-tile (of this output tile, or the first ci-tile of the block's next output tile) are issued
    // ahead of its MFMAs and transformed / written to the other LDS buffer right after them; one barrier
    // per ci-tile.  The staging step is straight-line code on purpose -- no condition sits between the
    // loads and their use (a conditional definition turns into PHI copies placed right behind the loads,
    // i.e. a wait for them before the MFMAs).  At the very end of a block's walk "next" aliases the current
    // tile, so the one redundant prefetch reads valid addresses and lands in the LDS buffer nobody reads.
    // ---- ring pipeline (TAPS > 0, bf16 modes) ---------------------------------------------------------------
    // On the bf16 matrix instruction one tap of a 16-channel tile is 4 MFMAs = 128 pipe cycles per wave (384 in the
    // split mode) -- a sixteenth of the fp32 tap -- so "one tap ahead" no longer covers anything: the L2 round trip of
    // the weight fragments (~700 cycles) and each LDS round trip (~150) are exposed on every tap, and that, not the
    // matrix pipe, LDS bandwidth or instruction count, is where the generic loop spends its time in these modes.
    // With the tap count known at compile time the loop over the taps of a ci-tile unrolls and
    //   * the weight fragments live in a RING with one slot per tap: slot j is consumed by tap j of ci-tile i and
    //     refilled at once with tap j of ci-tile i+1 -- every fragment is requested a whole ci-tile before its use;
    //   * B fragments are read one tap ahead into a second register set (one exposed LDS round trip per ci-tile);
    //   * the raw x loads of ci-tile i+2 are issued while ci-tile i computes and are transformed / written to LDS
    //     during ci-tile i+1: loads return in order, so the wait for a ring slot that was requested a ci-tile ago never
    //     drains the x loads behind it, and they get a whole ci-tile of latency budget instead of one tap.
    // The arithmetic per output element (order of ci-tiles, taps, k-halves; the three split terms) is the generic
    // loop's, so the results are bit-identical to it.
    if constexpr (TAPS > 0 && BF16) {
        static_assert(BF16 && KC == 16 && (HT * HTN == 4 || HT * HTN == 8), "the ring pipeline: bf16 modes, four or eight MFMA tiles per wave");
        constexpr int NT_ = HT * HTN;                              // MFMA tiles (accumulators) per wave
        constexpr int NQ = (SPLIT ? 2 : 1) * HT;
        uint4 ring[TAPS][NQ];
        auto load_ring = [&](int m0, int ci0, int j) {
            constexpr int FB = SPLIT ? 32 : 16;
            const char* base = reinterpret_cast<const char*>(p.wb) + ((long)((j * nt16 + ci0 / 16) * 2) * p.cout_p + m0) * FB;
            const uint4* wp = reinterpret_cast<const uint4*>(base + a_lane);
#pragma unroll
            for (int mi = 0; mi < HT; ++mi) {
                if constexpr (SPLIT) { ring[j][2 * mi] = wp[64 * mi]; ring[j][2 * mi + 1] = wp[64 * mi + 1]; }
                else ring[j][mi] = wp[32 * mi];
            }
        };
        // One tap = 4 MFMAs (12 in the split mode).  The staging transform of the NEXT ci-tile is cut into 8 units (a
        // channel pair x one of the 4 columns a lane carries: affine, Snake / LeakyReLU, validity, bf16 pack, split) and
        // the units of a tap are issued BETWEEN its MFMAs: a wave issues in order, so VALU work placed after a block of
        // MFMAs starts only when the last of them has been accepted by the matrix pipe, and two workgroups per CU that
        // run the same code fall into step (both stage, then both multiply) -- measured on the 128 -> 128, k = 11 launch:
        // the MFMAs, the staging and the epilogue each cost their full time, nothing overlapped.  In the shadow of its
        // own MFMAs a wave's VALU work is free.
        constexpr int NU = (CPW / 2) * NCOL * 4;                   // staging units per ci-tile: (pair, float4 sweep, column)
        unsigned hiw[CPW / 2][NCOL][4], low[CPW / 2][NCOL][4];
        // Per-channel parameters of the ci-tile being staged, formed by the first unit of each channel pair and pinned
        // (left to itself the compiler rebuilds scale / offset / selects in every unit: 80 of the 216 vector instructions
        // of a k = 11 ci-tile, and the per-wave instruction stream is what bounds this kernel -- tools/trace_conv.py).
        // A padded channel gets scale = offset = 0, which stages exact zeros for finite inputs without a per-element
        // select; the column validity is applied once to the packed dword.
        float scx[CPW], ofx[CPW], arx[CPW], ialx[CPW];
        auto xunit = [&](const Tile& T, int ci0, int u) {          // u = (pair * NCOL + sweep) * 4 + column-in-float4, a constant after unrolling
            const int pr = u / (4 * NCOL), k = (u >> 2) % NCOL, e = u & 3;
            if (u % (4 * NCOL) == 0) {
#if K82_RING_SMEM
                // The pair's parameters come through the SCALAR cache (one s_load_dwordx8 + two s_load_dwordx2 per pair):
                // as wave-uniform vector loads they were 6 of the 21 KB a wave pulls through the L1 return path per
                // ci-tile, and that path (TD 87 % busy, profiles/r02_pmc_conv_bf16_ring_k11.txt) is what this kernel
                // waits for.  Read-only data, written by earlier kernels; the ring is only launched for cin % 16 == 0,
                // so both channels of a pair are real and adjacent.
                typedef unsigned uintx8 __attribute__((ext_vector_type(8)));
                typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
                const int cp = ci0 + 2 * (wave + NW * pr);          // == ci0 + chan(2 * pr)
                float sc2[2], of2[2];
                if (p.tr == TR_NORM) {
                    const float4* pn = p.norm + (long)T.b * p.nstride + cp;
                    uintx8 n8;
                    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(n8) : "s"(pn) : "memory");
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float mean = __uint_as_float(n8[4 * h]), rstd = __uint_as_float(n8[4 * h + 1]), g1 = __uint_as_float(n8[4 * h + 2]), beta = __uint_as_float(n8[4 * h + 3]);
                        sc2[h] = rstd * g1;
                        of2[h] = fmaf(-mean, sc2[h], beta);
                    }
                } else {
#pragma unroll
                    for (int h = 0; h < 2; ++h) { sc2[h] = nid.y * nid.z; of2[h] = fmaf(-nid.x, sc2[h], nid.w); }
                }
                if constexpr (SNAKE) {
                    const float* pa = p.alpha + cp;
                    const float* pi = p.inv_alpha + cp;
                    uintx2 a2, i2;
                    asm volatile("s_load_dwordx2 %0, %2, 0x0\n\ts_load_dwordx2 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a2), "=&s"(i2) : "s"(pa), "s"(pi) : "memory");
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        arx[2 * pr + h] = __uint_as_float(a2[h]) * kInvTwoPi;
                        ialx[2 * pr + h] = __uint_as_float(i2[h]);
                        asm volatile("" : "+v"(arx[2 * pr + h]), "+v"(ialx[2 * pr + h]));
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    scx[2 * pr + h] = sc2[h]; ofx[2 * pr + h] = of2[h];
                    asm volatile("" : "+v"(scx[2 * pr + h]), "+v"(ofx[2 * pr + h]));
                }
#else
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int cc = 2 * pr + h;
                    const bool real = ci0 + chan(cc) < p.cin;
                    const float4 nm = (p.tr == TR_NORM) ? nmr[cc] : nid;
                    const float sc = nm.y * nm.z, of = fmaf(-nm.x, sc, nm.w);
                    scx[cc] = real ? sc : 0.f;
                    ofx[cc] = real ? of : 0.f;
                    if constexpr (SNAKE) { arx[cc] = alr[cc] * kInvTwoPi; ialx[cc] = ialr[cc]; asm volatile("" : "+v"(arx[cc])); }
                    asm volatile("" : "+v"(scx[cc]), "+v"(ofx[cc]));
                }
#endif
            }
            float uu[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int cc = 2 * pr + h;
                float v;
                if constexpr (XB) {
                    const unsigned d = e == 0 ? rawb[pr][k].x : e == 1 ? rawb[pr][k].y : e == 2 ? rawb[pr][k].z : rawb[pr][k].w;
                    v = xb_half(d, h);
                } else {
                    v = e == 0 ? raw[cc][k].x : e == 1 ? raw[cc][k].y : e == 2 ? raw[cc][k].z : raw[cc][k].w;
                }
                float w = fmaf(v, scx[cc], ofx[cc]);
                if constexpr (SNAKE) w = snake_fast(w, arx[cc], ialx[cc]);
                else w = fmaxf(w, w * slope);
                uu[h] = w;
            }
            const bool valid = (unsigned)(T.tbase + 4 * (lane + 64 * k) + e) < (unsigned)T.Lin;      // t < 0 wraps: invalid too
            const unsigned hi = pack_bf16(uu[0], uu[1]);
            hiw[pr][k][e] = valid ? hi : 0u;
            if constexpr (SPLIT) {
                const unsigned lo = pack_bf16(uu[0] - __uint_as_float(hi << 16), uu[1] - __uint_as_float(hi & 0xffff0000u));
                low[pr][k][e] = valid ? lo : 0u;
            }
        };
        auto xstore = [&](int pk, float* dst) {                    // pk = pair * NCOL + sweep
            const int pr = pk / NCOL, k = pk % NCOL;
            const int col = 4 * (lane + 64 * k);
            float* d = (col < WP) ? dst + (wave + NW * pr) * WP + col : patch + 4 * lane;
            *reinterpret_cast<uint4*>(d) = make_uint4(hiw[pr][k][0], hiw[pr][k][1], hiw[pr][k][2], hiw[pr][k][3]);
            if constexpr (SPLIT) {
                float* d2 = (col < WP) ? dst + (8 + wave + NW * pr) * WP + col : patch + 4 * lane;
                *reinterpret_cast<uint4*>(d2) = make_uint4(low[pr][k][0], low[pr][k][1], low[pr][k][2], low[pr][k][3]);
            }
        };
        // tap J with staging units [U0, U1) of the next ci-tile threaded between its MFMAs
        auto mma_ring = [&](auto jc, auto u0c, auto u1c, const BFrag (&bf)[HTN], const Tile& T1, int ci1, float* xsn) {
            constexpr int J = decltype(jc)::value, U0 = decltype(u0c)::value, U1 = decltype(u1c)::value;
            constexpr int NM = (SPLIT ? 3 : 1) * NT_;
            static_for<NM>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                constexpr int term = SPLIT ? m / NT_ : 2, mi = (m % NT_) / HTN, ni = (m % NT_) % HTN;
                if constexpr (SPLIT) {     // per accumulator: a_lo b_hi, a_hi b_lo, a_hi b_hi (small terms first), as mma_half
                    const bf16x8 a = __builtin_bit_cast(bf16x8, ring[J][2 * mi + (term == 0 ? 1 : 0)]);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, term == 1 ? bf[ni].l.v : bf[ni].h.v, acc[mi][ni], 0, 0, 0);
                } else {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ring[J][mi]), bf[ni].h.v, acc[mi][ni], 0, 0, 0);
                }
                // units spread evenly over the NM gaps of this tap
                constexpr int lo = U0 + ((U1 - U0) * m + NM - 1) / NM, hi = U0 + ((U1 - U0) * (m + 1) + NM - 1) / NM;
                static_for<hi - lo>([&](auto uc) {
                    constexpr int uq = lo + decltype(uc)::value;
                    xunit(T1, ci1, uq);
                    if constexpr ((uq & 3) == 3) xstore(uq >> 2, xsn);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        Tile cur, nxt;
        int q = next_valid(blockIdx.x, cur);
        if (q < 0) return;
        // prologue: ci-tile 0 into LDS buffer 0, the raw loads of ci-tile 1 in flight, the ring filled for ci-tile 0
        load_tile(cur, 0);
        write_tile(cur, 0, lds);
        load_tile(cur, KC, !K82_RING_SMEM);                       // ntiles >= 2 (host-checked): ci-tile 1 is this tile's
#pragma unroll
        for (int j = 0; j < TAPS; ++j) load_ring(cur.m0, 0, j);
        __syncthreads();
        int g = 0;
        while (true) {
            const int qn = next_valid(q + G, nxt);
            const bool have_next = qn >= 0;
            if (!have_next) nxt = cur;                            // the walk's last prefetches re-read valid addresses
            const float bias_l = p.bias[min(cur.m0 + lane, p.cout_p - 1)];
            zero_acc();
            TR(1);
            for (int it = 0; it < ntiles; ++it) {
                TR(2);
                const float* xsb = lds + (g & 1) * (KC * WP);
                float* xsn = lds + ((g + 1) & 1) * (KC * WP);
                const bool more1 = it + 1 < ntiles, more2 = it + 2 < ntiles;
                Tile S1, S2;                                      // by value (a reference to "cur or nxt" sends both to scratch)
                S1.b = more1 ? cur.b : nxt.b; S1.m0 = more1 ? cur.m0 : nxt.m0; S1.Lin = more1 ? cur.Lin : nxt.Lin; S1.tbase = more1 ? cur.tbase : nxt.tbase;
                S2.b = more2 ? cur.b : nxt.b; S2.Lin = more2 ? cur.Lin : nxt.Lin; S2.tbase = more2 ? cur.tbase : nxt.tbase;
                const int ci1 = more1 ? (it + 1) * KC : 0;
                const int ci2 = more2 ? (it + 2) * KC : (it + 2 - ntiles) * KC;
                const int off0 = cur.shift;
                BFrag B[2][HTN];
#pragma unroll
                for (int ni = 0; ni < HTN; ++ni) read_b(xsb, off0, ni, B[0][ni]);
                // staging units per tap: two while they last (k >= 4: taps 0..3), all 8 over the first two taps for k = 3;
                // the raw loads of ci-tile i+2 go out as soon as the last unit has read the registers they land in
                constexpr int UPT0 = (NU + TAPS - 1) / TAPS, UPT = UPT0 < 2 ? 2 : UPT0, ULAST = (NU + UPT - 1) / UPT - 1;
                static_for<TAPS>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    if constexpr (j + 1 < TAPS) {
#pragma unroll
                        for (int ni = 0; ni < HTN; ++ni) read_b(xsb, (j + 1) * p.dil + off0, ni, B[(j + 1) & 1][ni]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    constexpr int u0 = j * UPT < NU ? j * UPT : NU, u1 = (j + 1) * UPT < NU ? (j + 1) * UPT : NU;
                    mma_ring(jc, std::integral_constant<int, u0>{}, std::integral_constant<int, u1>{}, B[j & 1], S1, ci1, xsn);
                    load_ring(S1.m0, ci1, j);                     // slot j again: tap j of the next ci-tile
                    if constexpr (j == ULAST) load_tile(S2, ci2, !K82_RING_SMEM);
                    __builtin_amdgcn_sched_barrier(0);
                });
                TR(5);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                ++g;
            }
            TR(6);
            epilogue(cur, bias_l);
            TR(7);
            if (!have_next) break;
            cur = nxt;
            q = qn;
        }
        return;
    }

    if constexpr (TAPS == 0) {
    Tile cur, nxt;
    // Tile walk start.  (An XCD-aware renumbering -- the 64 workgroups of one XCD, b % 8, walking consecutive
    // tiles so halo and shared x tiles stay behind one L2 -- measured 1-6 % SLOWER on the same box than this
    // round-robin order, which spreads neighbouring tiles over all eight L2s and memory channels; kept simple.)
    // split-K (EPI_PART): one (tile, K block) per workgroup, blockIdx = tile * nblk + block; the tile walk ends after it
    // (q + G >= total).  it0 / it1: the staged ci-tiles of this workgroup's block.
    constexpr bool PART = EPI == EPI_PART;
    const int tpb = PART ? p.kflush * 16 / KC : ntiles;
    const int it0 = PART ? ((int)blockIdx.x % p.nblk) * tpb : 0;
    const int it1 = PART ? min(ntiles, it0 + tpb) : ntiles;
    int q = next_valid(PART ? (int)blockIdx.x / p.nblk : (int)blockIdx.x, cur);
    if (q < 0) return;
    if (LDS3 && tid == 0) *stepcnt = 0u;
    load_tile(cur, it0 * KC);
    write_tile(cur, it0 * KC, lds);
    load_a(cur.m0, it0 * KC, cur.tap0, a0);
    __syncthreads();
    int g = 0;                                   // ci-tile steps done: buffer g % NBUF holds the current ci-tile
    int bcur = 0, bnxt = 1;
    while (true) {
        const int qn = next_valid(q + G, nxt);
        const bool have_next = qn >= 0;
        if (!have_next) nxt = cur;
        const float bias_l = p.bias[min(cur.m0 + lane, p.cout_p - 1)];   // bias is padded to cout_p; lands during the main loop
        zero_acc();
        TR(1);
        int kleft = KSPLIT ? p.kflush : 0;                 // 16-channel tiles until the next fold
        int tap = cur.tap0;
        auto next_tap = [&](int t) -> int {
            const unsigned mm = (t >= 31) ? 0u : (cur.mask >> (t + 1));
            return mm ? t + 1 + __builtin_ctz(mm) : cur.tap0;
        };
        for (int it = it0; it < it1; ++it) {
            const int ci0 = it * KC;
            const bool more = it + 1 < it1;
            const float* xsb = lds + bcur * (KC * WP);
            float* xsn = lds + bnxt * (KC * WP);
            Tile S;                                           // whose x tile / weights come next (by value:
            S.b = more ? cur.b : nxt.b;                       // a reference to "cur or nxt" sends both to scratch)
            S.m0 = more ? cur.m0 : nxt.m0;
            S.Lin = more ? cur.Lin : nxt.Lin;
            S.tbase = more ? cur.tbase : nxt.tbase;
            S.tap0 = more ? cur.tap0 : nxt.tap0;
            const int sci0 = more ? ci0 + KC : it0 * KC;      // (PART: it0 of the same block -- a valid address, the tile is not used)
            // The taps of one ci-tile ping-pong between the two fragment sets: even steps compute from a0 while
            // the loads of the next step fill a1, odd steps the other way round -- no register rotation per tap
            // (16 v_mov, each of which costs the SIMD ~10 cycles that the partner wave's MFMAs do not cover) and no
            // wait at the end of a step.  Every ci-tile starts from a0, so an ODD tap count ends with one rotation
            // a0 <- a1 (through opaque moves: plain copies are coalesced across the loop and the MFMAs then wait,
            // in-order vmcnt, for the loads issued just before them).
            ASet& X = a0;
            ASet& Y = a1;
            TR(2);
            if constexpr (BF16) read_b(xsb, tap * p.dil + cur.shift, 0, bpre);
            else read_b32(xsb + lhi * WP + nw + l31 + tap * p.dil + cur.shift, 0, bpf);
            {
                {   // first active tap: stage + compute
                    const bool single = cur.nact == 1;
                    const int ntap = single ? S.tap0 : next_tap(tap);
                    load_tile(S, sci0);
                    load_a(single ? S.m0 : cur.m0, single ? sci0 : ci0, ntap, Y);
                    __builtin_amdgcn_sched_barrier(0);     // loads are issued BEFORE the MFMAs (the scheduler sinks them otherwise)
                    mfma_tap(xsb, tap * p.dil + cur.shift, ntap * p.dil + cur.shift, X);    // (single: ntap is a tap of this conv too)
                    __builtin_amdgcn_sched_barrier(0);     // ... and waited for only after the MFMAs are in the pipe
                    TR(3);
#if K82_TRACE
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // the x tile has arrived (4 weight loads behind it)
                    TR(8);
#endif
                    write_tile(S, sci0, xsn);
                    if constexpr (LDS3) {
                        // (asm: through the generic pointer the compiler makes these FLAT operations with a vmcnt(0) wait)
                        if (lane == 0) asm volatile("ds_add_u32 %0, %1" :: "v"((unsigned)(size_t)stepcnt), "v"(1u) : "memory");
                    }
                    TR(4);
                    tap = ntap;
                }
                int j = 1;
                for (; j + 1 < cur.nact; j += 2) {         // two taps per trip: compute from Y, then from X
                    {
                        const int ntap = next_tap(tap);
                        load_a(cur.m0, ci0, ntap, X);
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_tap(xsb, tap * p.dil + cur.shift, ntap * p.dil + cur.shift, Y);
                        __builtin_amdgcn_sched_barrier(0);
                        tap = ntap;
                    }
                    {
                        const bool last = j + 2 == cur.nact;
                        const int ntap = last ? S.tap0 : next_tap(tap);
                        load_a(last ? S.m0 : cur.m0, last ? sci0 : ci0, ntap, Y);
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_tap(xsb, tap * p.dil + cur.shift, ntap * p.dil + cur.shift, X);
                        __builtin_amdgcn_sched_barrier(0);
                        tap = ntap;
                    }
                }
                if (j < cur.nact) {                        // an even tap count leaves one step: the last
                    load_a(S.m0, sci0, S.tap0, X);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_tap(xsb, tap * p.dil + cur.shift, tap * p.dil + cur.shift, Y);
                    __builtin_amdgcn_sched_barrier(0);
                    tap = S.tap0;
                }
                if (cur.nact & 1) {
                    if constexpr (BF16) {
#pragma unroll
                        for (int i = 0; i < (SPLIT ? 2 : 1) * HT; ++i) {
                            asm volatile("v_mov_b32 %0, %1" : "=v"(a0.q[i].x) : "v"(a1.q[i].x));
                            asm volatile("v_mov_b32 %0, %1" : "=v"(a0.q[i].y) : "v"(a1.q[i].y));
                            asm volatile("v_mov_b32 %0, %1" : "=v"(a0.q[i].z) : "v"(a1.q[i].z));
                            asm volatile("v_mov_b32 %0, %1" : "=v"(a0.q[i].w) : "v"(a1.q[i].w));
                        }
                    } else {
#pragma unroll
                        for (int kk = 0; kk < KC / 2; ++kk) {
                            asm volatile("v_mov_b32 %0, %1" : "=v"(a0.f[kk][0]) : "v"(a1.f[kk][0]));
                            if constexpr (HT == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(a0.f[kk][1]) : "v"(a1.f[kk][1]));
                        }
                    }
                }
            }
            if constexpr (KSPLIT) {
                kleft -= KC / 16;
                if (kleft <= 0 && more) { fold_acc(true); kleft = p.kflush; }      // (wave-uniform)
            }
            // raw barrier: LDS traffic only.  __syncthreads() would also drain vmcnt(0), i.e. wait here for the
            // weight fragments just requested for the next ci-tile.
            TR(5);
            ++g;
            if constexpr (LDS3) {
                const unsigned target = (unsigned)(NW * g);
                // (bounded at ~10 ms: every wave reaches every step, so the bound is never met; it only keeps a bug from hanging
                // the GPU.  A wave whose bound DOES run out goes on -- every wave must reach the end of the grid -- but says so
                // in *p.err: the tile it reads next may be unstaged, and the engine fails the call instead of returning samples.)
                // spin_limit and err are read from the kernel-argument segment on the slow path only (see epi_args: values
                // that live in SGPRs across the main loop are spilled to VGPR lanes)
                auto poll = [&]() -> bool {
                    unsigned seen;
                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"((unsigned)(size_t)stepcnt) : "memory");
                    return (unsigned)__builtin_amdgcn_readfirstlane((int)seen) >= target;
                };
                if (!poll()) {
                    typedef const ConvP __attribute__((address_space(4))) * KArg;
                    KArg k = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
                    asm volatile("" : "+s"(k));
                    const int limit = k->spin_limit;
                    bool arrived = false;
                    for (int spin = 1; spin < limit; ++spin) {
                        __builtin_amdgcn_s_sleep(1);
                        if (poll()) { arrived = true; break; }
                    }
                    if (!arrived) {
                        int* e = (int*)k->err;
                        if (e && lane == 0) *e = 2;
                    }
                }
                asm volatile("" ::: "memory");
                bcur = bnxt;
                bnxt = bnxt == NBUF - 1 ? 0 : bnxt + 1;
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                bcur ^= 1; bnxt ^= 1;
            }
        }
        TR(6);
        fold_acc(false);                                   // KSPLIT: the last block joins the others
        epilogue(cur, bias_l);
        TR(7);
        if (!have_next) break;
        cur = nxt;
        q = qn;
    }
    }
}

extern std::atomic<long> g_shape_launches[CONV_NSHAPE];      // conv_mfma.hip
static inline unsigned ncu(const ConvP& p) { return p.cus > 0 ? (unsigned)p.cus : 256u; }
bool launch_ring_bf16(const ConvP& p, int epi, long total, hipStream_t s);      // conv_ring.hip: false = not a ring shape, nothing launched

// the bf16 ring kernels of the Snake convs with t1 as bf16 channel pairs in HBM: the producer (plain epilogue, y_b16) and
// the consumer (residual / running-mean epilogue, x_b16)
template <int WM, int WN, int TAPS>
static bool launch_b16(const ConvP& p, int epi, unsigned grid, hipStream_t s) {
    if (p.y_b16 && !p.x_b16 && epi == EPI_PLAIN)
        hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, 1, EPI_PLAIN, true, PREC_BF16, 16, TAPS, 4, false, false, true>), dim3(grid), dim3(64 * WM * WN), 0, s, p);
    else if (p.x_b16 && !p.y_b16 && epi == EPI_RES)
        hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, 1, EPI_RES, true, PREC_BF16, 16, TAPS, 4, false, true, false>), dim3(grid), dim3(64 * WM * WN), 0, s, p);
    else if (p.x_b16 && !p.y_b16 && epi == EPI_ACC)
        hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, 1, EPI_ACC, true, PREC_BF16, 16, TAPS, 4, false, true, false>), dim3(grid), dim3(64 * WM * WN), 0, s, p);
    else return false;
    return true;
}
template <int WM, int WN, bool SNAKE, int BF16, int HT = 2, int KCT = 16, int TAPS = 0, int HTN = HT>
static void launch_epi(const ConvP& p, int epi, unsigned grid, hipStream_t s) {
    if constexpr (SNAKE && BF16 == PREC_BF16 && TAPS > 0 && HT == 1 && HTN == 4) {
        if (p.x_b16 || p.y_b16) {
            if (!launch_b16<WM, WN, TAPS>(p, epi, grid, s)) fprintf(stderr, "k82: no kernel for this bf16-activation conv (epilogue %d)\n", epi);
            return;
        }
    }
    switch (epi) {
    case EPI_PLAIN:  hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, HT, EPI_PLAIN, SNAKE, BF16, KCT, TAPS, HTN>), dim3(grid), dim3(64 * WM * WN), 0, s, p); break;
    case EPI_RES:    hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, HT, EPI_RES, SNAKE, BF16, KCT, TAPS, HTN>), dim3(grid), dim3(64 * WM * WN), 0, s, p); break;
    case EPI_ACC:    hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, HT, EPI_ACC, SNAKE, BF16, KCT, TAPS, HTN>), dim3(grid), dim3(64 * WM * WN), 0, s, p); break;
    default:
        if constexpr (SNAKE && TAPS > 0) break;                      // no Snake conv has a scalar epilogue (launch_bf16 never asks)
        else hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, HT, EPI_SCALAR, SNAKE, BF16, KCT, TAPS, HTN>), dim3(grid), dim3(64 * WM * WN), 0, s, p);
        break;
    }
}