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
// A operand (weights) goes global->VGPR directly: lane l reads w[tap][ci0+2kk+(l>>5)][m+(l&31)],
// two 128-B segments, L2-resident.  B operand: ds_read_b32 of xs[2kk+(l>>5)][n+(l&31)+tap*dil],
// 32 consecutive floats per half-wave = conflict-free.
// A wave owns a 64(co) x 64(t) tile = 2x2 MFMA tiles = 64 accumulator VGPRs; WM x WN waves per block.
#include "kernels.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float gelu_new(float x) {
    const float k = 0.7978845608028654f;   // sqrt(2/pi)
    return 0.5f * x * (1.0f + tanhf(k * (x + 0.044715f * x * x * x)));
}

template <int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void conv_mfma_kernel(ConvP p) {
    constexpr int BM = 64 * WM, BN = 64 * WN, KC = 16, WP = BN + 64, NW = WM * WN;
    __shared__ float xs[KC * WP];
    const int b = blockIdx.z;
    const int Lin = p.lin.len(b), Lout = p.lout.len(b);
    const int n0 = blockIdx.x * BN;
    if (n0 >= Lout) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.y * BM + wm * 64, nw = wn * 64;
    const int l31 = lane & 31, lhi = lane >> 5;
    const unsigned mask = p.tapmask ? p.tapmask[m0 >> 6] : 0xffffffffu;

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float* xb = p.x + (long)b * p.xbs;
    const int W = BN + (p.taps - 1) * p.dil;

    for (int ci0 = 0; ci0 < p.cin_p; ci0 += KC) {
        __syncthreads();
        // ---- stage the transformed x tile: wave w takes channels w, w+NW, ...; lanes sweep t
        for (int c = wave; c < KC; c += NW) {
            const int ci = ci0 + c;
            const bool cvalid = ci < p.cin;
            float4 nm = make_float4(0.f, 1.f, 1.f, 0.f);
            float al = 1.f, ial = 1.f;
            if (cvalid) {
                if (p.tr == TR_NORM) nm = p.norm[(long)b * p.nstride + ci];
                if (p.act == ACT_SNAKE) { al = p.alpha[ci]; ial = p.inv_alpha[ci]; }
            }
            const float* xr = xb + (long)ci * p.xpitch;
            for (int col = lane; col < W; col += 64) {
                const int t = n0 - p.pad + col;
                float v = 0.f;
                if (cvalid && t >= 0 && t < Lin) {
                    v = xr[t];
                    if (p.tr == TR_NORM) v = ((v - nm.x) * nm.y) * nm.z + nm.w;
                    else if (p.tr == TR_SCALE) v = v * p.in_scale;
                    if (p.act == ACT_LEAKY) v = v < 0.f ? v * p.slope : v;
                    else if (p.act == ACT_SNAKE) { const float sn = sinf(al * v); v = v + ial * (sn * sn); }
                }
                xs[c * WP + col] = v;
            }
        }
        __syncthreads();
        // ---- MFMA over (tap, ci) of this ci-tile
        for (int tap = 0; tap < p.taps; ++tap) {
            if (!((mask >> tap) & 1u)) continue;
            const float* wp = p.w + ((long)(tap * p.cin_p + ci0 + lhi) * p.cout_p) + m0 + l31;
            const float* xsp = xs + lhi * WP + nw + l31 + tap * p.dil;
#pragma unroll
            for (int kk = 0; kk < KC / 2; ++kk) {
                const float a0 = wp[(long)(2 * kk) * p.cout_p];
                const float a1 = wp[(long)(2 * kk) * p.cout_p + 32];
                const float b0 = xsp[2 * kk * WP];
                const float b1 = xsp[2 * kk * WP + 32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
    }

    // ---- epilogue.  D layout: column j = lane&31 (t), row i = (r&3) + 8*(r>>2) + 4*(lane>>5) (co)
    const long yb = (long)b * p.ybs, rb = (long)b * p.rbs, ab = (long)b * p.abs_;
    const int upmask = (1 << p.up_cout_shift) - 1;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const int t = n0 + nw + ni * 32 + l31;
                if (m < p.cout && t < Lout) {
                    float v = acc[mi][ni][r];
                    if (p.bias) v += p.bias[m];
                    if (p.out_act == OACT_GELU_NEW) v = gelu_new(v);
                    int row = m, col = t;
                    if (p.up_s) { row = m & upmask; col = p.up_s * t + (m >> p.up_cout_shift) + p.up_off; }
                    if (p.res) v += p.res[rb + (long)row * p.rpitch + (col >> p.res_shift)];
                    v *= p.out_scale;
                    if (p.y) p.y[yb + (long)row * p.ypitch + col] = v;
                    if (p.accbuf) {
                        float* a = p.accbuf + ab + (long)row * p.apitch + col;
                        *a = (p.acc_init ? 0.f : *a) + v * p.acc_scale;
                    }
                }
            }
}

void launch_conv(const ConvP& p, hipStream_t s) {
    if (p.Lmax_out <= 0 || p.B <= 0) return;
    if (p.cout_p % 128 == 0) {
        dim3 grid((p.Lmax_out + 127) / 128, p.cout_p / 128, p.B);
        hipLaunchKernelGGL((conv_mfma_kernel<2, 2>), grid, dim3(256), 0, s, p);
    } else {
        dim3 grid((p.Lmax_out + 255) / 256, p.cout_p / 64, p.B);
        hipLaunchKernelGGL((conv_mfma_kernel<1, 4>), grid, dim3(256), 0, s, p);
    }
}

// Algorithmic FLOPs: 2 * cout * cin * (non-zero taps) * sum_b Lout_b, un-padded channel counts.
double conv_flops(const ConvP& p, const int* host_lens) {
    double L = 0;
    for (int b = 0; b < p.B; ++b) L += host_lens ? (double)p.lout.lmul * host_lens[b] + p.lout.ladd : p.lout.ladd;
    const double taps = p.up_s ? 2.0 : (double)p.taps;
    return 2.0 * p.cout * (double)p.cin * taps * L;
}
