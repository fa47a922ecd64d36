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
// after that tap's MFMAs are issued, one barrier per ci-tile.  The A operand (weights) goes
// global->VGPR directly (lane l reads w[tap][ci0+2kk+(l>>5)][m+(l&31)], two 128-B segments,
// L2-resident) and is prefetched one whole tap (16 registers) ahead of the MFMAs that consume it,
// across ci-tile boundaries too.  B operand: ds_read_b32 of
// xs[2kk+(l>>5)][n+(l&31)+tap*dil], 32 consecutive floats per half-wave = conflict-free.
// A wave owns a 64(co) x 64(t) tile = 2x2 MFMA tiles = 64 accumulator VGPRs; WM x WN waves per block.
#include "kernels.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

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

template <int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void conv_mfma_kernel(ConvP p) {
    constexpr int BM = 64 * WM, BN = 64 * WN, KC = 16, WP = BN + 68, NW = WM * WN;
    constexpr int CPW = KC / NW;               // channels staged per wave
    constexpr int NCOL = (WP / 4 + 63) / 64;   // float4 sweeps (64 lanes x 4 columns) per channel
    static_assert(KC % NW == 0, "KC must be a multiple of the wave count");
    __shared__ __attribute__((aligned(16))) float xs[2][KC * WP];
    __shared__ __attribute__((aligned(16))) float patches[NW][32 * 36];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // provably wave-uniform: Tile stays in SGPRs
    const int wm = wave / WN, wn = wave % WN;
    const int nw = wn * 64;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int ntiles = p.cin_p / KC;
    const int MT = p.cout_p / BM, NT = (p.Lmax_out + BN - 1) / BN;
    const int total = MT * NT * p.B, G = gridDim.x;
    const unsigned tapall = (p.taps >= 32) ? 0xffffffffu : ((1u << p.taps) - 1u);

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
        T.m0 = mt * BM + wm * 64;
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

    floatx16 acc[2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };

    float4 raw[CPW][NCOL];
    // raw global loads of one ci-tile (no transform yet): wave w takes channels w, w+NW, ...
    auto load_tile = [&](const Tile& T, int ci0) {
        const float* xb = p.x + (long)T.b * p.xbs;
#pragma unroll
        for (int cc = 0; cc < CPW; ++cc) {
            const int ci = ci0 + wave + cc * NW;
            const float* xr = xb + (long)ci * p.xpitch;
            const bool cvalid = ci < p.cin;
#pragma unroll
            for (int k = 0; k < NCOL; ++k) {
                const int col = 4 * (lane + 64 * k);
                const int t = T.tbase + col;
                raw[cc][k] = (cvalid && col < T.W && t >= 0 && t < T.Lin) ? *reinterpret_cast<const float4*>(xr + t)
                                                                          : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    // transform (AdaIN affine, activation) and write to the LDS buffer; zero padding stays zero
    auto write_tile = [&](const Tile& T, int ci0, float* dst) {
#pragma unroll
        for (int cc = 0; cc < CPW; ++cc) {
            const int c = wave + cc * NW;
            const int ci = ci0 + c;
            const bool cvalid = ci < p.cin;
            float4 nm = make_float4(0.f, 1.f, 1.f, 0.f);
            float al = 1.f, ial = 1.f;
            if (cvalid) {
                if (p.tr == TR_NORM) nm = p.norm[(long)T.b * p.nstride + ci];
                if (p.act == ACT_SNAKE) { al = p.alpha[ci]; ial = p.inv_alpha[ci]; }
            }
#pragma unroll
            for (int k = 0; k < NCOL; ++k) {
                const int col = 4 * (lane + 64 * k);
                const int t = T.tbase + col;
                if (col < T.W) {
                    float v[4] = {raw[cc][k].x, raw[cc][k].y, raw[cc][k].z, raw[cc][k].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float u = 0.f;
                        if (cvalid && t + e >= 0 && t + e < T.Lin) {
                            u = v[e];
                            if (p.tr == TR_NORM) u = ((u - nm.x) * nm.y) * nm.z + nm.w;
                            else if (p.tr == TR_SCALE) u = u * p.in_scale;
                            if (p.act == ACT_LEAKY) u = u < 0.f ? u * p.slope : u;
                            else if (p.act == ACT_SNAKE) u = snake(u, al, ial);
                        }
                        v[e] = u;
                    }
                    *reinterpret_cast<float4*>(dst + c * WP + col) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    };

    float a0[KC / 2][2], a1[KC / 2][2];     // A fragments of the tap being consumed / the next one
    auto load_a = [&](const Tile& T, int ci0, int tap, float (&a)[KC / 2][2]) {
        const float* wp = p.w + ((long)(tap * p.cin_p + ci0 + lhi) * p.cout_p) + T.m0 + l31;
#pragma unroll
        for (int kk = 0; kk < KC / 2; ++kk) {
            a[kk][0] = wp[(long)(2 * kk) * p.cout_p];
            a[kk][1] = wp[(long)(2 * kk) * p.cout_p + 32];
        }
    };
    auto mfma_tap = [&](const float* xsb, int off, const float (&a)[KC / 2][2]) {
        const float* xsp = xsb + lhi * WP + nw + l31 + off;
#pragma unroll
        for (int kk = 0; kk < KC / 2; ++kk) {
            const float b0 = xsp[2 * kk * WP];
            const float b1 = xsp[2 * kk * WP + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][0], b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][0], b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][1], b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][1], b1, acc[1][1], 0, 0, 0);
        }
    };

    // ---- epilogue of one output tile.  D layout: column j = lane&31 (t), row i = (r&3) + 8*(r>>2)
    // + 4*(lane>>5) (co).  Plain convs send each 32x32 accumulator tile through a per-wave LDS patch
    // ([32][36] floats) and read it back as float4 along t: 4 dwordx4 stores of 8 rows x 128 B per
    // tile instead of 16 dword stores, and float4 residual / running-mean traffic.  The stores are
    // asynchronous: the wave goes straight on to the MFMAs of its next tile.
    auto epilogue = [&](const Tile& T) {
        const long yb = (long)T.b * p.ybs, rb = (long)T.b * p.rbs, ab = (long)T.b * p.abs_;
        const bool wide = p.up_s == 0 && p.res_shift == 0;
        if (wide) {
            float* patch = patches[wave];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lhi) * 36 + l31] = acc[mi][ni][r];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int i = (lane >> 3) + 8 * q, c4 = (lane & 7) * 4;
                        const float4 v4 = *reinterpret_cast<const float4*>(patch + i * 36 + c4);
                        const int m = T.m0 + mi * 32 + i;
                        const int t = T.n0 + nw + ni * 32 + c4;
                        if (m < p.cout && t < T.Lout && !((p.dbg & 1) && v4.x != 12345.678f)) {
                            float v[4] = {v4.x, v4.y, v4.z, v4.w};
                            const float bs = p.bias ? p.bias[m] : 0.f;
                            const bool full = t + 3 < T.Lout;
                            float rs[4] = {0.f, 0.f, 0.f, 0.f}, av[4] = {0.f, 0.f, 0.f, 0.f};
                            if (p.res) {
                                const float* rp = p.res + rb + (long)m * p.rpitch + t;
                                if (full) { const float4 r4 = *reinterpret_cast<const float4*>(rp); rs[0] = r4.x; rs[1] = r4.y; rs[2] = r4.z; rs[3] = r4.w; }
                                else for (int e = 0; e < 4; ++e) if (t + e < T.Lout) rs[e] = rp[e];
                            }
                            float* apn = p.accbuf ? p.accbuf + ab + (long)m * p.apitch + t : nullptr;
                            if (apn && !p.acc_init) {
                                if (full) { const float4 a4 = *reinterpret_cast<const float4*>(apn); av[0] = a4.x; av[1] = a4.y; av[2] = a4.z; av[3] = a4.w; }
                                else for (int e = 0; e < 4; ++e) if (t + e < T.Lout) av[e] = apn[e];
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float u = v[e] + bs;
                                if (p.out_act == OACT_GELU_NEW) u = gelu_new(u);
                                u = (u + rs[e]) * p.out_scale;
                                v[e] = u;
                                av[e] = av[e] + u * p.acc_scale;
                            }
                            if (p.y) {
                                float* yp = p.y + yb + (long)m * p.ypitch + t;
                                if (full) *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
                                else for (int e = 0; e < 4; ++e) if (t + e < T.Lout) yp[e] = v[e];
                            }
                            if (apn) {
                                if (full) *reinterpret_cast<float4*>(apn) = make_float4(av[0], av[1], av[2], av[3]);
                                else for (int e = 0; e < 4; ++e) if (t + e < T.Lout) apn[e] = av[e];
                            }
                        }
                    }
                }
            return;
        }
        const int upmask = (1 << p.up_cout_shift) - 1;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = T.m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    const int t = T.n0 + nw + ni * 32 + l31;
                    if (m < p.cout && t < T.Lout && !((p.dbg & 1) && acc[mi][ni][r] != 12345.678f)) {
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
    };

    // ---- the flattened (output tile, ci-tile, active tap) step sequence.  Step s consumes the A
    // fragments prefetched during step s-1; the raw x loads of the NEXT ci-tile (of this output
    // tile, or the first ci-tile of the block's next output tile) are issued at the first tap of a
    // ci-tile and transformed / written to the other LDS buffer right after that tap's MFMAs are
    // issued; one barrier per ci-tile.  Nothing but the very first tile of a block starts cold.
    Tile cur, nxt;
    int q = next_valid(blockIdx.x, cur);
    if (q < 0) return;
    load_tile(cur, 0);
    write_tile(cur, 0, xs[0]);
    load_a(cur, 0, cur.tap0, a0);
    __syncthreads();
    int g = 0;                                   // parity of the LDS buffer holding the current ci-tile
    while (true) {
        const int qn = next_valid(q + G, nxt);
        const bool have_next = qn >= 0;
        zero_acc();
        int tap = cur.tap0;
        for (int it = 0; it < ntiles; ++it) {
            const int ci0 = it * KC;
            const bool more = it + 1 < ntiles;
            const float* xsb = xs[g & 1];
            float* xsn = xs[(g + 1) & 1];
            for (int j = 0; j < cur.nact; ++j) {
                const bool last_in_tile = (j == cur.nact - 1);
                // one call site each for load_tile / load_a / write_tile (they inline large bodies)
                const bool stage = j == 0 && (more || have_next) && !(p.dbg & 2);
                const Tile& S = more ? cur : nxt;
                const int sci0 = more ? ci0 + KC : 0;
                if (stage) load_tile(S, sci0);
                const bool intile = !last_in_tile || more;
                const Tile& AT = intile ? cur : nxt;
                int ntap = nxt.tap0;
                if (intile) {
                    const unsigned mm = (tap >= 31) ? 0u : (cur.mask >> (tap + 1));
                    ntap = mm ? tap + 1 + __builtin_ctz(mm) : cur.tap0;
                }
                const int aci0 = !last_in_tile ? ci0 : (more ? ci0 + KC : 0);
                if (intile || have_next) load_a(AT, aci0, ntap, a1);
                mfma_tap(xsb, tap * p.dil + cur.shift, a0);
                if (stage) write_tile(S, sci0, xsn);
                tap = ntap;
#pragma unroll
                for (int kk = 0; kk < KC / 2; ++kk) { a0[kk][0] = a1[kk][0]; a0[kk][1] = a1[kk][1]; }
            }
            __syncthreads();
            ++g;
        }
        epilogue(cur);
        if (!have_next) break;
        cur = nxt;
        q = qn;
    }
}

// Persistent launch: at most CUS x BLOCKS_PER_CU workgroups walk the output tiles.
void launch_conv(const ConvP& p, hipStream_t s) {
    if (p.Lmax_out <= 0 || p.B <= 0) return;
    const int kMaxBlocks = 256 * 2;
    if (p.cout_p % 128 == 0) {
        const long total = (long)((p.Lmax_out + 127) / 128) * (p.cout_p / 128) * p.B;
        hipLaunchKernelGGL((conv_mfma_kernel<2, 2>), dim3((unsigned)(total < kMaxBlocks ? total : kMaxBlocks)), dim3(256), 0, s, p);
    } else {
        const long total = (long)((p.Lmax_out + 255) / 256) * (p.cout_p / 64) * p.B;
        hipLaunchKernelGGL((conv_mfma_kernel<1, 4>), dim3((unsigned)(total < kMaxBlocks ? total : kMaxBlocks)), dim3(256), 0, s, p);
    }
}

// Algorithmic FLOPs: 2 * cout * cin * (non-zero taps) * sum_b Lout_b, un-padded channel counts.
double conv_flops(const ConvP& p, const int* host_lens) {
    double L = 0;
    for (int b = 0; b < p.B; ++b) L += host_lens ? (double)p.lout.lmul * host_lens[b] + p.lout.ladd : p.lout.ladd;
    const double taps = p.up_s ? 2.0 : (double)p.taps;
    return 2.0 * p.cout * (double)p.cin * taps * L;
}
