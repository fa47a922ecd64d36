// Implicit-GEMM Conv1d for gfx950: launch_conv and the fp32 / bf16x3 / generic-bf16 instantiations of the kernel template in
// conv_mfma_kernel.h (design notes there); the bf16 ring pipeline's instantiations live in conv_ring.hip.
#include "conv_mfma_kernel.h"

#define REQUIRE_SNAKE_FREE(p) do { if ((p).act == ACT_SNAKE) { fprintf(stderr, "k82: Snake conv with <= 32 output channels is not built\n"); return; } } while (0)
// split-K, second step: y = epilogue(b_0 + b_1 + ... in block order) -- the operations of the conv kernel's own epilogue on
// the value its blocked summation (KSPLIT) would have left in the accumulator, so the two paths give the same bits
__global__ __launch_bounds__(256) void splitk_reduce_kernel(ConvP p) {
    const int b = blockIdx.z, m = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    if (t >= p.lout.len(b)) return;
    const float* pp = p.part + ((long)b * p.cout_p + m) * p.ppitch + t;
    const long zs = (long)p.B * p.cout_p * p.ppitch;
    float r = pp[0];
    for (int z = 1; z < p.nblk; ++z) r += pp[z * zs];
    float u = r + p.bias[m];
    if (p.out_act == OACT_GELU_NEW) u = gelu_new(u);
    if (p.res) u += p.res[(long)b * p.rbs + (long)m * p.rpitch + t];
    u *= p.out_scale;
    p.y[(long)b * p.ybs + (long)m * p.ypitch + t] = u;
}
// The same for launches whose epilogue also leaves instance-norm partials (ConvP::stat) or that are wide enough for 16-byte
// accesses: a thread owns four consecutive columns, eight lanes a 32-column span -- the lane layout of the conv kernel's wide
// epilogues, so the partials (4 sequential adds per lane, then sum8) are the bits that epilogue writes.
__global__ __launch_bounds__(256) void splitk_reduce4_kernel(ConvP p) {
    const int b = blockIdx.z, m = blockIdx.y * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int t = (blockIdx.x * 64 + lane) * 4, L = p.lout.len(b);
    if (m >= p.cout) return;                                    // wave-uniform
    const bool live = t < L;
    const int tc = live ? t : 0;
    const float* pp = p.part + ((long)b * p.cout_p + m) * p.ppitch + tc;
    const long zs = (long)p.B * p.cout_p * p.ppitch;
    float4 r = *reinterpret_cast<const float4*>(pp);
    for (int z = 1; z < p.nblk; ++z) {
        const float4 a = *reinterpret_cast<const float4*>(pp + z * zs);
        r.x += a.x; r.y += a.y; r.z += a.z; r.w += a.w;
    }
    const float bs = p.bias[m];
    float v[4] = {r.x, r.y, r.z, r.w};
    float rs[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.res) { const float4 u = *reinterpret_cast<const float4*>(p.res + (long)b * p.rbs + (long)m * p.rpitch + tc); rs[0] = u.x; rs[1] = u.y; rs[2] = u.z; rs[3] = u.w; }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float u = v[e] + bs;
        if (p.out_act == OACT_GELU_NEW) u = gelu_new(u);
        if (p.res) u += rs[e];
        u *= p.out_scale;
        v[e] = u;
    }
    if (p.stat) {                                               // uniform
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float u = (live && t + e < L) ? v[e] : 0.f;
            s1 += u;
            s2 = fmaf(u, u, s2);
        }
        s1 = sum8(s1);
        s2 = sum8(s2);
        if ((lane & 7) == 0 && live) p.stat[(long)b * p.sbs + (long)m * p.snt + (t >> 5)] = make_float2(s1, s2);
    }
    if (live) *reinterpret_cast<float4*>(p.y + (long)b * p.ybs + (long)m * p.ypitch + t) = make_float4(v[0], v[1], v[2], v[3]);
}
// K blocks a small fp32 launch is split into (0: not split).  The conditions are those under which launch_conv would take the
// 64 x 64-tile kernels WITH blocked summation, plus: few enough tiles that the chip is mostly idle, a plain or residual epilogue.
int conv_splitk_blocks(const ConvP& p) {
    if (p.wb || p.cout_p % 128 != 0 || p.act == ACT_SNAKE || p.kflush <= 0 || p.tapmask || p.up_s || p.res_shift || p.accbuf || !p.y) return 0;
    if (p.cin_p <= 16 * p.kflush) return 0;
    const unsigned cus = p.cus > 0 ? (unsigned)p.cus : 256u;
    const long total = (long)((p.Lmax_out + 127) / 128) * (p.cout_p / 128) * p.B;
    static const long small_env = [] { const char* e = getenv("K82_SMALL_GRID"); return e ? atol(e) : -1L; }();
    const long small_grid = small_env >= 0 ? small_env : (long)K82_SMALL_GRID * cus / 256;
    if (!(total < small_grid || p.Lmax_out <= 64)) return 0;
    const long t64 = (long)((p.Lmax_out + 63) / 64) * (p.cout_p / 64) * p.B;
    if (t64 > (long)cus) return 0;
    const int kc16 = (p.taps == 1) ? 2 : 1;                    // the 1-tap small-launch kernel stages 32 channels per tile
    if ((p.kflush % kc16) != 0) return 0;
    return (p.cin_p / 16 + p.kflush - 1) / p.kflush;
}
std::atomic<long> g_shape_launches[CONV_NSHAPE];
long conv_shape_launches(int shape) { return shape >= 0 && shape < CONV_NSHAPE ? g_shape_launches[shape].load() : -1; }

template <int WM, int WN, int HT = 2, int KCT = 16, int HTN = HT>
static void launch_shape(const ConvP& p, int epi, unsigned grid, hipStream_t s) {
    if constexpr (HTN != HT) {          // the 32-row shape: no Snake conv has <= 32 output channels
        launch_epi<WM, WN, false, PREC_F32, HT, KCT, 0, HTN>(p, epi, grid, s);
        return;
    }
    if (p.act == ACT_SNAKE) launch_epi<WM, WN, true, PREC_F32, HT, KCT>(p, epi, grid, s);
    else if (WM >= 2 && p.kflush > 0 && epi != EPI_ACC && p.cin_p > 16 * p.kflush) {
        // blocked summation (KSPLIT): no Snake conv and no running-mean epilogue asks for it; the 1 x 4 shape (<= 64 output
        // channels: the duration and F0 / N heads, K <= 512) is not built with it
        if constexpr (WM >= 2) {
            ++g_shape_launches[CONV_SHAPE_KSPLIT];
            switch (epi) {
            case EPI_PLAIN: hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, HT, EPI_PLAIN, false, PREC_F32, KCT, 0, HT, true>), dim3(grid), dim3(64 * WM * WN), 0, s, p); break;
            case EPI_RES:   hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, HT, EPI_RES, false, PREC_F32, KCT, 0, HT, true>), dim3(grid), dim3(64 * WM * WN), 0, s, p); break;
            default:        hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, HT, EPI_SCALAR, false, PREC_F32, KCT, 0, HT, true>), dim3(grid), dim3(64 * WM * WN), 0, s, p); break;
            }
        }
    } else launch_epi<WM, WN, false, PREC_F32, HT, KCT>(p, epi, grid, s);
}
static void launch_bf16(const ConvP& p, int epi, long total, hipStream_t s) {       // 2x2 shape only
    // plain bf16 only: in the split mode the ring (8 registers per tap) and the doubled B fragments spill, and the build
    // measured slower than the two-set loop (r02_ab_ring4.log)
    if (!p.wb_split && launch_ring_bf16(p, epi, total, s)) return;
    const unsigned cus = ncu(p);
    const long cap = (long)cus * (p.wb_split ? K82_OCC_SPLIT : K82_OCC_BF16);
    const unsigned grid = (unsigned)(total < cap ? total : cap);
    if (p.wb_split) {
        // 4 x 1 waves of 32 x 128 in the two-set loop (2 x 2 waves of 64 x 64: 2.1 % slower, profiles/r02_ab_split_32x128.txt)
        // (4 x 2 waves on 128 x 256 tiles for large launches, as the plain-bf16 ring does: 4474 vs 4475 audio-s/s with the
        // barrier, 4615 vs 4662 with the step counter; not kept)
#if K82_SPLIT_BM256
        if (p.cout_p % 256 == 0 && total >= 4 * (long)cus && !(p.dbg & 2)) {      // as fp32: 256 x 128 tiles (8 x 1 waves), one workgroup per CU
            if (p.act == ACT_SNAKE) launch_epi<8, 1, true, PREC_SPLIT, 1, 16, 0, 4>(p, epi, cus, s);
            else launch_epi<8, 1, false, PREC_SPLIT, 1, 16, 0, 4>(p, epi, cus, s);
            ++g_shape_launches[CONV_SHAPE_BM256]; ++g_shape_launches[CONV_SHAPE_LDS3];
            return;
        }
#endif
        if (p.act == ACT_SNAKE) launch_epi<4, 1, true, PREC_SPLIT, 1, 16, 0, 4>(p, epi, grid, s);
        else launch_epi<4, 1, false, PREC_SPLIT, 1, 16, 0, 4>(p, epi, grid, s);
        ++g_shape_launches[CONV_SHAPE_LDS3];
    } else {
        if (p.act == ACT_SNAKE) launch_epi<2, 2, true, PREC_BF16>(p, epi, grid, s);
        else launch_epi<2, 2, false, PREC_BF16>(p, epi, grid, s);
    }
}

// Persistent launch: at most CUS x BLOCKS_PER_CU workgroups walk the output tiles.
// Host-checked operand contract (engine.hip run_conv): bias != nullptr and padded to cout_p; accbuf implies
// res and a wide epilogue; y != nullptr unless accbuf; all pitches % 4 == 0 and bases 16-byte aligned.
#if K82_TRACE
static void launch_conv_inner(const ConvP& p, hipStream_t s);
void launch_conv(const ConvP& p0, hipStream_t s) {
    // trace build: every launch stamps into one device buffer; the last launch's stamps are written to $K82_TRACE_FILE
    static unsigned long long* buf = nullptr;
    const size_t bytes = (size_t)768 * 4 * 1024 * 8;
    if (!buf && hipMalloc(&buf, bytes) != hipSuccess) abort();
    hipMemsetAsync(buf, 0, bytes, s);
    ConvP p = p0;
    p.trace = buf;
    launch_conv_inner(p, s);
    if (const char* f = getenv("K82_TRACE_FILE")) {
        hipStreamSynchronize(s);
        void* h = malloc(bytes);
        hipMemcpy(h, buf, bytes, hipMemcpyDeviceToHost);
        if (FILE* fp = fopen(f, "wb")) { fwrite(h, 1, bytes, fp); fclose(fp); }
        free(h);
    }
}
static void launch_conv_inner(const ConvP& p, hipStream_t s) {
#else
void launch_conv(const ConvP& p, hipStream_t s) {
#endif
    if (p.Lmax_out <= 0 || p.B <= 0) return;
    const unsigned cus = ncu(p);
    const long kMaxBlocks = (long)cus * K82_OCC_F32;
    static const long small_env = [] { const char* e = getenv("K82_SMALL_GRID"); return e ? atol(e) : -1L; }();   // env: tuning only
    const long small_grid = small_env >= 0 ? small_env : (long)K82_SMALL_GRID * cus / 256;
    const int epi = (p.up_s || p.res_shift) ? EPI_SCALAR : p.accbuf ? EPI_ACC : p.res ? EPI_RES : EPI_PLAIN;
    if (p.cout_p % 128 == 0) {
        const long total = (long)((p.Lmax_out + 127) / 128) * (p.cout_p / 128) * p.B;
        if (p.wb) launch_bf16(p, epi, total, s);
        else if (total < small_grid || p.Lmax_out <= 64) {
            // (Lmax_out <= 64: token-rate layers of short sentences -- a 128-column tile would be at least half padding;
            // the book workloads spend a fifth of their time in phase 1 on exactly these.  Same bits either way.)
            // too few 128 x 128 tiles to occupy the chip: 64 x 64 tiles (32 x 32 per wave), four times as many
            const long t64 = (long)((p.Lmax_out + 63) / 64) * (p.cout_p / 64) * p.B;
            // workgroups per CU of the 64 x 64-tile kernels: they hold 16 accumulator registers and 44 KB of LDS, so more than
            // two are resident per CU, and a launch of slightly more tiles than 2 x CUs does not pay for a second, almost
            // empty round of the tile walk (K82_SMALL_OCC: tuning only)
            static const long small_occ = [] { const char* e = getenv("K82_SMALL_OCC"); return e ? atol(e) : (long)K82_SMALL_OCC; }();
            const long cap64 = (long)cus * (small_occ > 0 ? small_occ : 2);
            const unsigned grid = (unsigned)(t64 < cap64 ? t64 : cap64);
            if (p.part && p.nblk >= 2) {                          // split-K: (tile, K block) per workgroup, then the ordered sum + epilogue
                const unsigned g2 = (unsigned)(t64 * p.nblk);
                if (p.taps == 1) hipLaunchKernelGGL((conv_mfma_kernel<2, 2, 1, EPI_PART, false, PREC_F32, 32>), dim3(g2), dim3(256), 0, s, p);
                else hipLaunchKernelGGL((conv_mfma_kernel<2, 2, 1, EPI_PART, false, PREC_F32, 16>), dim3(g2), dim3(256), 0, s, p);
                if (p.stat) hipLaunchKernelGGL(splitk_reduce4_kernel, dim3((p.Lmax_out + 255) / 256, (p.cout + 3) / 4, p.B), dim3(256), 0, s, p);
                else hipLaunchKernelGGL(splitk_reduce_kernel, dim3((p.Lmax_out + 255) / 256, p.cout, p.B), dim3(256), 0, s, p);
                ++g_shape_launches[CONV_SHAPE_SPLITK];
                return;
            }
            if (p.taps == 1 && (epi == EPI_PLAIN || epi == EPI_RES)) launch_shape<2, 2, 1, 32>(p, epi, grid, s);
            else launch_shape<2, 2, 1>(p, epi, grid, s);
            ++g_shape_launches[CONV_SHAPE_LDS3];
        }
#if K82_F32_BM256
        // >= 256 output channels, large launches: 4 x 2 waves on 256 x 128 tiles, one workgroup per CU -- the staged x tile
        // (and its vector work, which excludes the fp32 MFMAs) is shared by twice the rows
        else if (p.cout_p % 256 == 0 && total >= 4 * (long)cus && !(p.dbg & 2)) {
            launch_shape<4, 2>(p, epi, cus, s);
            ++g_shape_launches[CONV_SHAPE_BM256]; ++g_shape_launches[CONV_SHAPE_LDS3];
        }
#endif
        else { launch_shape<2, 2>(p, epi, (unsigned)(total < kMaxBlocks ? total : kMaxBlocks), s); ++g_shape_launches[CONV_SHAPE_LDS3]; }
    } else if (p.cout_p == 32) {
        // <= 32 output channels (conv_post: 22; the F0 / N heads: 1): 1 x 4 waves of 32 x (32 K82_N32_HTN) -- half the MFMAs
        // of the 64-row shape for the same staged x tile (22 useful rows of 32 instead of 22 of 64)
        constexpr int BN32 = 4 * 32 * K82_N32_HTN;
        const long total = (long)((p.Lmax_out + BN32 - 1) / BN32) * p.B;
        const long cap = (long)cus * (K82_N32_HTN > 2 ? 1 : K82_OCC_F32);
        REQUIRE_SNAKE_FREE(p);
        launch_shape<1, 4, 1, 16, K82_N32_HTN>(p, epi, (unsigned)(total < cap ? total : cap), s);
    } else {
        const long total = (long)((p.Lmax_out + 255) / 256) * (p.cout_p / 64) * p.B;
        launch_shape<1, 4>(p, epi, (unsigned)(total < kMaxBlocks ? total : kMaxBlocks), s);
    }
}

// Algorithmic FLOPs: 2 * cout * cin * (non-zero taps) * sum_b Lout_b, un-padded channel counts.
double conv_flops(const ConvP& p, const int* host_lens) {
    double L = 0;
    for (int b = 0; b < p.B; ++b) L += host_lens ? (double)p.lout.lmul * host_lens[b] + p.lout.ladd : p.lout.ladd;
    const double taps = (p.up_s || p.tapmask) ? 2.0 : (double)p.taps;      // transposed-conv phase rows: 2 of the 3 taps are non-zero
    return 2.0 * p.cout * (double)p.cin * taps * L;
}
