// The bf16 ring pipeline's instantiations of the conv kernel template (conv_mfma_kernel.h): its own translation unit, so that it
// compiles beside conv_mfma.hip.
#include "conv_mfma_kernel.h"

bool launch_ring_bf16(const ConvP& p, int epi, long total, hipStream_t s) {
    constexpr int PREC = PREC_BF16;
    // the ring pipeline: all taps active, a compile-time tap count, at least two ci-tiles (see the kernel)
    // (a tap mask -- the transposed convs' phase rows, two of three taps non-zero per 64-row tile -- is simply ignored: the masked
    // taps' weights are exact zeros, so the ring's all-taps loop adds 0 x finite = 0 to the accumulators: same values as the
    // generic loop that skips them, at 1.5x the MFMAs of a kernel that runs 2.5x as fast on these launches)
    if ((p.dbg & 1) || (p.tapmask && (p.taps != 3 || p.act == ACT_SNAKE)) || p.cin_p < 32 || (p.cin % 16 != 0 && !(p.cin_pad_ok && p.tr == TR_NORM && K82_RING_SMEM))) return false;   // (cin % 16: the staging reads its parameters per channel PAIR)   // dbg = 1: K82_RING=0 (A/B switch of the engine)
    const bool snake = p.act == ACT_SNAKE;
    if (snake && epi == EPI_SCALAR) return false;
    // (Measured and not kept: 4 x 1 waves of 32 x 256 at one workgroup per CU -- weight traffic halves again, but with one
    // wave per SIMD the k = 11 launch went from 0.475 to 0.556 ms; the kernel template still supports HT * HTN = 8.)
    // Large launches: 4 x 2 waves, 128 x 256 tiles, ONE workgroup per CU -- the two waves that own the same 32 weight rows
    // run in step (one barrier per ci-tile), so the second one's fragment loads hit the L1 lines the first one just
    // missed on, and the x halo is shared by twice the columns: +2.9 % on configs[1] in bf16 (6972 -> 7173 audio-s/s,
    // profiles/r02_ab_ring_wn2_bf16.txt); single launches of the shapes lose 3-7 % at k >= 7 and the book workloads do
    // not move, so launches with fewer than two rounds of 256-column tiles keep 4 x 1 waves and two workgroups per CU.
    const unsigned cus = ncu(p);
#if K82_RING_BM256
    if (p.cout_p % 256 == 0 && total >= 4 * (long)cus && !(p.dbg & 2)) {    // 256 x 128 tiles, 8 x 1 waves, one workgroup per CU
        bool done = true;
        if (p.taps == 3) {
            if (snake) launch_epi<8, 1, true, PREC, 1, 16, 3, 4>(p, epi, cus, s);
            else launch_epi<8, 1, false, PREC, 1, 16, 3, 4>(p, epi, cus, s);
        } else if (snake && p.taps == 7) launch_epi<8, 1, true, PREC, 1, 16, 7, 4>(p, epi, cus, s);
        else if (snake && p.taps == 11) launch_epi<8, 1, true, PREC, 1, 16, 11, 4>(p, epi, cus, s);
        else done = false;
        if (done) { ++g_shape_launches[CONV_SHAPE_BM256]; ++g_shape_launches[CONV_SHAPE_RING]; return true; }
    }
#endif
    const long t256 = (long)((p.Lmax_out + 255) / 256) * (p.cout_p / 128) * p.B;
    if (K82_RING_WN == 2 && t256 >= 2 * (long)cus) {
        const unsigned grid = cus;
        if (p.taps == 3) {
            if (snake) launch_epi<4, 2, true, PREC, 1, 16, 3, 4>(p, epi, grid, s);
            else launch_epi<4, 2, false, PREC, 1, 16, 3, 4>(p, epi, grid, s);
        } else if (!snake) return false;
        else if (p.taps == 7) launch_epi<4, 2, true, PREC, 1, 16, 7, 4>(p, epi, grid, s);
        else if (p.taps == 11) launch_epi<4, 2, true, PREC, 1, 16, 11, 4>(p, epi, grid, s);
        else return false;
        ++g_shape_launches[CONV_SHAPE_RING_WN2]; ++g_shape_launches[CONV_SHAPE_RING];
        return true;
    }
    const unsigned grid = (unsigned)(total < (long)cus * K82_OCC_RING ? total : (long)cus * K82_OCC_RING);     // K82_OCC_RING workgroups per CU
    if (p.taps == 3) {
        if (snake) launch_epi<4, 1, true, PREC, 1, 16, 3, 4>(p, epi, grid, s);
        else launch_epi<4, 1, false, PREC, 1, 16, 3, 4>(p, epi, grid, s);
    } else if (!snake) return false;
    else if (p.taps == 7) launch_epi<4, 1, true, PREC, 1, 16, 7, 4>(p, epi, grid, s);
    else if (p.taps == 11) launch_epi<4, 1, true, PREC, 1, 16, 11, 4>(p, epi, grid, s);
    else return false;
    ++g_shape_launches[CONV_SHAPE_RING];
    return true;
}