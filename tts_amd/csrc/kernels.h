// Kernel parameter blocks and launch prototypes shared by conv_mfma.hip, ops.hip, engine.hip.
// Every activation is channel-first: x[b][c][t] at  x + b*bstride + c*pitch + t,  and every
// sentence b has its own valid length  L_b = lmul*lens[b] + ladd  (lens = tokens per sentence
// at token rate, frames per sentence at frame rate).  Positions t >= L_b hold garbage that no
// kernel ever reads un-masked, so padding is bit-invisible.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define K82_HARM 9            // fundamental + 8 overtones (SineGen harmonic_num = 8)
#define K82_MAXTOK_P1 513   // 512 tokens + 1 (prefix-sum array of the alignment kernel)

struct LenSpec {            // per-sentence valid length
    const int* lens;        // device i32[B]; nullptr => every sentence has length ladd
    int lmul, ladd;
    __host__ __device__ int len(int b) const { return lens ? lmul * lens[b] + ladd : ladd; }
    // PACKED layout (phase 1, token rate): the sentences of a batch lie side by side on ONE time axis, x[c][off[b] + t]
    // with off[b] = T_0 + ... + T_{b-1} (device i32[B]) and a row pitch of the whole batch -- every Linear of phase 1 is a
    // 1-tap conv, for which the batch then is one long sentence (dense 128-column tiles whatever the sentence lengths:
    // a tile of a 35-token sentence alone is 3/4 padding and re-reads the whole weight matrix for it).  nullptr: the
    // padded layout, one row block of bs floats per sentence.  Kernels that work per sentence (embedding, attention,
    // layer norm, BiLSTM, durations, length regulation) take their base offset from base().
    const int* off;
    __host__ __device__ long base(int b, long bs) const { return off ? (long)off[b] : (long)b * bs; }
};

enum { ACT_NONE = 0, ACT_LEAKY = 1, ACT_SNAKE = 2 };
enum { TR_NONE = 0, TR_SCALE = 1, TR_NORM = 2 };
enum { OACT_NONE = 0, OACT_GELU_NEW = 1 };

// Implicit-GEMM Conv1d on v_mfma_f32_32x32x2_f32 (conv_mfma.hip).
struct ConvP {
    const float* x; long xbs; int xpitch;      // input  [B][cin_p][xpitch]
    const float* w;                            // weights [taps][cin_p][cout_p] (tap-major, cout contiguous)
    const uint16_t* wb;                        // bf16 image of the same weights (nullptr: fp32 MFMA):
                                               // [taps][cin_p/16][2 k-halves][cout_p][8], 16 B per (row, k-half)
    const float* wf;                           // fp32 fragment image [taps][cin_p/16][2 k-halves][cout_p][8]: element j is
                                               // w[tap][16 it + 2 j + half][co], the 8 A values a lane feeds to the 8
                                               // fp32 MFMA k-steps of a ci-tile, 32 contiguous bytes (nullptr: 1x4 shape)
    int wb_split;                              // 1: wb is the split image [..][cout_p][hi 8 | lo 8] (32 B per row and
                                               // k-half), hi = rne_bf16(w), lo = rne_bf16(w - hi): the bf16x3 mode
    const float* bias;                         // [cout_p] or nullptr
    float* y; long ybs; int ypitch;            // output (nullptr: accumulate only)
    LenSpec lin, lout;                         // input / output valid lengths
    int cin, cin_p, cout, cout_p, taps, dil, pad;
    // input transform, applied to valid positions only (zero padding comes after it):
    //   TR_SCALE: v = in_scale*x ; TR_NORM: v = ((x-mean)*rstd)*g + beta with norm[b*nstride+ci] = {mean,rstd,g,beta}
    //   then ACT_LEAKY: v<0 ? slope*v : v ; ACT_SNAKE: v + inv_alpha*sin(alpha*v)^2
    int tr; float in_scale; const float4* norm; int nstride;
    int act; float slope; const float* alpha; const float* inv_alpha;
    // epilogue: v = out_act(acc + bias) ; v = (v + res[t >> res_shift]) * out_scale ; y = v ; accbuf (+)= v*acc_scale
    int out_act;
    const float* res; long rbs; int rpitch; int res_shift;
    float out_scale;
    float* accbuf; long abs_; int apitch; int acc_init; float acc_scale;
    // transposed-conv phase interleave: row m = r*up_cout + co is stored at y[co][up_s*t + r + up_off]
    int up_s, up_cout_shift, up_off;
    const unsigned* tapmask;                   // per 64-row M tile: bitmask of non-zero taps (nullptr: all)
    int B, Lmax_out;                           // grid extents
    // bf16 activations in HBM (bf16 mode, resblock-internal tensor t1 of istftnet.AdaINResBlock1): the tensor is stored in
    // the MFMA-operand order of the staged LDS tile -- [C / 2][pitch] dwords, dword (q, t) = bf16(x[2q][t]) | bf16(x[2q+1][t]) << 16
    // (round-to-nearest-even) -- in the first half of the fp32 buffer it replaces (same bs, same pitch).  y_b16: this conv
    // stores its output so (plain epilogue, ring kernels only; the statistics partials are those of the fp32 values);
    // x_b16: this conv reads its input so (half the bytes through the L1 return path that bounds the ring kernels).
    int x_b16, y_b16;
    int cin_pad_ok;                            // 1: the caller zeroed norm[b][cin .. cin_p) (TR_NORM): the padded channels of the last 16-channel
                                               // tile stage exact zeros without a per-element select, which lets a conv whose cin is not a
                                               // multiple of 16 (the decoder's 1090- and 514-channel inputs) take the bf16 ring kernels
    int dbg;                                   // test switches of the engine: bit 0 = no ring pipeline (K82_RING=0), bit 1 = no 256 x 128 tiles (K82_BM256=0)
    int kflush;                                // > 0: blocked summation over K (fp32 kernels, conv_mfma.hip KSPLIT) -- the accumulator is folded
                                               // into a second one every kflush 16-channel tiles; 0: one sequential chain
    // split-K (small fp32 launches with blocked summation, conv_mfma.hip EPI_PART): part != nullptr -- workgroup (tile, z) sums
    // K block z (kflush 16-channel tiles) alone and stores the raw block sum at part[((z * B + b) * cout_p + m) * ppitch + t];
    // splitk_reduce then adds the nblk block sums IN BLOCK ORDER and applies the epilogue: the bits of the one-workgroup kernel,
    // whose K chain (one dependent global -> LDS -> MFMA step per 32 channels) is what a single sentence's Linears wait for.
    float* part; int nblk, ppitch;
    int cus;                                   // compute units of the device (hipDeviceProp_t::multiProcessorCount): persistent grids and the
                                               // large-launch thresholds scale with it (0: 256, an unpartitioned MI355X)
    int spin_limit;                            // polls an LDS3 wave waits for its workgroup's step counter before it gives up (engine: 1 << 16)
    int* err;                                  // device word, set to 2 when that wait ran out (the tile it then reads may be unstaged): the engine
                                               // checks it behind its stream synchronize and fails the call with K82_EHIP.  nullptr: not reported
    unsigned long long* trace;                 // K82_TRACE builds only (tools/trace_conv.py): per-wave phase time stamps; nullptr otherwise
    // instance-norm partials of the stored output (wide epilogues only): stat[b*sbs + m*snt + t/32] = (sum y, sum y^2)
    // over the valid columns of the 32-column span; every entry is written by exactly one wave (no atomics), and the
    // 32-column granularity does not depend on the tile shape a launch picks.  nullptr: not collected.
    float2* stat; long sbs; int snt;
};
void launch_conv(const ConvP& p, hipStream_t s);
// launches of the large-tile families since the process started (tests assert that a bit-identity comparison really
// ran them): 0 = 256 x 128 workgroup tiles (fp32 4 x 2 waves / bf16 modes 8 x 1 waves), 1 = bf16 ring 128 x 256 tiles (4 x 2 waves),
// 2 = bf16 ring pipeline (any shape), 3 = LDS3 two-set kernels, 4 = blocked-summation (KSPLIT) kernels
enum { CONV_SHAPE_BM256 = 0, CONV_SHAPE_RING_WN2 = 1, CONV_SHAPE_RING = 2, CONV_SHAPE_LDS3 = 3, CONV_SHAPE_KSPLIT = 4, CONV_SHAPE_SPLITK = 5, CONV_NSHAPE = 6 };
int conv_splitk_blocks(const ConvP& p);                    // K blocks launch_conv would split this launch into if p.part were set (0: none)
long conv_shape_launches(int shape);
double conv_flops(const ConvP& p, const int* host_lens);   // algorithmic FLOPs of one launch

// ---- ops.hip ---------------------------------------------------------------------------
// AdaIN statistics: norm[b*nstride+c] = {mean, rstd, 1+gamma, beta} over t < L_b.
// The same from the partials a conv epilogue left (ConvP::stat): fp64 sum over ceil(L_b/32) entries per row.
void launch_adain_finalize(const float2* part, long sbs, int snt, LenSpec ls, int C, int B, const float* gb, int gbstride,
                           int goff, float4* norm, int nstride, hipStream_t s);
// dst = src with (1+gamma, beta) of another AdaIN layer over the same tensor
void launch_adain_regamma(const float4* src, float4* dst, int nstride, int C, int B, const float* gb, int gbstride, int goff,
                          hipStream_t s);
void launch_adain_stats(const float* x, long xbs, int xpitch, LenSpec ls, int C, int B,
                        const float* gb, int gbstride, int goff,   // gamma at gb[b*gbstride+goff+c], beta at +C
                        float4* norm, int nstride, int Lmax, hipStream_t s);
// gb[b][r] = W[r][:] . style[b][soff:soff+128] + bias[r]
void launch_style_fc(const float* W, const float* bias, int rows, const float* ref_s, int soff,
                     float* gb, int gbstride, int B, hipStream_t s);
// LayerNorm over channels of [B][C][pitch] per (b,t): y = ((x+res)-mean)*rstd*w[c]+bias[c]  (w/bias per channel)
// or AdaLN: (1+gamma_b[c])*xn + beta_b[c] when gb != nullptr.  Optional LeakyReLU afterwards.
void launch_layernorm_cols(const float* x, const float* res, float* y, long bs, long ybs, int pitch, LenSpec ls,
                           int C, int B, int Tmax, float eps, const float* w, const float* bias,
                           const float* gb, int gbstride, int goff, float leaky, hipStream_t s);
void launch_albert_embed(const int64_t* ids, int Tmax, const float* word, const float* pos, const float* type0,
                         const float* lnw, const float* lnb, float eps, float* y, long ybs, int ypitch,
                         LenSpec ls, int B, int n_token, hipStream_t s);
void launch_embed(const int64_t* ids, int Tmax, const float* emb, int C, float* y, long ybs, int ypitch,
                  LenSpec ls, int B, int n_token, hipStream_t s);
// qkv [B][3*H*64][pitch] -> ctx [B][H*64][pitch]
void launch_attention(const float* qkv, long qbs, int qpitch, float* ctx, long cbs, int cpitch,
                      LenSpec ls, int heads, int B, int Tmax, hipStream_t s);
// BiLSTM recurrence, hidden 256.  gx [B][2048][pitch] (fwd gates i,f,g,o then reverse), whh [2][256(k)][256(j)][4]
void launch_lstm(const float* gx, long gbs, int gpitch, const float* whh, float* y, long ybs, int ypitch,
                 LenSpec ls, int B, hipStream_t s);
void launch_lstm_split(const float* gx, long gbs, int gpitch, const float* whh, float* y, long ybs, int ypitch, LenSpec ls,
                       int B, float* hx, int* err, int cus, int spin_limit, hipStream_t s);
// The split kernel is correct only while all four workgroups of a recurrence are resident together, which the launcher
// can promise only if the handle has the chip to itself.  A second process or a second handle on the same GPU
// (basic.tts_kokoro_2_play_pause_works.py:34-41,136-146 pre-rolls in the parent while the producer child synthesizes)
// can hold the CUs a partner needs: the bounded spin (spin_limit polls) then sets *err, every wave leaves, and the
// engine re-runs the phase on lstm_kernel -- bit-identical by construction -- and stays on it for a while.
// The split recurrence is used up to CUs / 2 sentences (four rounds of the grid).  Its four workgroups per (direction, sentence) are adjacent in
// dispatch order; with more than CUs / 8 sentences the grid runs in rounds (in-order dispatch: at most one recurrence
// straddles the residency boundary and gets its missing parts as soon as any complete one finishes).  Per
// configs[1]-shaped step, split vs streaming kernel: B = 16: 9.4 vs 26.6 ms, B = 32: 10.1 vs 28.0, B = 64: 19.6 vs 36.2.
// rows [r0, r0+n) of y get value v[b*vstride + voff + i] for t < L_b (style broadcast)
void launch_fill_rows(float* y, long ybs, int ypitch, int r0, int n, const float* v, int vstride, int voff,
                      LenSpec ls, int B, int Tmax, hipStream_t s);
void launch_durations(const float* logits, long lbs, int lpitch, int ndur, const float* speed,
                      const int* lens, int B, int Tmax, int* dur, float* raw, hipStream_t s, const int* loff = nullptr);   // loff: packed logits
// idx[b][f] = token index of frame f ; nfr[b] = sum dur
void launch_alignment(const int* dur, int Tmax, const int* lens, int B, int* idx, int ipitch, int* nfr, hipStream_t s);
void launch_gather(const float* x, long xbs, int xpitch, int C, const int* idx, int ipitch, const int* nfr,
                   float* y, long ybs, int ypitch, int B, int Fmax, hipStream_t s, const int* xoff = nullptr);        // xoff: packed token rows
// depthwise ConvTranspose1d k3 s2 p1 op1 over act(norm(x)) : [C][L] -> [C][2L]
void launch_pool_up2(const float* x, long xbs, int xpitch, const float4* norm, int nstride, float slope,
                     const float* w, const float* bias, float* y, long ybs, int ypitch, LenSpec ls,
                     int C, int B, int Lmax, hipStream_t s);
// Conv1d 1->1 k3 s2 p1 on a single row: y[f] = w0*x[2f-1]+w1*x[2f]+w2*x[2f+1]+b, f < F_b
void launch_conv_k3s2(const float* x, long xbs, const float* w3b, float* y, long ybs, const int* nfr,
                      int B, int Fmax, hipStream_t s);
void launch_copy_rows(const float* x, long xbs, int xpitch, float* y, long ybs, int ypitch, int rows,
                      LenSpec ls, int B, int Lmax, hipStream_t s);
// harmonic source (source.hip)
int source_init_constants();   // fills the __constant__ window / twiddle tables of the current device
void launch_src_phase(const float* f0, long fbs, const int* nfr, int B, float* ph, long pbs, int ppitch, hipStream_t s);
void launch_src_har(const float* f0, long fbs, const float* ph, long pbs, int ppitch, const int* nfr,
                    const float* noise, const int64_t* noise_off, const uint64_t* seeds,
                    const float* lin_w, const float* lin_b, float* har_src, long hbs, int B, int Fmax, hipStream_t s);
// STFT n_fft 20 hop 5 (center, reflect) -> har [B][32][p1] (22 valid rows: 11 mag, 11 phase) and its
// stride-6 polyphase image harp [B][144][p0]: harp[ci*6+r][q] = har[ci][6q+r-3]
void launch_stft(const float* har_src, long hbs, const int* nfr, int B, int Fmax,
                 float* har, long bs1, int p1, float* harp, long bs0, int p0, hipStream_t s);
int launch_phase_interleave(const float* yp, long pbs, int ppitch, float* y, long ybs, int ypitch, const float* res, long rbs,
                             int rpitch, LenSpec lin, int cout, int s, int off, int B, int Lmax, hipStream_t st);
void launch_reflect_fix(float* y, long ybs, int ypitch, const float* xs, long xbs, int xpitch, int C, int B, hipStream_t s);
// conv_post output [B][>=22][p1] -> audio: spec=exp(x[:11]), phase=sin(x[11:]), iSTFT n_fft 20 hop 5
void launch_istft(const float* cp, long cbs, int cpitch, const int* nfr, int B, int Fmax,
                  float* audio, const int64_t* aoff, hipStream_t s);
void launch_pcm16(const float* src, int16_t* dst, int64_t n, hipStream_t s);
