// Bandwidth- and latency-bound kernels around the MFMA conv: AdaIN statistics, LayerNorm,
// embeddings, attention, the BiLSTM recurrence, durations/alignment/gather (length
// regulation), the depthwise up-sampling pool and small copies.  All channel-first
// (kernels.h).  64-wide wavefronts throughout; every reduction is deterministic (fixed
// order, no float atomics) so two runs give identical bits.
#include "kernels.h"

#define DIVUP(a, b) (((a) + (b)-1) / (b))

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sumf(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// ---------------------------------------------------------------- AdaIN statistics (row a13)
__device__ __forceinline__ void adain_finish(double s, double q, int L, int b, int c, int C, const float* gb, int gbstride,
                                             int goff, float4* norm, int nstride) {
    __shared__ double sh[8];
    const int tid = threadIdx.x;
    s = wave_sum(s);
    q = wave_sum(q);
    const int w = tid >> 6;
    if ((tid & 63) == 0) { sh[w] = s; sh[4 + w] = q; }
    __syncthreads();
    if (tid == 0) {
        s = sh[0] + sh[1] + sh[2] + sh[3];
        q = sh[4] + sh[5] + sh[6] + sh[7];
        const double mean = s / L;
        double var = q / L - mean * mean;
        if (var < 0) var = 0;
        const float rstd = (float)(1.0 / sqrt(var + 1e-5));
        const float g = 1.0f + gb[(long)b * gbstride + goff + c];
        const float be = gb[(long)b * gbstride + goff + C + c];
        norm[(long)b * nstride + c] = make_float4((float)mean, rstd, g, be);
    }
}
// One workgroup per (c, b) row: one coalesced float4 sweep, sums in fp64 (one-pass mean and
// E[x^2] are exact enough in double: 53-bit accumulators over <= 2.5e5 fp32 samples).
__global__ __launch_bounds__(256) void adain_stats_kernel(const float* x, long xbs, int xpitch, LenSpec ls, int C,
                                                         const float* gb, int gbstride, int goff, float4* norm,
                                                         int nstride) {
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int L = ls.len(b);
    const float* xr = x + (long)b * xbs + (long)c * xpitch;
    double s = 0.0, q = 0.0;
    const int L4 = L >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(xr);
    // four independent 16-byte loads in flight per thread (one per trip left the sweep at 0.52 of the HBM rate: a row of
    // 176 k floats is 172 dependent load -> add trips); the adds keep the order of the one-load loop, so the sums are
    // the bits they were
    int i = tid;
    for (; i + 768 < L4; i += 1024) {
        const float4 v0 = x4[i], v1 = x4[i + 256], v2 = x4[i + 512], v3 = x4[i + 768];
        s += (double)v0.x + (double)v0.y + (double)v0.z + (double)v0.w;
        q += (double)v0.x * v0.x + (double)v0.y * v0.y + (double)v0.z * v0.z + (double)v0.w * v0.w;
        s += (double)v1.x + (double)v1.y + (double)v1.z + (double)v1.w;
        q += (double)v1.x * v1.x + (double)v1.y * v1.y + (double)v1.z * v1.z + (double)v1.w * v1.w;
        s += (double)v2.x + (double)v2.y + (double)v2.z + (double)v2.w;
        q += (double)v2.x * v2.x + (double)v2.y * v2.y + (double)v2.z * v2.z + (double)v2.w * v2.w;
        s += (double)v3.x + (double)v3.y + (double)v3.z + (double)v3.w;
        q += (double)v3.x * v3.x + (double)v3.y * v3.y + (double)v3.z * v3.z + (double)v3.w * v3.w;
    }
    for (; i < L4; i += 256) {
        const float4 v = x4[i];
        s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
        q += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    for (int i = (L4 << 2) + tid; i < L; i += 256) {
        const float v = xr[i];
        s += v;
        q += (double)v * v;
    }
    adain_finish(s, q, L, b, c, C, gb, gbstride, goff, norm, nstride);
}
// (sum, sum of squares) per 32 columns from the producing conv's epilogue -> the same float4 per row
__global__ __launch_bounds__(256) void adain_finalize_kernel(const float2* part, long sbs, int snt, LenSpec ls, int C,
                                                            const float* gb, int gbstride, int goff, float4* norm,
                                                            int nstride) {
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int L = ls.len(b), nt = (L + 31) >> 5;
    const float2* pr = part + (long)b * sbs + (long)c * snt;
    double s = 0.0, q = 0.0;
    for (int i = tid; i < nt; i += 256) {
        const float2 v = pr[i];
        s += (double)v.x;
        q += (double)v.y;
    }
    adain_finish(s, q, L, b, c, C, gb, gbstride, goff, norm, nstride);
}
// The three parallel resblocks of a generator stage normalise the same tensor with their own (gamma, beta): one
// sweep, then the other two take its (mean, rstd) -- the bits a second sweep would give.
__global__ __launch_bounds__(256) void adain_regamma_kernel(const float4* src, float4* dst, int nstride, int C, const float* gb,
                                                           int gbstride, int goff) {
    const int c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (c >= C) return;
    float4 v = src[(long)b * nstride + c];
    v.z = 1.0f + gb[(long)b * gbstride + goff + c];
    v.w = gb[(long)b * gbstride + goff + C + c];
    dst[(long)b * nstride + c] = v;
}
void launch_adain_regamma(const float4* src, float4* dst, int nstride, int C, int B, const float* gb, int gbstride, int goff,
                          hipStream_t s) {
    hipLaunchKernelGGL(adain_regamma_kernel, dim3(DIVUP(C, 256), B), dim3(256), 0, s, src, dst, nstride, C, gb, gbstride, goff);
}
void launch_adain_finalize(const float2* part, long sbs, int snt, LenSpec ls, int C, int B, const float* gb, int gbstride,
                           int goff, float4* norm, int nstride, hipStream_t s) {
    hipLaunchKernelGGL(adain_finalize_kernel, dim3(C, B), dim3(256), 0, s, part, sbs, snt, ls, C, gb, gbstride, goff, norm,
                       nstride);
}
void launch_adain_stats(const float* x, long xbs, int xpitch, LenSpec ls, int C, int B, const float* gb,
                        int gbstride, int goff, float4* norm, int nstride, int Lmax, hipStream_t s) {
    (void)Lmax;
    hipLaunchKernelGGL(adain_stats_kernel, dim3(C, B), dim3(256), 0, s, x, xbs, xpitch, ls, C, gb, gbstride, goff,
                       norm, nstride);
}

// --------------------------------------------------------------------------- style FCs
// One wave per output row: 128-long dot product, 2 elements per lane.
__global__ __launch_bounds__(256) void style_fc_kernel(const float* W, const float* bias, int rows,
                                                      const float* ref_s, int soff, float* gb, int gbstride) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* s = ref_s + (long)b * 256 + soff;
    const float* w = W + (long)r * 128;
    float a = w[lane] * s[lane] + w[lane + 64] * s[lane + 64];
    a = wave_sumf(a);
    if (lane == 0) gb[(long)b * gbstride + r] = a + bias[r];
}
void launch_style_fc(const float* W, const float* bias, int rows, const float* ref_s, int soff, float* gb,
                     int gbstride, int B, hipStream_t s) {
    hipLaunchKernelGGL(style_fc_kernel, dim3(DIVUP(rows, 4), B), dim3(256), 0, s, W, bias, rows, ref_s, soff, gb,
                       gbstride);
}

// ------------------------------------------------------------- LayerNorm over channels
// Workgroup = 64 positions (lane = t, coalesced) x 16 channel groups (wave w owns channels w, w+16, ...): every
// value is loaded ONCE into registers (up to 48 per thread, all loads independent), the 16 partial sums meet in
// LDS.  Two-pass (mean, then centred variance) like the reference's layer_norm.  C <= 768 (checked by the launcher,
// which falls back to the serial kernel below for anything larger).
// NI = C / 16 is a template parameter (48: ALBERT, 32: AdaLN / text encoder): no channel-validity test anywhere, and the
// loads are issued as BLOCKS in front of a scheduling barrier.  (Written as one loop -- load, add, next -- the compiler
// waited for every load before it issued the next one, and again for every gain / offset pair in the store loop: 96
// sequential L2 round trips, 27 us for a kernel that moves 120 KB; a tenth of a single sentence's latency.)  Same
// operations in the same order as that loop: same bits.
template <int NI>
__global__ __launch_bounds__(1024) void layernorm_cols_kernel(const float* x, const float* res, float* y, long bs,
                                                             long ybs, int pitch, LenSpec ls, int C, float eps, const float* w,
                                                             const float* bias, const float* gb, int gbstride,
                                                             int goff, float leaky) {
    __shared__ float red[2][16][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int L = ls.len(b);
    const int t = blockIdx.x * 64 + lane;
    const bool ok = t < L;
    const int tc = ok ? t : 0;                               // clamped: every lane loads, only valid lanes store
    const float* xc = x + ls.base(b, bs) + tc + (long)wv * pitch;
    const long cstep = (long)16 * pitch;
    float v[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) v[i] = xc[i * cstep];
    __builtin_amdgcn_sched_barrier(0);
    if (res) {                                               // (uniform; no caller passes a residual today)
        const float* rc = res + ls.base(b, bs) + tc + (long)wv * pitch;
#pragma unroll
        for (int i0 = 0; i0 < NI; i0 += 16) {
            float r[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) r[k] = rc[(i0 + k) * cstep];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 16; ++k) v[i0 + k] = v[i0 + k] + r[k];
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) s += v[i];
    red[0][wv][lane] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) tot += red[0][k][lane];
    const float mean = tot / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const float d = v[i] - mean;
        v[i] = d;
        q = fmaf(d, d, q);
    }
    red[1][wv][lane] = q;
    __syncthreads();
    float qt = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) qt += red[1][k][lane];
    const float rstd = 1.0f / sqrtf(qt / C + eps);
    float* yc = y + ls.base(b, ybs) + tc + (long)wv * pitch;
    // gain / offset of channel c: (1 + gamma, beta) of the AdaLN style row or (w, bias), through pointers chosen once
    // (0 + w == w and a * b == b * a exactly: the bits of the two-branch form)
    const float* gp = (gb ? gb + (long)b * gbstride + goff : w) + wv;
    const float* bp = (gb ? gb + (long)b * gbstride + goff + C : bias) + wv;
    const float gadd = gb ? 1.0f : 0.0f;
    const float lk = leaky;
#pragma unroll
    for (int i0 = 0; i0 < NI; i0 += 16) {
        float ga[16], be[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { ga[k] = gp[16 * (i0 + k)]; be[k] = bp[16 * (i0 + k)]; }
        __builtin_amdgcn_sched_barrier(0);
        if (ok) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                float o = v[i0 + k] * rstd;
                o = fmaf(gadd + ga[k], o, be[k]);             // (spelled out: left to contraction, the packed form of this loop was not fused)
                o = (lk != 1.0f && o < 0.f) ? o * lk : o;
                yc[(i0 + k) * cstep] = o;
            }
        }
    }
}
// serial fallback (C > 768): thread per (b, t), three passes
__global__ __launch_bounds__(64) void layernorm_cols_serial_kernel(const float* x, const float* res, float* y, long bs,
                                                                  long ybs, int pitch, LenSpec ls, int C, float eps,
                                                                  const float* w, const float* bias, const float* gb,
                                                                  int gbstride, int goff, float leaky) {
    const int b = blockIdx.y, t = blockIdx.x * 64 + threadIdx.x;
    if (t >= ls.len(b)) return;
    const float* xc = x + ls.base(b, bs) + t;
    const float* rc = res ? res + ls.base(b, bs) + t : nullptr;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += xc[(long)c * pitch] + (rc ? rc[(long)c * pitch] : 0.f);
    const float mean = s / C;
    float q = 0.f;
    for (int c = 0; c < C; ++c) {
        const float d = xc[(long)c * pitch] + (rc ? rc[(long)c * pitch] : 0.f) - mean;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(q / C + eps);
    float* yc = y + ls.base(b, ybs) + t;
    const float* g = gb ? gb + (long)b * gbstride + goff : nullptr;
    for (int c = 0; c < C; ++c) {
        float v = (xc[(long)c * pitch] + (rc ? rc[(long)c * pitch] : 0.f) - mean) * rstd;
        if (g) v = (1.0f + g[c]) * v + g[C + c];
        else v = v * w[c] + bias[c];
        if (leaky != 1.0f) v = v < 0.f ? v * leaky : v;
        yc[(long)c * pitch] = v;
    }
}
void launch_layernorm_cols(const float* x, const float* res, float* y, long bs, long ybs, int pitch, LenSpec ls, int C,
                           int B, int Tmax, float eps, const float* w, const float* bias, const float* gb,
                           int gbstride, int goff, float leaky, hipStream_t s) {
    if (C == 768)
        hipLaunchKernelGGL(layernorm_cols_kernel<48>, dim3(DIVUP(Tmax, 64), B), dim3(1024), 0, s, x, res, y, bs, ybs, pitch, ls,
                           C, eps, w, bias, gb, gbstride, goff, leaky);
    else if (C == 512)
        hipLaunchKernelGGL(layernorm_cols_kernel<32>, dim3(DIVUP(Tmax, 64), B), dim3(1024), 0, s, x, res, y, bs, ybs, pitch, ls,
                           C, eps, w, bias, gb, gbstride, goff, leaky);
    else
        hipLaunchKernelGGL(layernorm_cols_serial_kernel, dim3(DIVUP(Tmax, 64), B), dim3(64), 0, s, x, res, y, bs, ybs, pitch,
                           ls, C, eps, w, bias, gb, gbstride, goff, leaky);
}

// ------------------------------------------------------------------------- embeddings
__global__ __launch_bounds__(64) void albert_embed_kernel(const int64_t* ids, int Tmax, const float* word,
                                                         const float* pos, const float* type0, const float* lnw,
                                                         const float* lnb, float eps, float* y, long ybs, int ypitch,
                                                         LenSpec ls, int n_token) {
    const int b = blockIdx.y, t = blockIdx.x * 64 + threadIdx.x;
    if (t >= ls.len(b)) return;
    // ids that live on the device cannot be validated by the host: an id outside the table reads row 0 / n_token-1
    // instead of faulting (host ids are rejected with K82_EINVAL before any launch)
    const int64_t id = min(max(ids[(long)b * Tmax + t], (int64_t)0), (int64_t)(n_token - 1));
    const float* wv = word + id * 128;
    const float* pv = pos + (long)t * 128;
    float s = 0.f;
    for (int c = 0; c < 128; ++c) s += (wv[c] + type0[c]) + pv[c];
    const float mean = s / 128.f;
    float q = 0.f;
    for (int c = 0; c < 128; ++c) {
        const float d = (wv[c] + type0[c]) + pv[c] - mean;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(q / 128.f + eps);
    float* yc = y + ls.base(b, ybs) + t;
    for (int c = 0; c < 128; ++c) yc[(long)c * ypitch] = ((wv[c] + type0[c]) + pv[c] - mean) * rstd * lnw[c] + lnb[c];
}
void launch_albert_embed(const int64_t* ids, int Tmax, const float* word, const float* pos, const float* type0,
                         const float* lnw, const float* lnb, float eps, float* y, long ybs, int ypitch, LenSpec ls,
                         int B, int n_token, hipStream_t s) {
    hipLaunchKernelGGL(albert_embed_kernel, dim3(DIVUP(Tmax, 64), B), dim3(64), 0, s, ids, Tmax, word, pos, type0,
                       lnw, lnb, eps, y, ybs, ypitch, ls, n_token);
}

__global__ __launch_bounds__(64) void embed_kernel(const int64_t* ids, int Tmax, const float* emb, int C, float* y,
                                                  long ybs, int ypitch, LenSpec ls, int n_token) {
    const int b = blockIdx.y, t = blockIdx.x * 64 + threadIdx.x;
    if (t >= ls.len(b)) return;
    const float* e = emb + min(max(ids[(long)b * Tmax + t], (int64_t)0), (int64_t)(n_token - 1)) * C;
    float* yc = y + (long)b * ybs + t;
    for (int c = 0; c < C; ++c) yc[(long)c * ypitch] = e[c];
}
void launch_embed(const int64_t* ids, int Tmax, const float* emb, int C, float* y, long ybs, int ypitch, LenSpec ls,
                  int B, int n_token, hipStream_t s) {
    hipLaunchKernelGGL(embed_kernel, dim3(DIVUP(Tmax, 64), B), dim3(64), 0, s, ids, Tmax, emb, C, y, ybs, ypitch, ls, n_token);
}

// ------------------------------------------------------------------ attention (row a5)
// One wave per (32 queries, head, sentence) on v_mfma_f32_32x32x2_f32 (exact fp32), flash style over 32-key tiles:
//   S^T = K Q^T   (rows = keys, columns = queries): A = K[key][d] straight from the channel-first qkv tensor (lane =
//                  key: coalesced rows), B = the lane's 32 pre-scaled Q values (lane = query), 32 MFMAs
//   online softmax per COLUMN: a lane and its partner (lane ^ 32) hold the 32 keys of one query -- the running
//                  max / sum / rescale are lane-local plus one cross-half exchange, no LDS
//   O^T += V^T P^T (rows = d, columns = queries): the B operand of k-step s is the lane's OWN P register s -- the
//                  step simply consumes the key that register holds (key order inside a tile does not matter to the
//                  sum) -- and A = V[d][that key] from the V tile staged in LDS ([d][33]: conflict-free), 32 MFMAs
// so no score or probability ever moves between lanes.  (The first version, one query per lane with 64 x 64 scalar
// FMAs per key tile, spent 15 ms per configs[1] step and 160 us per launch on a single sentence.)
typedef float attn_f16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(64) void attention_kernel(const float* qkv, long qbs, int qpitch, float* ctx, long cbs,
                                                      int cpitch, LenSpec ls, int heads) {
    __shared__ float Vs[64 * 33];
    const int b = blockIdx.z, hd = blockIdx.y, lane = threadIdx.x, l31 = lane & 31, lhi = lane >> 5;
    const int T = ls.len(b);
    const int q0 = blockIdx.x * 32;
    if (q0 >= T) return;
    const int HD = heads * 64;
    const float* qb = qkv + ls.base(b, qbs) + (long)(hd * 64) * qpitch;
    const float* kb = qb + (long)HD * qpitch;
    const float* vb = kb + (long)HD * qpitch;
    const int tq = min(q0 + l31, T - 1);
    float qf[32];
#pragma unroll
    for (int s = 0; s < 32; ++s) qf[s] = qb[(long)(2 * s + lhi) * qpitch + tq] * 0.125f;
    attn_f16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m = -1e30f, l = 0.f;
    for (int j0 = 0; j0 < T; j0 += 32) {
        const int kk = min(j0 + l31, T - 1);
        __syncthreads();                                   // the previous tile's V reads are done
#pragma unroll 8
        for (int i = 0; i < 32; ++i) Vs[(lhi * 32 + i) * 33 + l31] = vb[(long)(lhi * 32 + i) * qpitch + kk];
        attn_f16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll 8
        for (int s = 0; s < 32; ++s)
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kb[(long)(2 * s + lhi) * qpitch + kk], qf[s], sacc, 0, 0, 0);
        float pv[16], tmax = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = j0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            pv[r] = key < T ? sacc[r] : -1e30f;
            tmax = fmaxf(tmax, pv[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float mn = fmaxf(m, tmax);
        const float corr = expf(m - mn);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = j0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            pv[r] = key < T ? expf(pv[r] - mn) : 0.f;
            psum += pv[r];
        }
        psum += __shfl_xor(psum, 32);
        l = l * corr + psum;
        m = mn;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= corr; o1[r] *= corr; }
        __syncthreads();                                   // V tile visible
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int kidx = (s & 3) + 8 * (s >> 2) + 4 * lhi;
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[l31 * 33 + kidx], pv[s], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[(32 + l31) * 33 + kidx], pv[s], o1, 0, 0, 0);
        }
    }
    if (q0 + l31 < T) {
        const float inv = 1.0f / l;
        float* cb = ctx + ls.base(b, cbs) + (long)(hd * 64) * cpitch + q0 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = (r & 3) + 8 * (r >> 2) + 4 * lhi;
            cb[(long)d * cpitch] = o0[r] * inv;
            cb[(long)(32 + d) * cpitch] = o1[r] * inv;
        }
    }
}
void launch_attention(const float* qkv, long qbs, int qpitch, float* ctx, long cbs, int cpitch, LenSpec ls,
                      int heads, int B, int Tmax, hipStream_t s) {
    hipLaunchKernelGGL(attention_kernel, dim3(DIVUP(Tmax, 32), heads, B), dim3(64), 0, s, qkv, qbs, qpitch, ctx, cbs,
                       cpitch, ls, heads);
}

// ------------------------------------------------------ BiLSTM recurrence (rows a7, a8, a10, a11)
// One workgroup per (direction, sentence), thread j owns hidden unit j: its 4 gate rows of
// W_hh stream from L2 as one float4 per k (layout [dir][k][j][i,f,g,o], 1 MB per direction,
// L2-resident and shared by all sentences), h is double-buffered in LDS (one barrier per
// step), c stays in a register.  fp32 throughout: the duration branch must round exactly.
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// 1024 threads: thread (part p = tid >> 8, unit j = tid & 255) owns the k-quarter [64p, 64p + 64) of unit j's four
// gate rows, so four times as many W_hh loads are in flight per step as with one thread per unit (the step is
// bound by the latency of streaming 1 MB from L2 through one CU, not by its 256 k MACs); the partial sums meet
// in LDS.  The four gate pre-activations of the NEXT step are prefetched while this step's weights stream.
__global__ __launch_bounds__(1024) void lstm_kernel(const float* gx, long gbs, int gpitch, const float* whh, float* y,
                                                   long ybs, int ypitch, LenSpec ls) {
    __shared__ __attribute__((aligned(16))) float hbuf[2][256];
    __shared__ __attribute__((aligned(16))) float4 part[4][256];
    const int dir = blockIdx.x, b = blockIdx.y, j = threadIdx.x & 255, pt = threadIdx.x >> 8;
    const int T = ls.len(b);
    const float4* W = reinterpret_cast<const float4*>(whh) + (long)dir * 256 * 256 + (long)(64 * pt) * 256 + j;
    const float* g0 = gx + ls.base(b, gbs) + (long)(dir * 1024 + j) * gpitch;
    const long gs = (long)256 * gpitch;
    float* yo = y + ls.base(b, ybs) + (long)(dir * 256 + j) * ypitch;
    float c = 0.f;
    if (pt == 0) hbuf[0][j] = 0.f;
    float4 gn = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pt == 0 && T > 0) {
        const int t0 = dir ? T - 1 : 0;
        gn = make_float4(g0[t0], g0[gs + t0], g0[2 * gs + t0], g0[3 * gs + t0]);
    }
    __syncthreads();
    int p = 0;
    for (int s = 0; s < T; ++s) {
        const int t = dir ? T - 1 - s : s;
        const float4 gcur = gn;
        if (pt == 0 && s + 1 < T) {                                  // next step's input projection, in flight early
            const int tn = dir ? t - 1 : t + 1;
            gn = make_float4(g0[tn], g0[gs + tn], g0[2 * gs + tn], g0[3 * gs + tn]);
        }
        float ai = 0.f, af = 0.f, ag = 0.f, ao = 0.f;
        const float4* h4 = reinterpret_cast<const float4*>(hbuf[p]) + 16 * pt;
#pragma unroll 8
        for (int k4 = 0; k4 < 16; ++k4) {
            const float4 hv = h4[k4];
            const float4 w0 = W[(long)(4 * k4 + 0) * 256];
            const float4 w1 = W[(long)(4 * k4 + 1) * 256];
            const float4 w2 = W[(long)(4 * k4 + 2) * 256];
            const float4 w3 = W[(long)(4 * k4 + 3) * 256];
            ai += w0.x * hv.x; af += w0.y * hv.x; ag += w0.z * hv.x; ao += w0.w * hv.x;
            ai += w1.x * hv.y; af += w1.y * hv.y; ag += w1.z * hv.y; ao += w1.w * hv.y;
            ai += w2.x * hv.z; af += w2.y * hv.z; ag += w2.z * hv.z; ao += w2.w * hv.z;
            ai += w3.x * hv.w; af += w3.y * hv.w; ag += w3.z * hv.w; ao += w3.w * hv.w;
        }
        part[pt][j] = make_float4(ai, af, ag, ao);
        __syncthreads();
        if (pt == 0) {
            const float4 p0 = part[0][j], p1 = part[1][j], p2 = part[2][j], p3 = part[3][j];
            ai = gcur.x + ((p0.x + p1.x) + (p2.x + p3.x));
            af = gcur.y + ((p0.y + p1.y) + (p2.y + p3.y));
            ag = gcur.z + ((p0.z + p1.z) + (p2.z + p3.z));
            ao = gcur.w + ((p0.w + p1.w) + (p2.w + p3.w));
            const float ig = sigmoidf_(ai), fg = sigmoidf_(af), gg = tanhf(ag), og = sigmoidf_(ao);
            c = fg * c + ig * gg;
            const float hn = og * tanhf(c);
            hbuf[p ^ 1][j] = hn;
            yo[t] = hn;
        }
        __syncthreads();
        p ^= 1;
    }
}
// Up to CUs / 2 sentences: one (direction, sentence) recurrence spread over FOUR workgroups, 64 hidden units
// each, with the unit's W_hh rows RESIDENT IN REGISTERS (thread = k-quarter x unit x gate pair: 64 k x 2 gates = 128
// VGPRs) instead of 1 MB streamed from L2 through one CU every step (7.8 us at 64 B/clk -- the step time of the
// kernel above when only two workgroups run).  Per step each workgroup publishes its 64 new h values in global memory
// as (value, step) words that the other three poll (bounded spin).  Arithmetic is IDENTICAL to lstm_kernel's: the
// same four k-quarter chains per gate in the same order, the same partial-sum tree, the same gate code -- a sentence
// gives the same bits alone (this kernel) and in a large batch (that kernel).
// Co-residency: the four parts of a recurrence are adjacent in dispatch order and the launcher uses this kernel only
// when every workgroup of the grid fits on the chip at once (one 512-thread, 256-VGPR workgroup per CU); a spin that
// exceeds its bound sets *err and every wave leaves its loop.
// NS sentences per workgroup share the register-resident weights (the launcher picks NS so that the whole grid is
// resident in ONE round: 8 * ceil(B / NS) workgroups <= CUs): the per-step exchange latency is paid once for NS
// recurrences; each sentence keeps its own chains, so the bits do not depend on NS.
template <int NS>
__global__ __launch_bounds__(512, 1) void lstm_split_kernel(const float* gx, long gbs, int gpitch, const float* whh, float* y,
                                                           long ybs, int ypitch, LenSpec ls, int B, float* hx, int* err,
                                                           int spin_limit) {
    __shared__ __attribute__((aligned(16))) float hbuf[NS][256];
    __shared__ float2 part[NS][4][2][64];
    __shared__ int dead;
    const int P = blockIdx.x & 3, dir = (blockIdx.x >> 2) & 1, grp = blockIdx.x >> 3, b0 = grp * NS;
    const int tid = threadIdx.x, pt = tid >> 7, g2 = (tid >> 6) & 1, jl = tid & 63, j = 64 * P + jl;
    int Tn[NS], Tmax = 0;
#pragma unroll
    for (int n = 0; n < NS; ++n) { Tn[n] = b0 + n < B ? ls.len(b0 + n) : 0; Tmax = max(Tmax, Tn[n]); }
    float2 w[64];
    {
        const float4* W = reinterpret_cast<const float4*>(whh) + (long)dir * 256 * 256 + (long)(64 * pt) * 256 + j;
#pragma unroll
        for (int k = 0; k < 64; ++k) w[k] = reinterpret_cast<const float2*>(W + (long)k * 256)[g2];
    }
    // gate stage: wave n (tid < 64 NS) finishes sentence n, lane = one of this workgroup's 64 hidden units
    const int fn = min(tid >> 6, NS - 1), fb = min(b0 + fn, B - 1);
    const bool fin = tid < 64 * NS;
    int Tf = 0;
#pragma unroll
    for (int n = 0; n < NS; ++n) Tf = (fn == n) ? Tn[n] : Tf;
    const float* g0 = gx + ls.base(fb, gbs) + (long)(dir * 1024 + j) * gpitch;
    const long gs = (long)256 * gpitch;
    float* yo = y + ls.base(fb, ybs) + (long)(dir * 256 + j) * ypitch;
    // exchange buffer: one 8-byte (value, step) word per hidden unit, sentence and step parity
    unsigned long long* H = reinterpret_cast<unsigned long long*>(hx) + (long)(grp * 2 + dir) * (2 * NS * 256);
    float c = 0.f;
    if (tid < 256)
#pragma unroll
        for (int n = 0; n < NS; ++n) hbuf[n][tid] = 0.f;
    if (tid == 0) dead = 0;
    float4 gn = make_float4(0.f, 0.f, 0.f, 0.f);
    if (fin && Tf > 0) {
        const int t0 = dir ? Tf - 1 : 0;
        gn = make_float4(g0[t0], g0[gs + t0], g0[2 * gs + t0], g0[3 * gs + t0]);
    }
    __syncthreads();
    for (int s = 0; s < Tmax; ++s) {
        const int t = dir ? Tf - 1 - s : s;
        const float4 gcur = gn;
        if (fin && s + 1 < Tf) {
            const int tn = dir ? t - 1 : t + 1;
            gn = make_float4(g0[tn], g0[gs + tn], g0[2 * gs + tn], g0[3 * gs + tn]);
        }
        // all NS recurrences every step, finished ones included (their sums are never read): no branch sits between
        // the 2 NS independent chains of 64 dependent FMAs, so they interleave and hide each other's latency
        float a0[NS], a1[NS];
#pragma unroll
        for (int n = 0; n < NS; ++n) { a0[n] = 0.f; a1[n] = 0.f; }
#pragma unroll
        for (int k4 = 0; k4 < 16; ++k4) {
#pragma unroll
            for (int n = 0; n < NS; ++n) {
                const float4 hv = (reinterpret_cast<const float4*>(hbuf[n]) + 16 * pt)[k4];
                a0[n] += w[4 * k4 + 0].x * hv.x; a1[n] += w[4 * k4 + 0].y * hv.x;
                a0[n] += w[4 * k4 + 1].x * hv.y; a1[n] += w[4 * k4 + 1].y * hv.y;
                a0[n] += w[4 * k4 + 2].x * hv.z; a1[n] += w[4 * k4 + 2].y * hv.z;
                a0[n] += w[4 * k4 + 3].x * hv.w; a1[n] += w[4 * k4 + 3].y * hv.w;
            }
        }
#pragma unroll
        for (int n = 0; n < NS; ++n) part[n][pt][g2][jl] = make_float2(a0[n], a1[n]);
        __syncthreads();
        if (fin) {
            if (s < Tf) {                                      // wave-uniform
                const float2 i0 = part[fn][0][0][jl], i1 = part[fn][1][0][jl], i2 = part[fn][2][0][jl], i3 = part[fn][3][0][jl];
                const float2 o0 = part[fn][0][1][jl], o1 = part[fn][1][1][jl], o2 = part[fn][2][1][jl], o3 = part[fn][3][1][jl];
                const float ai = gcur.x + ((i0.x + i1.x) + (i2.x + i3.x));
                const float af = gcur.y + ((i0.y + i1.y) + (i2.y + i3.y));
                const float ag = gcur.z + ((o0.x + o1.x) + (o2.x + o3.x));
                const float ao = gcur.w + ((o0.y + o1.y) + (o2.y + o3.y));
                const float ig = sigmoidf_(ai), fg = sigmoidf_(af), gg = tanhf(ag), og = sigmoidf_(ao);
                c = fg * c + ig * gg;
                const float hn = og * tanhf(c);
                yo[t] = hn;
                // The new h travels as ONE 8-byte device-scope atomic store of (value, step): single-copy atomic, it
                // bypasses the non-coherent caches by itself, and a reader that sees the step tag has the value -- no
                // flag, no counter, no fence, no wait for the store's acknowledgement (a release fence would also
                // write back, and an acquire invalidate, the L2 on this chip: per workgroup and step).
                const unsigned long long word = ((unsigned long long)(unsigned)(s + 1) << 32) | (unsigned long long)__float_as_uint(hn);
                __hip_atomic_store(&H[((s & 1) * NS + fn) * 256 + j], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // every thread fetches one unit of each live sentence, spinning (bounded) until its step tag arrives
        if (tid < 256) {
            unsigned long long hw[NS];
            int spins = 0;
            bool all;
            do {
                all = true;
#pragma unroll
                for (int n = 0; n < NS; ++n) {
                    hw[n] = __hip_atomic_load(&H[((s & 1) * NS + n) * 256 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    all = all && (s >= Tn[n] || (unsigned)(hw[n] >> 32) == (unsigned)(s + 1));
                }
                if (!all && ++spins > spin_limit) { dead = 1; *err = 1; break; }
            } while (!all);
#pragma unroll
            for (int n = 0; n < NS; ++n) hbuf[n][tid] = __uint_as_float((unsigned)hw[n]);
        }
        __syncthreads();
        if (dead) break;
    }
}
void launch_lstm(const float* gx, long gbs, int gpitch, const float* whh, float* y, long ybs, int ypitch, LenSpec ls,
                 int B, hipStream_t s) {
    hipLaunchKernelGGL(lstm_kernel, dim3(2, B), dim3(1024), 0, s, gx, gbs, gpitch, whh, y, ybs, ypitch, ls);
}
// hx: exchange buffer of 8-byte (value, step) words [ceil(B / NS)][2 dirs][2 parities][NS][256], i.e. 2048 floats per
// sentence with B rounded up to 4, zeroed here; err: one device int (sticky); cus: CUs of the device
void launch_lstm_split(const float* gx, long gbs, int gpitch, const float* whh, float* y, long ybs, int ypitch, LenSpec ls,
                       int B, float* hx, int* err, int cus, int spin_limit, hipStream_t s) {
    (void)hipMemsetAsync(hx, 0, (size_t)((B + 3) / 4 * 4) * 2048 * sizeof(float), s);      // step tags start at 0
    if (8 * B <= cus)
        hipLaunchKernelGGL(lstm_split_kernel<1>, dim3(8 * B), dim3(512), 0, s, gx, gbs, gpitch, whh, y, ybs, ypitch, ls, B, hx, err, spin_limit);
    else if (8 * ((B + 1) / 2) <= cus)
        hipLaunchKernelGGL(lstm_split_kernel<2>, dim3(8 * ((B + 1) / 2)), dim3(512), 0, s, gx, gbs, gpitch, whh, y, ybs, ypitch, ls, B, hx, err, spin_limit);
    else
        hipLaunchKernelGGL(lstm_split_kernel<4>, dim3(8 * ((B + 3) / 4)), dim3(512), 0, s, gx, gbs, gpitch, whh, y, ybs, ypitch, ls, B, hx, err, spin_limit);
}

// ------------------------------------------------------------------------ small helpers
__global__ void fill_rows_kernel(float* y, long ybs, int ypitch, int r0, const float* v, int vstride, int voff,
                                 LenSpec ls) {
    const int b = blockIdx.z, i = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    if (t >= ls.len(b)) return;
    y[ls.base(b, ybs) + (long)(r0 + i) * ypitch + t] = v[(long)b * vstride + voff + i];
}
void launch_fill_rows(float* y, long ybs, int ypitch, int r0, int n, const float* v, int vstride, int voff,
                      LenSpec ls, int B, int Tmax, hipStream_t s) {
    hipLaunchKernelGGL(fill_rows_kernel, dim3(DIVUP(Tmax, 256), n, B), dim3(256), 0, s, y, ybs, ypitch, r0, v,
                       vstride, voff, ls);
}

__global__ void copy_rows_kernel(const float* x, long xbs, int xpitch, float* y, long ybs, int ypitch, LenSpec ls) {
    const int b = blockIdx.z, r = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    if (t >= ls.len(b)) return;
    y[(long)b * ybs + (long)r * ypitch + t] = x[(long)b * xbs + (long)r * xpitch + t];
}
void launch_copy_rows(const float* x, long xbs, int xpitch, float* y, long ybs, int ypitch, int rows, LenSpec ls,
                      int B, int Lmax, hipStream_t s) {
    hipLaunchKernelGGL(copy_rows_kernel, dim3(DIVUP(Lmax, 256), rows, B), dim3(256), 0, s, x, xbs, xpitch, y, ybs,
                       ypitch, ls);
}

// ----------------------------------------------------------------- durations (row a8)
// duration = sum_k sigmoid(logit_k) / speed ; round half-to-even ; clamp(min=1).  The 50
// sigmoids are rounded to fp32 one by one (as torch does), summed in fp64 and rounded once,
// so the sum is the correctly rounded value of what any fp32 summation order approximates.
__global__ void durations_kernel(const float* logits, long lbs, int lpitch, int ndur, const float* speed,
                                 const int* lens, int Tmax, int* dur, float* raw, const int* loff) {
    const int b = blockIdx.y, t = blockIdx.x * 64 + threadIdx.x;
    if (t >= Tmax) return;
    int d = 0;
    float r = 0.f;
    if (t < lens[b]) {
        const float* lg = logits + (loff ? (long)loff[b] : (long)b * lbs) + t;
        double s = 0.0;
        for (int k = 0; k < ndur; ++k) s += (double)sigmoidf_(lg[(long)k * lpitch]);
        r = __fdiv_rn((float)s, speed[b]);
        const float rr = rintf(r);
        d = rr < 1.0f ? 1 : (int)rr;
    }
    dur[(long)b * Tmax + t] = d;
    raw[(long)b * Tmax + t] = r;
}
void launch_durations(const float* logits, long lbs, int lpitch, int ndur, const float* speed, const int* lens,
                      int B, int Tmax, int* dur, float* raw, hipStream_t s, const int* loff) {
    hipLaunchKernelGGL(durations_kernel, dim3(DIVUP(Tmax, 64), B), dim3(64), 0, s, logits, lbs, lpitch, ndur, speed,
                       lens, Tmax, dur, raw, loff);
}

// -------------------------------------------------------- alignment + gather (row a9)
// The reference builds a one-hot [T, F] matrix and multiplies; that is a gather.
__global__ __launch_bounds__(256) void alignment_kernel(const int* dur, int Tmax, const int* lens, int* idx,
                                                       int ipitch, int* nfr) {
    __shared__ int cum[K82_MAXTOK_P1];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int T = lens[b];
    const int* d = dur + (long)b * Tmax;
    if (tid == 0) {
        int a = 0;
        for (int t = 0; t < T; ++t) { cum[t] = a; a += d[t]; }
        cum[T] = a;
        nfr[b] = a;
    }
    __syncthreads();
    int* ib = idx + (long)b * ipitch;
    for (int t = tid; t < T; t += 256) {
        const int s = cum[t], e = cum[t + 1];
        for (int f = s; f < e && f < ipitch; ++f) ib[f] = t;
    }
}
void launch_alignment(const int* dur, int Tmax, const int* lens, int B, int* idx, int ipitch, int* nfr,
                      hipStream_t s) {
    hipLaunchKernelGGL(alignment_kernel, dim3(B), dim3(256), 0, s, dur, Tmax, lens, idx, ipitch, nfr);
}

// Length regulation.  A thread owns 4 consecutive frames (one int4 of token indices, reused for GC channels)
// and writes one float4 per channel; the token-rate rows it reads are tiny and L2-resident, so the traffic is
// the frame-rate write.  Rows are padded to 4 floats: a float4 that starts inside the row is stored whole.
#define GATHER_GC 8
__global__ __launch_bounds__(256) void gather_kernel(const float* x, long xbs, int xpitch, int C, const int* idx, int ipitch,
                                                     const int* nfr, float* y, long ybs, int ypitch, const int* xoff) {
    const int b = blockIdx.z, c0 = blockIdx.y * GATHER_GC, f = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int F = nfr[b];
    if (f >= F) return;
    const int4 i4 = *reinterpret_cast<const int4*>(idx + (long)b * ipitch + f);
    const int i0 = i4.x, i1 = f + 1 < F ? i4.y : i0, i2 = f + 2 < F ? i4.z : i0, i3 = f + 3 < F ? i4.w : i0;
    const float* xb = x + (xoff ? (long)xoff[b] : (long)b * xbs);
    float* yb = y + (long)b * ybs + f;
#pragma unroll
    for (int k = 0; k < GATHER_GC; ++k) {
        const int c = min(c0 + k, C - 1);            // the clamped duplicate rewrites the same values: harmless
        const float* xr = xb + (long)c * xpitch;
        *reinterpret_cast<float4*>(yb + (long)c * ypitch) = make_float4(xr[i0], xr[i1], xr[i2], xr[i3]);
    }
}
void launch_gather(const float* x, long xbs, int xpitch, int C, const int* idx, int ipitch, const int* nfr, float* y,
                   long ybs, int ypitch, int B, int Fmax, hipStream_t s, const int* xoff) {
    // host-checked: ipitch, ypitch multiples of 4 (int4 / float4 accesses), Fmax <= ipitch
    hipLaunchKernelGGL(gather_kernel, dim3(DIVUP(DIVUP(Fmax, 4), 256), DIVUP(C, GATHER_GC), B), dim3(256), 0, s, x, xbs,
                       xpitch, C, idx, ipitch, nfr, y, ybs, ypitch, xoff);
}

// ------------------------------------------- depthwise ConvTranspose1d k3 s2 p1 op1 ("pool")
// out[2q] = w1*v[q] + b ; out[2q+1] = w2*v[q] + w0*v[q+1] + b ; v = leaky(adain(x))
__global__ void pool_up2_kernel(const float* x, long xbs, int xpitch, const float4* norm, int nstride, float slope,
                                const float* w, const float* bias, float* y, long ybs, int ypitch, LenSpec ls) {
    const int b = blockIdx.z, c = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
    const int L = ls.len(b);
    if (q >= L) return;
    const float4 nm = norm[(long)b * nstride + c];
    const float* xr = x + (long)b * xbs + (long)c * xpitch;
    float v0 = ((xr[q] - nm.x) * nm.y) * nm.z + nm.w;
    v0 = v0 < 0.f ? v0 * slope : v0;
    float v1 = 0.f;
    if (q + 1 < L) {
        v1 = ((xr[q + 1] - nm.x) * nm.y) * nm.z + nm.w;
        v1 = v1 < 0.f ? v1 * slope : v1;
    }
    const float w0 = w[c * 3 + 0], w1 = w[c * 3 + 1], w2 = w[c * 3 + 2], bb = bias[c];
    float* yr = y + (long)b * ybs + (long)c * ypitch;
    yr[2 * q] = w1 * v0 + bb;
    yr[2 * q + 1] = (w2 * v0 + w0 * v1) + bb;
}
void launch_pool_up2(const float* x, long xbs, int xpitch, const float4* norm, int nstride, float slope,
                     const float* w, const float* bias, float* y, long ybs, int ypitch, LenSpec ls, int C, int B,
                     int Lmax, hipStream_t s) {
    hipLaunchKernelGGL(pool_up2_kernel, dim3(DIVUP(Lmax, 256), C, B), dim3(256), 0, s, x, xbs, xpitch, norm, nstride,
                       slope, w, bias, y, ybs, ypitch, ls);
}

// Conv1d(1,1,k=3,stride=2,pad=1) of the F0 / N curves: [2F] -> [F]
__global__ void conv_k3s2_kernel(const float* x, long xbs, const float* w3b, float* y, long ybs, const int* nfr) {
    const int b = blockIdx.y, f = blockIdx.x * 256 + threadIdx.x;
    const int F = nfr[b];
    if (f >= F) return;
    const float* xr = x + (long)b * xbs;
    const float xm = f > 0 ? xr[2 * f - 1] : 0.f;
    y[(long)b * ybs + f] = ((w3b[0] * xm + w3b[1] * xr[2 * f]) + w3b[2] * xr[2 * f + 1]) + w3b[3];
}
void launch_conv_k3s2(const float* x, long xbs, const float* w3b, float* y, long ybs, const int* nfr, int B, int Fmax,
                      hipStream_t s) {
    hipLaunchKernelGGL(conv_k3s2_kernel, dim3(DIVUP(Fmax, 256), B), dim3(256), 0, s, x, xbs, w3b, y, ybs, nfr);
}

// column 0 of the reflection-padded stage-1 tensor: x[0] = ups[1] + xsrc[0] = (x[2] - xsrc[2]) + xsrc[0]
__global__ void reflect_fix_kernel(float* y, long ybs, int ypitch, const float* xs, long xbs, int xpitch, int C) {
    const int b = blockIdx.y, c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    float* yr = y + (long)b * ybs + (long)c * ypitch;
    const float* sr = xs + (long)b * xbs + (long)c * xpitch;
    yr[0] = (yr[2] - sr[2]) + sr[0];
}
void launch_reflect_fix(float* y, long ybs, int ypitch, const float* xs, long xbs, int xpitch, int C, int B,
                        hipStream_t s) {
    hipLaunchKernelGGL(reflect_fix_kernel, dim3(DIVUP(C, 64), B), dim3(64), 0, s, y, ybs, ypitch, xs, xbs, xpitch, C);
}

// Phase interleave of a transposed convolution computed as a conv over s * cout phase rows:
//     y[b][co][s t + r + off] = yp[b][r cout + co][t] + res[b][co][s t + r + off]          (t < Lin_b, r < s)
// The conv writes its phase-major output with the wide (128-byte row) epilogue, and this pass does the stride-s
// scatter through LDS so that both its loads and its stores are whole contiguous rows -- the conv's own interleaving
// epilogue stored single dwords s floats apart (MFMA pipe busy 11 % in the bf16x3 mode, 56 % in fp32).  The addition
// is the same single fp32 add the fused epilogue performed: identical bits.
template <int S>
__global__ __launch_bounds__(256) void phase_interleave_kernel(const float* yp, long pbs, int ppitch, float* y, long ybs, int ypitch,
                                                              const float* res, long rbs, int rpitch, LenSpec lin, int cout, int off) {
    __shared__ float tile[256 * S];                                  // [t][r]: the output order of this block's span
    const int b = blockIdx.z, co = blockIdx.y, t0 = blockIdx.x * 256, tid = threadIdx.x;
    const int L = lin.len(b);
    if (t0 >= L) return;
    const int n = min(256, L - t0);
    if (tid < n) {
        const float* src = yp + (long)b * pbs + (long)co * ppitch + t0 + tid;
#pragma unroll
        for (int r = 0; r < S; ++r) tile[tid * S + r] = src[(long)r * cout * ppitch];
    }
    __syncthreads();
    const long yo = (long)b * ybs + (long)co * ypitch + (long)S * t0 + off;
    const long ro = (long)b * rbs + (long)co * rpitch + (long)S * t0 + off;
    for (int o = tid; o < n * S; o += 256) y[yo + o] = tile[o] + res[ro + o];
}
int launch_phase_interleave(const float* yp, long pbs, int ppitch, float* y, long ybs, int ypitch, const float* res, long rbs,
                             int rpitch, LenSpec lin, int cout, int s, int off, int B, int Lmax, hipStream_t st) {
    const dim3 g(DIVUP(Lmax, 256), cout, B);
    if (s == 10)
        hipLaunchKernelGGL(phase_interleave_kernel<10>, g, dim3(256), 0, st, yp, pbs, ppitch, y, ybs, ypitch, res, rbs, rpitch, lin, cout, off);
    else if (s == 6)
        hipLaunchKernelGGL(phase_interleave_kernel<6>, g, dim3(256), 0, st, yp, pbs, ppitch, y, ybs, ypitch, res, rbs, rpitch, lin, cout, off);
    else
        return -1;                                                   // only the model's two rates are built
    return 0;
}

// clip(x,-1,1)*32767 -> int16, truncating like numpy's astype('<i2')
// (basic.tts_kokoro_2_play_pause_works.py:25)
__global__ void pcm16_kernel(const float* src, int16_t* dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = src[i];
    v = fminf(fmaxf(v, -1.0f), 1.0f) * 32767.0f;
    dst[i] = (int16_t)v;
}
void launch_pcm16(const float* src, int16_t* dst, int64_t n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(pcm16_kernel, dim3((unsigned)DIVUP(n, 256)), dim3(256), 0, s, src, dst, n);
}
