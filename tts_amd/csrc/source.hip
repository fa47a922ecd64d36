// Harmonic source (SineGen + SourceModuleHnNSF), STFT of the source, and the final iSTFT
// (SURVEY.md 8a rows a14, a15, a18).  The phase path mirrors oracle/kokoro_ref.py::sinegen_phase
// operation by operation with explicitly rounded fp32 intrinsics (no FMA contraction): sin() is
// taken of phases of 1e5..1e6 rad, where one fp32 ulp of the ARGUMENT is 0.01-0.06 rad, so the
// argument has to be bit-identical, not merely close.
#include "kernels.h"
#include <math.h>

// HIP implements __fmul_rn/__fadd_rn as plain * and +, which -ffp-contract=fast (the hipcc default)
// may fuse into an FMA: one rounding fewer = 1 ulp of a 1e4-rad phase.  Contraction is switched
// off for this whole file (and again on its compile line in the Makefile).
#pragma clang fp contract(off)

#define DIVUP(a, b) (((a) + (b)-1) / (b))

__constant__ float c_win[20];     // periodic hann, float(0.5 - 0.5 cos(2 pi n / 20))
__constant__ float c_cos[20];     // cos(2 pi j / 20)
__constant__ float c_sin[20];     // sin(2 pi j / 20)

int source_init_constants() {
    float w[20], c[20], s[20];
    for (int n = 0; n < 20; ++n) {
        w[n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * n / 20.0));
        c[n] = (float)cos(2.0 * M_PI * n / 20.0);
        s[n] = (float)sin(2.0 * M_PI * n / 20.0);
    }
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_win), w, sizeof(w)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_cos), c, sizeof(c)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_sin), s, sizeof(s)) != hipSuccess) return -1;
    return 0;
}

#define PI_F 3.14159274101257324f   // float(pi), what torch multiplies by

// ---- low-rate phase: ph[b][h][i] = float(cumsum_double(rad)) * 2 * pi * 300 ------------------
// The cumulative sum is sequential in fp64 and rounded to fp32 per element, exactly as torch's
// CPU cumsum does for float tensors.  2F <= a few thousand steps; 9 lanes per sentence.
__global__ __launch_bounds__(64) void src_phase_kernel(const float* f0, long fbs, const int* nfr, float* ph, long pbs,
                                                      int ppitch) {
    const int b = blockIdx.x, h = threadIdx.x;
    if (h >= K82_HARM) return;
    const int n = 2 * nfr[b];
    const float* f = f0 + (long)b * fbs;
    float* out = ph + (long)b * pbs + (long)h * ppitch;
    const float hm = (float)(h + 1);
    double cs = 0.0;
    for (int i = 0; i < n; ++i) {
        const float fn = __fmul_rn(f[i], hm);
        const float q = __fdiv_rn(fn, 24000.0f);
        float r = fmodf(q, 1.0f);
        if (r < 0.f) r = __fadd_rn(r, 1.0f);
        cs += (double)r;
        const float c = (float)cs;
        out[i] = __fmul_rn(__fmul_rn(__fmul_rn(c, 2.0f), PI_F), 300.0f);
    }
}
void launch_src_phase(const float* f0, long fbs, const int* nfr, int B, float* ph, long pbs, int ppitch,
                      hipStream_t s) {
    hipLaunchKernelGGL(src_phase_kernel, dim3(B), dim3(64), 0, s, f0, fbs, nfr, ph, pbs, ppitch);
}

// ---- counter-based N(0,1): splitmix64 + Box-Muller (oracle twin: kokoro_ref.counter_noise) ---
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float k82_noise_normal(uint64_t base, uint64_t n, int h) {
    const uint64_t x = splitmix64(base + n * 16ull + (uint64_t)h);
    const float u1 = (float)(((double)(x >> 40) + 1.0) / 16777216.0);
    const float u2 = (float)((double)((x >> 8) & 0xFFFFFFull) / 16777216.0);
    const float r = sqrtf(__fmul_rn(-2.0f, logf(u1)));
    return __fmul_rn(r, cosf(__fmul_rn(__fmul_rn(2.0f, PI_F), u2)));
}

// ---- per-sample harmonic source: x300 linear phase interpolation, 9 sines, uv gate, noise,
// Linear(9->1), tanh.  One thread per output sample (600 per frame).
__global__ __launch_bounds__(256) void src_har_kernel(const float* f0, long fbs, const float* ph, long pbs, int ppitch,
                                                     const int* nfr, const float* noise, const int64_t* noise_off,
                                                     const uint64_t* seeds, const float* lin_w, const float* lin_b,
                                                     float* har_src, long hbs) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int F = nfr[b];
    if (n >= 600 * F) return;
    const int nlo = 2 * F;
    const float scale = (float)(1.0 / 300.0);
    // torch's CPU upsample_linear1d fuses these two multiply-adds (oracle: sinegen_phase)
    float src = __fmaf_rn(scale, __fadd_rn((float)n, 0.5f), -0.5f);
    src = fmaxf(src, 0.0f);
    const int i0 = (int)floorf(src);
    const float lam = __fsub_rn(src, (float)i0);
    const int i1 = min(i0 + 1, nlo - 1);
    const float w0 = __fsub_rn(1.0f, lam);
    const float f0v = f0[(long)b * fbs + n / 300];
    const float uv = f0v > 10.0f ? 1.0f : 0.0f;
    const float namp = f0v > 10.0f ? 0.003f : (0.1f / 3.0f);
    const float* pb = ph + (long)b * pbs;
    const uint64_t base = splitmix64(seeds ? seeds[b] : 0ull);
    const float* nz = noise ? noise + (noise_off[b] + n) * K82_HARM : nullptr;
    float acc = 0.f;
#pragma unroll
    for (int h = 0; h < K82_HARM; ++h) {
        const float c = __fmul_rn(lam, pb[(long)h * ppitch + i1]);
        const float phase = __fmaf_rn(w0, pb[(long)h * ppitch + i0], c);
        const float sine = __fmul_rn(sinf(phase), 0.1f);
        const float z = nz ? nz[h] : k82_noise_normal(base, (uint64_t)n, h);
        const float sw = __fadd_rn(__fmul_rn(sine, uv), __fmul_rn(namp, z));
        acc += lin_w[h] * sw;
    }
    har_src[(long)b * hbs + n] = tanhf(acc + lin_b[0]);
}
void launch_src_har(const float* f0, long fbs, const float* ph, long pbs, int ppitch, const int* nfr,
                    const float* noise, const int64_t* noise_off, const uint64_t* seeds, const float* lin_w,
                    const float* lin_b, float* har_src, long hbs, int B, int Fmax, hipStream_t s) {
    hipLaunchKernelGGL(src_har_kernel, dim3(DIVUP(600 * Fmax, 256), B), dim3(256), 0, s, f0, fbs, ph, pbs, ppitch, nfr,
                       noise, noise_off, seeds, lin_w, lin_b, har_src, hbs);
}

// ---- STFT(n_fft 20, hop 5, hann, center=True/reflect): one thread per frame --------------------
// Writes har[b][0..10][m] = |X_k|, har[b][11..21][m] = angle(X_k) and the stride-6 polyphase
// image harp[b][ci*6 + r][q] = har[ci][6q + r - 3] that turns noise_convs[0] (k12, s6, p3) into
// a 2-tap stride-1 conv for the MFMA kernel.  DC and Nyquist bins have an exactly zero
// imaginary part (+0, as the r2c FFT returns), so their phase is 0 or +pi; so has frame 0,
// which is exactly symmetric under reflect padding (its FFT imaginary part is rounding noise).
__global__ __launch_bounds__(256) void stft_kernel(const float* har_src, long hbs, const int* nfr, float* har, long bs1,
                                                  int p1, float* harp, long bs0, int p0) {
    const int b = blockIdx.y;
    const int m = blockIdx.x * 256 + threadIdx.x;
    const int F = nfr[b];
    const int L = 600 * F;
    if (m > 120 * F) return;
    const float* x = har_src + (long)b * hbs;
    float v[20];
#pragma unroll
    for (int n = 0; n < 20; ++n) {
        int i = 5 * m - 10 + n;
        if (i < 0) i = -i;
        if (i >= L) i = 2 * (L - 1) - i;
        v[n] = c_win[n] * x[i];
    }
    float* hb = har + (long)b * bs1 + m;
    const int q = (m + 3) / 6, r = (m + 3) % 6;
    float* pb = harp + (long)b * bs0 + (long)r * p0 + q;
#pragma unroll
    for (int k = 0; k <= 10; ++k) {
        float re = 0.f, im = 0.f;
#pragma unroll
        for (int n = 0; n < 20; ++n) {
            const int j = (k * n) % 20;
            re += v[n] * c_cos[j];
            im -= v[n] * c_sin[j];
        }
        if (k == 0 || k == 10 || m == 0) im = 0.0f;   // m == 0: exactly symmetric frame, see oracle stft()
        const float mag = sqrtf(re * re + im * im);
        const float ang = atan2f(im, re);
        hb[(long)k * p1] = mag;
        hb[(long)(11 + k) * p1] = ang;
        pb[(long)(k * 6) * p0] = mag;
        pb[(long)((11 + k) * 6) * p0] = ang;
    }
}
void launch_stft(const float* har_src, long hbs, const int* nfr, int B, int Fmax, float* har, long bs1, int p1,
                 float* harp, long bs0, int p0, hipStream_t s) {
    hipLaunchKernelGGL(stft_kernel, dim3(DIVUP(120 * Fmax + 1, 256), B), dim3(256), 0, s, har_src, hbs, nfr, har, bs1,
                       p1, harp, bs0, p0);
}

// ---- iSTFT (row a18) fused with conv_post's exp / sin heads ------------------------------------
// One thread per hop m: the 5 output samples 5m..5m+4 (n = 5m+10+i in the un-trimmed signal)
// receive frames m-1..m+2 at window positions j = 10+i-5*dm.  irfft-20 is evaluated directly:
// y[j] = (X0 + (-1)^j X10 + 2 sum_{k=1..9} (Re X_k cos(2 pi k j/20) - Im X_k sin(2 pi k j/20))) / 20.
// HBM traffic: 22 floats read per frame (neighbour re-reads hit L1/L2) + 5 floats written.
__global__ __launch_bounds__(256) void istft_kernel(const float* cp, long cbs, int cpitch, const int* nfr, float* audio,
                                                   const int64_t* aoff) {
    // A block owns 256 consecutive hops; the spectra of its 259 frames (one before, two after) are turned into
    // (re, im) ONCE into LDS -- exp, sin and sincos are the cost of this kernel, and every frame feeds four hops.
    __shared__ float sre[11][260], sim[11][260];
    const int b = blockIdx.y;
    const int m0 = blockIdx.x * 256;
    const int F = nfr[b];
    if (m0 >= 120 * F) return;                      // block-uniform
    const int NF = 120 * F + 1;
    const float* c = cp + (long)b * cbs;
    for (int f = threadIdx.x; f < 259; f += 256) {
        const int mm = m0 - 1 + f;
        const bool ok = mm >= 0 && mm < NF;
        const int mc = min(max(mm, 0), NF - 1);
#pragma unroll
        for (int k = 0; k <= 10; ++k) {
            const float spec = expf(c[(long)k * cpitch + mc]);
            const float phs = sinf(c[(long)(11 + k) * cpitch + mc]);
            float sn, cs;
            sincosf(phs, &sn, &cs);
            sre[k][f] = ok ? spec * cs : 0.f;
            sim[k][f] = ok ? spec * sn : 0.f;
        }
    }
    __syncthreads();
    const int m = m0 + threadIdx.x;
    if (m >= 120 * F) return;
    float out[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, env[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dm = -1; dm <= 2; ++dm) {
        const int mm = m + dm;
        if (mm < 0 || mm >= NF) continue;
        const int f = threadIdx.x + dm + 1;
        float re[11], im[11];
#pragma unroll
        for (int k = 0; k <= 10; ++k) { re[k] = sre[k][f]; im[k] = sim[k][f]; }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int j = 10 + i - 5 * dm;          // 0..19 by construction
            float a = re[0] + ((j & 1) ? -re[10] : re[10]);
#pragma unroll
            for (int k = 1; k <= 9; ++k) {
                const int tw = (k * j) % 20;
                a += 2.0f * (re[k] * c_cos[tw] - im[k] * c_sin[tw]);
            }
            const float w = c_win[j];
            out[i] += w * (a * 0.05f);
            env[i] += w * w;
        }
    }
    float* o = audio + aoff[b] + 5 * (int64_t)m;
#pragma unroll
    for (int i = 0; i < 5; ++i) o[i] = out[i] / env[i];
}
void launch_istft(const float* cp, long cbs, int cpitch, const int* nfr, int B, int Fmax, float* audio,
                  const int64_t* aoff, hipStream_t s) {
    hipLaunchKernelGGL(istft_kernel, dim3(DIVUP(120 * Fmax, 256), B), dim3(256), 0, s, cp, cbs, cpitch, nfr, audio,
                       aoff);
}
