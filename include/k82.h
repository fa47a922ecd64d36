/* k82 -- C-ABI of the MI355X-native Kokoro-82M synthesis engine (libk82.so).
 *
 * This is the drop-in boundary for the ONE hot path of fastrizwaan/tts: what the reference
 * reaches through `kokoro_onnx.Kokoro(model, voices).create(text, voice, speed, lang)`
 * (speech7.py:82,127; basic.tts_kokoro.py:17,30), i.e. one `sess.run` of kokoro-v1.0.onnx with
 * inputs input_ids i64[1,T], style f32[1,256], speed f32[1] -> audio f32[600*F]
 * (SURVEY.md 8a row a4).  Plain pointers and sizes only; no torch types; never throws, never
 * aborts: every entry point returns 0 or a negative K82_E* code and k82_last_error() explains.
 *
 * Pointer arguments documented "host or device" are classified with hipPointerGetAttributes.
 *
 * Threads.  A handle serialises its own calls with a mutex, so EACH call is safe from any thread.
 * k82_synthesize is one call and therefore atomic: it is what concurrent callers use (the reference
 * runs two pool threads on one Kokoro object, epubviewer_06_latest_1_nov_2025_theme.py:407,692,715,
 * and a Timer thread, speech7.py:178).  The two-phase form is NOT atomic as a pair: k82_durations
 * leaves the token-rate state of its batch inside the handle and k82_decode consumes whatever state
 * the handle holds, so threads that use the pair on one handle must hold their own lock across it
 * (tts_amd.engine.Engine.session() does; every Python front door goes through it).
 *
 * Sharing the GPU.  For batches of up to CUs/2 sentences the BiLSTM recurrences run on a kernel whose
 * workgroups exchange h every step and must be co-resident, which holds when the handle has the chip
 * to itself.  When another process or handle occupies CUs (the reference pre-rolls in the parent while
 * a producer child synthesizes: basic.tts_kokoro_2_play_pause_works.py:34-41,136-146) a bounded wait
 * runs out, the call re-runs that phase on the streaming BiLSTM kernel (same bits) and the handle
 * stays on it for the next 64 phases; K82_LSTM_SPLIT_MAX_B=0 in the environment selects the streaming
 * kernel from the start.  No call fails or returns different samples because the GPU is shared.
 *
 * HIP is initialised lazily by k82_create in the calling process (construction inside a
 * forked/spawned child is fine: basic.tts_kokoro_2_play_pause_works.py:34-41).
 */
#ifndef K82_H
#define K82_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define K82_OK 0
#define K82_EINVAL (-22)   /* bad argument / shape */
#define K82_ENOMEM (-12)   /* device or host allocation failed */
#define K82_ENOENT (-2)    /* missing weight / unknown name */
#define K82_ENOSPC (-28)   /* caller's output buffer too small */
#define K82_EHIP (-5)      /* a HIP runtime call failed */
#define K82_ESTATE (-1)    /* call order violated (e.g. decode before durations) */

#define K82_SAMPLE_RATE 24000
#define K82_FRAME_SAMPLES 600
#define K82_STYLE_DIM 256
#define K82_MAX_TOKENS 512 /* 510 phonemes + 2 pad tokens */
#define K82_HARMONICS 9

typedef struct k82_handle k82_handle;

/* Architecture constants of hexgrad/Kokoro-82M config.json.  The engine is
 * written for exactly this architecture; k82_create rejects anything else. */
typedef struct k82_config {
    int32_t n_token;          /* 178 */
    int32_t hidden_dim;       /* 512 */
    int32_t style_dim;        /* 128 */
    int32_t n_layer;          /* 3   */
    int32_t max_dur;          /* 50  */
    int32_t text_kernel;      /* 5   */
    int32_t bert_hidden;      /* 768 */
    int32_t bert_heads;       /* 12  */
    int32_t bert_inter;       /* 2048 */
    int32_t bert_layers;      /* 12  */
    int32_t bert_max_pos;     /* 512 */
    int32_t bert_embed;       /* 128 */
} k82_config;

/* Source-noise control for the harmonic generator (SineGen's randn).
 * noise != NULL: explicit N(0,1) values, f32[sum_b 600*F_b][9], sentences concatenated in
 *   batch order (host or device).  Used for sample-exact parity against the oracle.
 * noise == NULL: the engine's counter-based generator, keyed per sentence by seeds[b]
 *   (host, u64[B]; NULL = seed 0 for every sentence). */
typedef struct k82_noise {
    const float* noise;
    const uint64_t* seeds;
} k82_noise;

/* replaces: Kokoro.__init__ -> ort.InferenceSession(model_path) (speech7.py:82). */
int k82_create(const k82_config* cfg, int device, k82_handle** out);
int k82_destroy(k82_handle* h);
/* A second execution context on the SAME weights: own stream, own workspace, own phase-1 state; nothing is uploaded
 * again (the fork reads the parent's device memory).  Calls on parent and fork do not serialise against each other, so
 * one host thread per handle overlaps one batch's token-rate phase (k82_durations: small launches and a device -> host
 * sync) with another batch's frame-rate phase (k82_decode) on one GPU; a batch gives the same bits on either handle.
 * The fork inherits the precision mode; it must be destroyed before its parent (k82_destroy(parent) returns K82_ESTATE
 * while forks live); a fork cannot be forked.
 * replaces: the reference's two worker threads on one Kokoro object (epubviewer_06_latest_1_nov_2025_theme.py:407,692,715),
 * which onnxruntime serves with concurrent sess.run calls on one InferenceSession. */
int k82_fork(k82_handle* parent, k82_handle** out);
const char* k82_last_error(const k82_handle* h); /* h may be NULL: last create error */

/* Weight upload.  `name` is "<submodule>.<state-dict key>" with weight-norm already folded
 * (e.g. "decoder.generator.ups.0.weight"), data is host f32 in PyTorch layout.  The layer
 * table in csrc/engine.hip names every tensor it needs; k82_finalize fails with K82_ENOENT
 * (naming the tensor) if one is missing or mis-shaped.
 * replaces: the weights embedded in kokoro-v1.0.onnx (io.github.fastrizwaan.tts.yaml:964-976). */
int k82_set_tensor(k82_handle* h, const char* name, const float* data, const int64_t* shape, int ndim);
int k82_finalize(k82_handle* h);

/* Phase 1 (token rate): ALBERT -> bert_encoder -> DurationEncoder -> LSTM -> duration_proj ->
 * round(sum(sigmoid)/speed).clamp(1).  Keeps the token-rate state inside the handle for
 * k82_decode.
 *   ids      i64[B][Tmax]  host or device; each row is [0, *phoneme ids, 0] then padding.  Host ids outside
 *                          [0, n_token) are rejected with K82_EINVAL; device ids cannot be inspected without a
 *                          copy and are clamped to the table inside the kernels (never an out-of-bounds read)
 *   lens     i32[B]        host; tokens per row including the two pad tokens (2..512)
 *   ref_s    f32[B][256]   host or device; [s_dec(128) | s_pros(128)]
 *   speed    f32[B]        host (NULL = 1.0)
 *   pred_dur i32[B][Tmax]  host out (may be NULL); 0 beyond lens[b]
 *   n_frames i64[B]        host out (may be NULL); F_b = sum of pred_dur
 *   forced_dur i32[B][Tmax] host (NULL = use predictions): test hook, SURVEY.md 8d cfg2/cfg4
 * replaces: the duration half of sess.run. */
int k82_durations(k82_handle* h, const int64_t* ids, const int32_t* lens, int B, int Tmax,
                  const float* ref_s, const float* speed, const int32_t* forced_dur,
                  int32_t* pred_dur, int64_t* n_frames, void* stream);

/* Phase 2 (frame rate): length regulation, F0/N predictor, text encoder, decoder, iSTFTNet
 * generator for the batch of the preceding k82_durations call.
 *   audio     f32[capacity] host or device out; sentence b occupies
 *             [audio_off[b], audio_off[b+1]) with audio_off[b+1]-audio_off[b] = 600*F_b
 *   audio_off i64[B+1] host out
 * Returns K82_ENOSPC (audio_off filled) if capacity < audio_off[B].
 * replaces: the synthesis half of sess.run. */
int k82_decode(k82_handle* h, const k82_noise* noise, float* audio, int64_t capacity,
               int64_t* audio_off, void* stream);

/* Convenience: k82_durations + k82_decode in one call (what one Kokoro.create chunk does). */
int k82_synthesize(k82_handle* h, const int64_t* ids, const int32_t* lens, int B, int Tmax,
                   const float* ref_s, const float* speed, const k82_noise* noise,
                   int32_t* pred_dur, float* audio, int64_t capacity, int64_t* audio_off,
                   void* stream);

/* Pre-size the handle's workspace for batches of up to B sentences of up to Tmax tokens and `frames` predicted
 * frames each (1 frame = 600 samples), so that no pass up to that size allocates: SURVEY.md 8b "handle owns weights
 * + workspace arena sized at create for (Bmax, Tmax, Fmax); no hidden allocations on the hot path".  Without it the
 * arenas grow on the first pass of each new size (one hipMalloc behind a sync) and never shrink.  Optional; call
 * after k82_finalize; invalidates the state a preceding k82_durations left.  The reference has no counterpart (its
 * runtime allocates per call). */
int k82_reserve(k82_handle* h, int B, int Tmax, int64_t frames);

/* Compute precision of the frame-rate convolutions (decoder + generator; SURVEY.md 8d cfg3 "bf16 decoder").
 *   mode 0 (default): every convolution on the exact-fp32 matrix instruction (v_mfma_f32_32x32x2_f32).
 *   mode 1: the decoder / generator convolutions round weights and staged activations to bf16 and use
 *           v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the resblock-internal tensor of every generator
 *           AdaINResBlock1 crosses HBM as bf16 channel pairs, every other activation (residual paths included) stays
 *           fp32; durations, the F0/N curves, the harmonic source, conv_post and the iSTFT stay fp32 (durations
 *           remain bit-exact).  Waveform SNR against mode 0: 46 dB.
 *   mode 2: "bf16x3" -- the same convolutions with every operand split into two bf16 numbers (hi + lo, 2^-17
 *           relative) and three bf16 MFMAs per product (a_lo b_hi + a_hi b_lo + a_hi b_hi, fp32 accumulation):
 *           fp32-grade results (waveform within 1e-5 of the CPU oracle; the fp32 bar is 1e-3) at 2.2x the mode-0
 *           throughput.  Everything mode 1 keeps in fp32 stays on the exact-fp32 instruction here too.
 * No counterpart in the reference (onnxruntime runs the fp32 graph); north_star's bf16 bar is SNR >= 40 dB.
 * Returns K82_EINVAL for any other mode. */
int k82_set_precision(k82_handle* h, int mode);

/* float -> 16-bit PCM as the reference callers do: clip(x,-1,1)*32767 -> '<i2'
 * (basic.tts_kokoro_2_play_pause_works.py:25).  src/dst host or device. */
int k82_pcm16(k82_handle* h, const float* src, int16_t* dst, int64_t n, void* stream);

/* k82_pcm16 for callers that cannot allocate.  No counterpart in the reference: it exists for callers that must
 * not reach hipMalloc -- tight serving loops, harnesses that forbid allocation.  (The property is "no allocation", not
 * "capturable": the call queries its pointers, may copy from pageable memory and synchronises the stream, none of
 * which is legal on a stream that is being captured into a HIP graph.)
 * Same arguments and result as k82_pcm16 (bit-identical output), but the device staging that k82_pcm16 takes from
 * the handle's workspace arena (which may grow) is a fixed static device array inside libk82.so:
 *   - at most K82_PCM16_STATIC_MAX samples per call, else K82_EINVAL and nothing is written;
 *   - the call performs no hipMalloc / hipFree / arena growth; the array's address is resolved once per device in
 *     k82_create;
 *   - every GPU the library runs on holds its own 24 MiB copy until the process ends; all handles of a process on
 *     one GPU share that GPU's copy: calls are serialised per GPU inside the library (a process-wide mutex per GPU,
 *     held across the call's stream synchronize -- two handles on one GPU convert one after the other), and the array
 *     keeps the last call's contents (each call overwrites exactly the [0, n) prefix it then reads). */
#define K82_PCM16_STATIC_MAX (4 * 1024 * 1024)
int k82_pcm16_static(k82_handle* h, const float* src, int16_t* dst, int64_t n, void* stream);

/* ---- measurement and test hooks (no counterpart in the reference) ---------------------- */

/* Per-kernel-family timing with HIP events on the engine's stream (bench.py roofline leg).
 * enable=1 starts recording for subsequent calls; k82_prof_read synchronises and returns the
 * accumulated milliseconds / launch count / algorithmic FLOPs (or bytes) of family `name`
 * ("conv_mfma", "conv_mfma_2x2", "conv_mfma_1x4", "lstm", "instnorm", "attention", "istft", "gather", "source",
 * "other") since enable.  The name "lstm_fallbacks" returns, in *launches, how many phases this handle has re-run
 * on the streaming BiLSTM kernel since it was created (see "Sharing the GPU" above). */
int k82_prof_enable(k82_handle* h, int enable);
int k82_prof_read(k82_handle* h, const char* name, double* ms, int64_t* launches, double* work);

/* Debug taps: with enable=1 every stage tensor is copied to the host after it is produced
 * (slow; parity tests only).  k82_debug_info gives dims {B, C, pitch}; k82_debug_copy copies
 * B*C*pitch floats. */
int k82_debug_enable(k82_handle* h, int enable);
int k82_debug_info(k82_handle* h, const char* name, int64_t dims[3]);
int k82_debug_copy(k82_handle* h, const char* name, float* dst);

/* Teacher forcing for parity tests against committed golden vectors (tests/golden/): replaces a stage tensor of the
 * NEXT k82_decode calls, right after the engine computed it, by host data -- "F0" / "N": f32[1][2F]; "har":
 * f32[22][120F+1] (magnitude | phase of the source STFT).  Batches of one sentence only; data == NULL removes the
 * override.  The model integrates F0 into a phase and takes angle() of near-empty STFT bins, so two correct fp32
 * implementations drift apart downstream of these three tensors (DESIGN.md "numerical conditioning"); with them
 * pinned, the waveform is comparable sample by sample. */
int k82_debug_force(k82_handle* h, const char* name, const float* data, int64_t rows, int64_t cols);

/* Stand-alone launch of the hot conv kernel (unit parity tests and kernel benchmarks).
 * y[b][co][t] = out_act( sum_{ci,k} w[co][ci][k] * act(x[b][ci][t + k*dil - pad]) + bias[co] )
 * x f32[B][cin][L], w f32[cout][cin][taps], y f32[B][cout][L], all host; act: 0 none,
 * 1 leaky(slope), 2 snake(alpha[ci]).  iters>1 repeats the launch and reports the average
 * device time of one launch in *ms (may be NULL). */
int k82_op_conv1d(k82_handle* h, const float* x, const float* w, const float* bias, float* y,
                  int B, int cin, int cout, int L, int taps, int dil, int act, float slope,
                  const float* alpha, int iters, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* K82_H */
