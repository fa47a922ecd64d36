// k82 engine: weight tables, workspace arenas, the two-phase Kokoro-82M forward, and the C-ABI
// declared in include/k82.h.  Everything here is host orchestration of the kernels in
// conv_mfma.hip / ops.hip / source.hip on one HIP stream; no kernel is launched from anywhere
// else.  Layout, masking and naming conventions: kernels.h.
#include "kernels.h"
#include "../../include/k82.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

struct K82Error : std::runtime_error {
    int code;
    K82Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
#define HIPCHK(x)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) throw K82Error(K82_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define REQUIRE(c, code, msg)                  \
    do {                                       \
        if (!(c)) throw K82Error(code, msg);   \
    } while (0)

inline int ru(int a, int m) { return (a + m - 1) / m * m; }

std::string g_create_error;

// ---- k82_pcm16_static: the one entry point that allocates nothing (include/k82.h) ------------------------------
// Staging for host -> device -> host conversion lives in a file-scope device array instead of the workspace arena.
// Every GPU the library runs on holds its own copy of the array at its own address until the process ends; host
// code never uses the array's NAME as an address (that is a host-side stand-in) -- it asks the runtime once per
// device, with that device current, and keeps the answer here.
constexpr int64_t kPcmStaticMax = (int64_t)4 << 20;                  // samples per call: 16 MiB of float + 8 MiB of int16
constexpr size_t kPcmStaticBytes = (size_t)kPcmStaticMax * (sizeof(float) + sizeof(int16_t));
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu[kMaxDevices];                                 // calls on one GPU share that GPU's array
void* g_scratch_ptr[kMaxDevices] = {nullptr};
std::mutex g_scratch_init_mu;

// grow-only chunked device arena: no hipMalloc in steady state
// Grow-only bump allocator for activations.  A pass (one k82_durations / k82_decode call) that outgrows the
// arena adds chunks on demand; the next reset() folds them into ONE chunk sized for the largest pass seen, so
// a stream of differently sized batches neither fragments nor accumulates device memory (steady state: no
// hipMalloc at all on the hot path).
struct Arena {
    struct Chunk { char* p; size_t cap; };
    std::vector<Chunk> chunks;
    size_t cur = 0, off = 0, chunk_default = (size_t)256 << 20;
    size_t pass_bytes = 0, high = 0;
    // Folds a pass that had to grow into one chunk.  The caller guarantees that nothing of the pass is still
    // running (decode() calls it behind its stream synchronize, so the pass that grew pays for the fold, not the
    // next one).
    void fold() {
        high = std::max(high, pass_bytes);
        if (chunks.size() <= 1) return;
        for (auto& c : chunks) (void)hipFree(c.p);
        chunks.clear();
        Chunk c;
        c.cap = high + high / 8 + ((size_t)64 << 20);
        if (hipMalloc((void**)&c.p, c.cap) == hipSuccess) chunks.push_back(c);
        else (void)hipGetLastError();                       // fall back to growing on demand
        cur = 0; off = 0;
    }
    void reset() {
        if (chunks.size() > 1) {
            (void)hipDeviceSynchronize();                   // nothing of the previous pass may still be running
            fold();
        }
        high = std::max(high, pass_bytes);
        cur = 0; off = 0; pass_bytes = 0;
    }
    void* alloc(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        if (bytes == 0) bytes = 256;
        pass_bytes += bytes;
        while (cur < chunks.size()) {
            if (off + bytes <= chunks[cur].cap) {
                void* r = chunks[cur].p + off;
                off += bytes;
                return r;
            }
            ++cur;
            off = 0;
        }
        Chunk c;
        c.cap = std::max(bytes, chunk_default);
        if (hipMalloc((void**)&c.p, c.cap) != hipSuccess) {
            (void)hipGetLastError();
            throw K82Error(K82_ENOMEM, "hipMalloc of " + std::to_string(c.cap) + " bytes failed (workspace)");
        }
        chunks.push_back(c);
        cur = chunks.size() - 1;
        off = bytes;
        return c.p;
    }
    float* f(size_t n) { return (float*)alloc(n * sizeof(float)); }
    // One chunk of at least `bytes` before the first pass needs it (k82_reserve): later passes up to that size never
    // reach hipMalloc.  Nothing of a previous pass may be in use.
    void reserve(size_t bytes) {
        size_t have = 0;
        for (auto& c : chunks) have = std::max(have, c.cap);
        if (chunks.size() == 1 && have >= bytes) return;
        (void)hipDeviceSynchronize();
        // the new chunk first: a reservation that does not fit leaves the arena as it was (it then grows on demand)
        Chunk c;
        c.cap = std::max(bytes, have);
        if (hipMalloc((void**)&c.p, c.cap) != hipSuccess) {
            (void)hipGetLastError();
            throw K82Error(K82_ENOMEM, "hipMalloc of " + std::to_string(c.cap) + " bytes failed (k82_reserve); the workspace keeps its " +
                                           std::to_string(capacity()) + " bytes and grows on demand");
        }
        for (auto& o : chunks) (void)hipFree(o.p);
        chunks.clear();
        cur = 0; off = 0;
        chunks.push_back(c);
        high = std::max(high, c.cap - c.cap / 8);
    }
    size_t capacity() const { size_t t = 0; for (auto& c : chunks) t += c.cap; return t; }
    void release() {
        for (auto& c : chunks) (void)hipFree(c.p);
        chunks.clear();
        cur = 0; off = 0; pass_bytes = 0; high = 0;
    }
};

}  // namespace
static __device__ __align__(256) unsigned char scratch[kPcmStaticBytes];
namespace {
// device address of `scratch` on the CURRENT device (looked up once per device)
void* scratch_address(int device) {
    REQUIRE(device >= 0 && device < kMaxDevices, K82_EINVAL, "device index beyond the static scratch table");
    std::lock_guard<std::mutex> lk(g_scratch_init_mu);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        REQUIRE(p != nullptr, K82_EHIP, "hipGetSymbolAddress returned no address for the static scratch array");
        g_scratch_ptr[device] = p;
    }
    return g_scratch_ptr[device];
}

struct HostT { std::vector<float> v; std::vector<int64_t> shape; };

struct ConvW {
    float* w = nullptr; float* bias = nullptr;
    uint16_t* wb = nullptr;            // bf16 image for the bf16 MFMA mode (2x2 tile shape only)
    uint16_t* wb2 = nullptr;           // split (hi | lo) bf16 image for the bf16x3 mode
    float* wf = nullptr;               // fp32 fragment image (2x2 tile shape): two 16-byte loads per lane, row and tap
    int cin = 0, cin_p = 0, cout = 0, cout_p = 0, taps = 0, pad = 0;
    unsigned* tapmask = nullptr; int up_s = 0, up_shift = 0;
};
struct LstmW { ConvW ih; float* whh = nullptr; };
struct AdaINW { int goff = 0, C = 0; };
struct ResBlk1dW {
    ConvW c1, c2, sc; bool has_sc = false, up = false; AdaINW n1, n2;
    float* poolw = nullptr; float* poolb = nullptr; int cin = 0, cout = 0;
};
struct ResBlock1W {
    ConvW c1[3], c2[3]; AdaINW n1[3], n2[3];
    float *a1[3], *ia1[3], *a2[3], *ia2[3]; int C = 0, k = 0;
};
struct LNW { float* w; float* b; };

enum { FAM_CONV22 = 0, FAM_CONV14, FAM_LSTM, FAM_STATS, FAM_ATTN, FAM_ISTFT, FAM_SRC, FAM_OTHER, FAM_GATHER, NFAM };
const char* const kFamNames[NFAM] = {"conv_mfma_2x2", "conv_mfma_1x4", "lstm", "instnorm", "attention", "istft", "source", "other", "gather"};

struct DebugT { std::vector<float> v; int64_t dims[3]; };

}  // namespace

namespace {
// Everything finalize() builds from the checkpoint: device pointers and shapes only (trivially copyable).  A handle made by
// k82_fork takes a copy of this block -- the same device memory, owned by the handle that uploaded it.
struct K82Weights {
    float *w_word, *w_pos, *w_type0; LNW ln_emb, ln_attn, ln_full;
    ConvW map_in, qkv, attn_dense, ffn, ffn_out, bert_enc;
    LstmW dlstm[3], plstm, shared, telstm;
    int adaln_off[3];
    ConvW dur_proj;
    ResBlk1dW f0blk[3], nblk[3]; ConvW f0proj, nproj;
    float* te_emb; ConvW te_conv[3]; LNW te_ln[3];
    ResBlk1dW enc, dec[4]; float *f0conv_w, *nconv_w; ConvW asr_res;
    float *lin_w, *lin_b; ConvW nconv0, nconv1, ups0, ups1, conv_post;
    ResBlock1W nres[2], rb[6];
    float *sfc_pros_w, *sfc_pros_b, *sfc_dec_w, *sfc_dec_b; int Gp = 0, Gd = 0;
};
}  // namespace

struct k82_handle : K82Weights {
    k82_config cfg{};
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t st = nullptr;          // stream of the current call
    std::mutex mu;
    std::string err;
    bool finalized = false;
    std::unordered_map<std::string, HostT> host;
    std::vector<void*> wallocs;        // device weight allocations
    k82_handle* parent = nullptr;      // k82_fork: the handle whose weights this one reads (it must outlive this one)
    std::atomic<int> forks{0};         // live handles forked from this one

    unsigned char* scratch = nullptr;  // this device's copy of the static scratch array (k82_pcm16_static)

    // ---- workspace + phase-1 state
    Arena aT, aF;
    bool have_phase1 = false;
    int* lstm_err = nullptr;           // device words: [0] set by lstm_split_kernel if a cross-workgroup wait ran out,
    int* conv_err = nullptr;           // [1] (= lstm_err + 1) by an LDS3 conv wave whose wait for its workgroup's step counter ran out
    int conv_spin_limit = 1 << 16;     // polls of that wait (K82_CONV_SPIN_LIMIT: test hook)
    int cus = 0;
    bool ring_off = false;             // K82_RING=0: the bf16 modes keep the generic tap loop (A/B and bit-identity tests only)
    bool bm256_off = false;            // K82_BM256=0: no 256 x 128 workgroup tiles (bit-identity tests only)
    bool dec_zone = false;             // decode() is inside the decoder (AdainResBlk1d stack), before the generator
    int kflush_div = 2;                // K82_KFLUSH_DIV (1, 2, 4): K blocks of 256 / div products.  2: a single sentence's split-K chains are half as long (p50 7.5 -> 7.2 ms); 4 gains nothing more
    bool dec_ksplit = true;            // K82_DEC_KSPLIT=0: no blocked summation / split-K in the decoder (A/B)
    bool splitk_on = true;             // K82_SPLITK=0: no split-K for small launches (A/B and bit-identity tests)
    float* skbuf = nullptr; size_t skcap = 0;   // split-K block sums: one fixed workspace per handle (floats)
    bool t1_b16 = true;                // K82_T1_BF16=0: the bf16 mode keeps the resblock-internal tensor t1 in fp32 in HBM (A/B and bit-identity tests)
    bool fused_stats = true;           // K82_FUSED_STATS=0: every instance-norm statistic from the sweep kernel (A/B only)
    int lstm_split_max_b = 0;          // set in finalize(): CUs / 2 (K82_LSTM_SPLIT_MAX_B=0 selects the streaming kernel everywhere)
    int lstm_spin_limit = 1 << 18;     // polls a split-LSTM workgroup waits for a partner's h (K82_LSTM_SPIN_LIMIT: test hook)
    int lstm_stream_left = 0;          // > 0: passes that still take the streaming kernel after a split-kernel timeout
    int64_t lstm_fallbacks = 0;        // phases re-run on the streaming kernel (k82_prof_read "lstm_fallbacks")
    bool split_used = false;           // the current pass launched lstm_split_kernel
    bool use_split(int nb) const { return nb <= lstm_split_max_b && lstm_stream_left == 0; }
    void begin_pass() { split_used = false; if (lstm_stream_left > 0) --lstm_stream_left; }
    // At a point where the stream has just been synchronised: true if a split-LSTM workgroup of this pass gave up
    // waiting for its partners (co-residency lost to another process / handle on the GPU).  The flag is cleared and
    // the next passes use the streaming kernel, which computes the same bits; the caller re-runs the pass.
    bool lstm_timed_out() {
        int e[2] = {0, 0};
        HIPCHK(hipMemcpy(e, lstm_err, sizeof(e), hipMemcpyDeviceToHost));
        if (e[1] != 0) {                  // not recoverable by a re-run: the kernel's own synchronisation failed
            HIPCHK(hipMemset(conv_err, 0, sizeof(int)));
            have_phase1 = false;
            throw K82Error(K82_EHIP, "conv_mfma_kernel: a wave's bounded wait for its workgroup's LDS step counter ran out (the tile it "
                                     "went on to read may have been unstaged); the results of this call were discarded");
        }
        if (!split_used || e[0] == 0) return false;
        HIPCHK(hipMemset(lstm_err, 0, sizeof(int)));
        lstm_stream_left = 64;
        ++lstm_fallbacks;
        return true;
    }
    int precision = 0;                 // k82_set_precision: 0 fp32 everywhere, 1 bf16 MFMA in decoder + generator
    bool bf16_now = false;             // true only while decode() is inside the decoder / generator
    bool bf16_zone = false;            // true while decode() is inside the decoder / generator, whatever the precision mode
    bool precise = true;               // K82_KSPLIT=0: no blocked summation anywhere (A/B and noise measurements)
    size_t aT_mark_cur = 0, aT_mark_off = 0;
    int B = 0, Tmax = 0, Tp = 0;
    int Ptot = 0, Pp = 0;              // phase 1 runs in the PACKED token layout (kernels.h LenSpec::off): sum of lens, row pitch
    std::vector<int> toff_h; int* toff_d = nullptr;    // column offset of every sentence on the packed axis
    std::vector<int> lens_h, dur_h;
    int64_t* ids_d = nullptr; int* lens_d = nullptr; float* refs_d = nullptr; float* speed_d = nullptr;
    float *gbp = nullptr, *gbd = nullptr, *dcat = nullptr; int* dur_d = nullptr; float* raw_d = nullptr;

    // ---- profiling / debug
    bool prof_on = false;
    std::vector<hipEvent_t> evpool; size_t evnext = 0;
    struct Rec { hipEvent_t a, b; int fam; double work; };
    std::vector<Rec> recs;
    double fam_ms[NFAM] = {0}; int64_t fam_n[NFAM] = {0}; double fam_work[NFAM] = {0};
    bool debug_on = false;
    std::map<std::string, DebugT> dbg;
    std::map<std::string, HostT> forced;   // k82_debug_force: stage tensors replaced after they are computed (B = 1)
    // "F0"/"N": [1][2F]; "har": [22][120F+1] (its polyphase image is rebuilt from it).  Parity tests against committed
    // golden vectors use this to hand the decoder the oracle's curves where the model is ill-conditioned.
    void apply_forced(const char* name, float* dst, int pitch, int rows, int cols) {
        auto it = forced.find(name);
        if (it == forced.end()) return;
        const HostT& t = it->second;
        REQUIRE(B == 1, K82_ESTATE, std::string("k82_debug_force(") + name + ") needs a batch of one sentence");
        REQUIRE(t.shape.size() == 2 && t.shape[0] == rows && t.shape[1] == cols, K82_EINVAL,
                std::string("forced ") + name + " has shape [" + std::to_string(t.shape[0]) + "," + std::to_string(t.shape[1]) +
                    "], this sentence needs [" + std::to_string(rows) + "," + std::to_string(cols) + "]");
        HIPCHK(hipMemcpy2DAsync(dst, (size_t)pitch * sizeof(float), t.v.data(), (size_t)cols * sizeof(float),
                                (size_t)cols * sizeof(float), rows, hipMemcpyHostToDevice, st));
    }

    // ------------------------------------------------------------------ utilities
    void* wmalloc(size_t bytes) {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(bytes, 256)) != hipSuccess) throw K82Error(K82_ENOMEM, "hipMalloc (weights) failed");
        wallocs.push_back(p);
        return p;
    }
    float* upload(const std::vector<float>& v) {
        float* d = (float*)wmalloc(v.size() * sizeof(float));
        HIPCHK(hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
        return d;
    }
    // bf16 (round-to-nearest-even) image of a device-layout weight tensor t[tap][cin_p][cout_p]:
    // [tap][cin_p/16][k-half][cout_p][8] -- the 16 bytes a lane of v_mfma_f32_32x32x16_bf16 needs, contiguous
    static uint16_t bf16_rne(float f) {
        uint32_t u;
        std::memcpy(&u, &f, 4);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);     // NaN stays NaN
        return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    }
    uint16_t* upload_bf16(const std::vector<float>& t, int taps, int cin_p, int cout_p) {
        if (cout_p % 128 != 0) return nullptr;
        const int nt = cin_p / 16;
        std::vector<uint16_t> o((size_t)taps * cin_p * cout_p);
        for (int kk = 0; kk < taps; ++kk)
            for (int it = 0; it < nt; ++it)
                for (int hh = 0; hh < 2; ++hh)
                    for (int co = 0; co < cout_p; ++co)
                        for (int j = 0; j < 8; ++j)
                            o[((((size_t)kk * nt + it) * 2 + hh) * cout_p + co) * 8 + j] =
                                bf16_rne(t[((size_t)kk * cin_p + it * 16 + 8 * hh + j) * cout_p + co]);
        uint16_t* d = (uint16_t*)wmalloc(o.size() * sizeof(uint16_t));
        HIPCHK(hipMemcpy(d, o.data(), o.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        return d;
    }
    // fp32 fragment image: [tap][cin_p/16][k-half][cout_p][8], element j = t[tap][16 it + 2 j + half][co]
    float* upload_frag(const std::vector<float>& t, int taps, int cin_p, int cout_p) {
        if (cout_p % 128 != 0) return nullptr;
        const int nt = cin_p / 16;
        // one spare all-zero 16-channel tile at the end: the 32-channel small-launch kernel reads a whole pair of tiles
        // even when cin_p / 16 is odd (the x side of that half-tile is zero)
        std::vector<float> o((size_t)taps * cin_p * cout_p + (size_t)16 * cout_p, 0.f);
        for (int kk = 0; kk < taps; ++kk)
            for (int it = 0; it < nt; ++it)
                for (int hh = 0; hh < 2; ++hh)
                    for (int co = 0; co < cout_p; ++co)
                        for (int j = 0; j < 8; ++j)
                            o[((((size_t)kk * nt + it) * 2 + hh) * cout_p + co) * 8 + j] =
                                t[((size_t)kk * cin_p + it * 16 + 2 * j + hh) * cout_p + co];
        return upload(o);
    }
    // split image for the bf16x3 mode: [tap][cin_p/16][k-half][cout_p][hi 8 | lo 8], w ~= hi + lo to 2^-17 relative
    uint16_t* upload_split(const std::vector<float>& t, int taps, int cin_p, int cout_p) {
        if (cout_p % 128 != 0) return nullptr;
        const int nt = cin_p / 16;
        std::vector<uint16_t> o((size_t)taps * cin_p * cout_p * 2);
        for (int kk = 0; kk < taps; ++kk)
            for (int it = 0; it < nt; ++it)
                for (int hh = 0; hh < 2; ++hh)
                    for (int co = 0; co < cout_p; ++co)
                        for (int j = 0; j < 8; ++j) {
                            const float w = t[((size_t)kk * cin_p + it * 16 + 8 * hh + j) * cout_p + co];
                            const uint16_t hi = bf16_rne(w);
                            const uint32_t hu = (uint32_t)hi << 16;
                            float hf;
                            std::memcpy(&hf, &hu, 4);
                            const size_t base = ((((size_t)kk * nt + it) * 2 + hh) * cout_p + co) * 16;
                            o[base + j] = hi;
                            o[base + 8 + j] = bf16_rne(w - hf);
                        }
        uint16_t* d = (uint16_t*)wmalloc(o.size() * sizeof(uint16_t));
        HIPCHK(hipMemcpy(d, o.data(), o.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        return d;
    }
    const HostT& get(const std::string& name, std::initializer_list<int64_t> shape) {
        auto it = host.find(name);
        REQUIRE(it != host.end(), K82_ENOENT, "missing weight tensor: " + name);
        std::vector<int64_t> s(shape);
        if (it->second.shape != s) {
            std::string a, b;
            for (auto d : it->second.shape) a += std::to_string(d) + ",";
            for (auto d : s) b += std::to_string(d) + ",";
            throw K82Error(K82_EINVAL, "weight " + name + " has shape [" + a + "] expected [" + b + "]");
        }
        return it->second;
    }
    hipEvent_t next_event() {
        if (evnext == evpool.size()) {
            hipEvent_t e;
            HIPCHK(hipEventCreate(&e));
            evpool.push_back(e);
        }
        return evpool[evnext++];
    }
    void prof_drain() {
        if (recs.empty()) return;
        HIPCHK(hipEventSynchronize(recs.back().b));
        for (auto& r : recs) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, r.a, r.b));
            fam_ms[r.fam] += ms; fam_n[r.fam] += 1; fam_work[r.fam] += r.work;
        }
        recs.clear();
        evnext = 0;
    }
    template <class F> void timed(int fam, double work, F&& f) {
        if (!prof_on) { f(); return; }
        hipEvent_t a = next_event(), b = next_event();
        HIPCHK(hipEventRecord(a, st));
        f();
        HIPCHK(hipEventRecord(b, st));
        recs.push_back({a, b, fam, work});
        if (recs.size() >= 4096) prof_drain();
    }
    void tap(const char* name, const float* p, int C, int pitch, long bs, int nb) {
        if (!debug_on) return;
        HIPCHK(hipStreamSynchronize(st));
        DebugT& d = dbg[name];
        d.dims[0] = nb; d.dims[1] = C; d.dims[2] = pitch;
        d.v.resize((size_t)nb * C * pitch);
        for (int b = 0; b < nb; ++b)
            HIPCHK(hipMemcpy(d.v.data() + (size_t)b * C * pitch, p + (long)b * bs, (size_t)C * pitch * sizeof(float), hipMemcpyDeviceToHost));
    }

    // a PACKED token-rate tensor [C][Pp], handed to the tests in the padded form {B, C, Tp} they know
    void tap_packed(const char* name, const float* p, int C) {
        if (!debug_on) return;
        HIPCHK(hipStreamSynchronize(st));
        DebugT& d = dbg[name];
        d.dims[0] = B; d.dims[1] = C; d.dims[2] = Tp;
        d.v.assign((size_t)B * C * Tp, 0.f);
        for (int b = 0; b < B; ++b)
            HIPCHK(hipMemcpy2D(d.v.data() + (size_t)b * C * Tp, (size_t)Tp * sizeof(float), p + toff_h[b], (size_t)Pp * sizeof(float),
                               (size_t)lens_h[b] * sizeof(float), C, hipMemcpyDeviceToHost));
    }

    // ------------------------------------------------------------------ weight builders
    // W host [cout][cin][k] -> device [k][cin_p][cout_p]
    ConvW mk_conv_raw(const std::vector<float>& W, const std::vector<float>* bias, int cout, int cin, int k) {
        ConvW c;
        c.cin = cin; c.cout = cout; c.taps = k; c.pad = (k - 1) / 2;
        c.cin_p = ru(cin, 16);
        c.cout_p = cout <= 32 ? 32 : cout <= 64 ? 64 : ru(cout, 128);       // 32: conv_post and the F0 / N heads (the 32-row kernel shape)
        std::vector<float> t((size_t)k * c.cin_p * c.cout_p, 0.f);
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci)
                for (int kk = 0; kk < k; ++kk)
                    t[((size_t)kk * c.cin_p + ci) * c.cout_p + co] = W[((size_t)co * cin + ci) * k + kk];
        c.w = upload(t);
        c.wb = upload_bf16(t, k, c.cin_p, c.cout_p);
        c.wb2 = upload_split(t, k, c.cin_p, c.cout_p);
        c.wf = upload_frag(t, k, c.cin_p, c.cout_p);
        std::vector<float> bp(c.cout_p, 0.f);        // always present, padded to cout_p (the kernel loads it unconditionally)
        if (bias) std::copy(bias->begin(), bias->end(), bp.begin());
        c.bias = upload(bp);
        return c;
    }
    ConvW mk_conv(const std::string& prefix, int cout, int cin, int k, bool has_bias = true) {
        const HostT& W = get(prefix + ".weight", {cout, cin, k});
        const std::vector<float>* b = has_bias ? &get(prefix + ".bias", {cout}).v : nullptr;
        return mk_conv_raw(W.v, b, cout, cin, k);
    }
    ConvW mk_linear(const std::string& prefix, int cout, int cin) {
        const HostT& W = get(prefix + ".weight", {cout, cin});
        return mk_conv_raw(W.v, &get(prefix + ".bias", {cout}).v, cout, cin, 1);
    }
    // ConvTranspose1d(cin, cout, k=2s, stride s, padding s/2) as a 3-tap conv over s*cout phase rows
    ConvW mk_convT(const std::string& prefix, int cin, int cout, int k, int s) {
        REQUIRE(k == 2 * s && (s % 2) == 0 && (cout % 64) == 0 && (cout & (cout - 1)) == 0, K82_EINVAL, "unsupported ConvTranspose1d geometry: " + prefix);
        const HostT& W = get(prefix + ".weight", {cin, cout, k});
        const HostT& bv = get(prefix + ".bias", {cout});
        const int p = s / 2;
        ConvW c;
        c.cin = cin; c.cin_p = ru(cin, 16); c.cout = s * cout; c.cout_p = ru(s * cout, 128); c.taps = 3; c.pad = 1;
        c.up_s = s;
        c.up_shift = 0;
        while ((1 << c.up_shift) < cout) ++c.up_shift;
        std::vector<float> t((size_t)3 * c.cin_p * c.cout_p, 0.f), bp(c.cout_p, 0.f);
        for (int r = 0; r < s; ++r)
            for (int d = -1; d <= 1; ++d) {
                const int kidx = r + p - s * d;
                if (kidx < 0 || kidx >= k) continue;
                for (int ci = 0; ci < cin; ++ci)
                    for (int co = 0; co < cout; ++co)
                        t[((size_t)(d + 1) * c.cin_p + ci) * c.cout_p + r * cout + co] = W.v[((size_t)ci * cout + co) * k + kidx];
            }
        for (int r = 0; r < s; ++r)
            for (int co = 0; co < cout; ++co) bp[r * cout + co] = bv.v[co];
        c.w = upload(t);
        c.wb = upload_bf16(t, 3, c.cin_p, c.cout_p);
        c.wb2 = upload_split(t, 3, c.cin_p, c.cout_p);
        c.wf = upload_frag(t, 3, c.cin_p, c.cout_p);
        c.bias = upload(bp);
        std::vector<unsigned> mask(c.cout_p / 64, 0x7u);
        for (size_t mt = 0; mt < mask.size(); ++mt) {
            const int r = (int)(mt * 64) / cout;
            if (r < s) mask[mt] = r < p ? 0x3u : 0x6u;
        }
        c.tapmask = (unsigned*)wmalloc(mask.size() * sizeof(unsigned));
        HIPCHK(hipMemcpy(c.tapmask, mask.data(), mask.size() * sizeof(unsigned), hipMemcpyHostToDevice));
        return c;
    }
    LstmW mk_lstm(const std::string& prefix, int n_in) {
        LstmW l;
        std::vector<float> W((size_t)2048 * n_in), b(2048), hh((size_t)2 * 256 * 256 * 4);
        const char* sfx[2] = {"", "_reverse"};
        for (int d = 0; d < 2; ++d) {
            const HostT& wi = get(prefix + ".weight_ih_l0" + sfx[d], {1024, n_in});
            const HostT& wh = get(prefix + ".weight_hh_l0" + sfx[d], {1024, 256});
            const HostT& bi = get(prefix + ".bias_ih_l0" + sfx[d], {1024});
            const HostT& bh = get(prefix + ".bias_hh_l0" + sfx[d], {1024});
            std::copy(wi.v.begin(), wi.v.end(), W.begin() + (size_t)d * 1024 * n_in);
            for (int i = 0; i < 1024; ++i) b[d * 1024 + i] = bi.v[i] + bh.v[i];
            for (int g = 0; g < 4; ++g)
                for (int j = 0; j < 256; ++j)
                    for (int k = 0; k < 256; ++k)
                        hh[(((size_t)d * 256 + k) * 256 + j) * 4 + g] = wh.v[((size_t)g * 256 + j) * 256 + k];
        }
        l.ih = mk_conv_raw(W, &b, 2048, n_in, 1);
        l.whh = upload(hh);
        return l;
    }
    struct SfcBuilder { std::vector<float> W, b; int rows = 0; };
    AdaINW mk_adain(SfcBuilder& sb, const std::string& prefix, int C) {
        const HostT& w = get(prefix + ".fc.weight", {2 * C, 128});
        const HostT& bb = get(prefix + ".fc.bias", {2 * C});
        AdaINW a;
        a.goff = sb.rows; a.C = C;
        sb.W.insert(sb.W.end(), w.v.begin(), w.v.end());
        sb.b.insert(sb.b.end(), bb.v.begin(), bb.v.end());
        sb.rows += 2 * C;
        return a;
    }
    ResBlk1dW mk_resblk1d(SfcBuilder& sb, const std::string& prefix, int cin, int cout, bool up) {
        ResBlk1dW r;
        r.cin = cin; r.cout = cout; r.up = up; r.has_sc = cin != cout;
        r.c1 = mk_conv(prefix + ".conv1", cout, cin, 3);
        r.c2 = mk_conv(prefix + ".conv2", cout, cout, 3);
        r.n1 = mk_adain(sb, prefix + ".norm1", cin);
        r.n2 = mk_adain(sb, prefix + ".norm2", cout);
        if (r.has_sc) r.sc = mk_conv(prefix + ".conv1x1", cout, cin, 1, false);
        if (up) {
            r.poolw = upload(get(prefix + ".pool.weight", {cin, 1, 3}).v);
            r.poolb = upload(get(prefix + ".pool.bias", {cin}).v);
        }
        return r;
    }
    ResBlock1W mk_resblock1(SfcBuilder& sb, const std::string& prefix, int C, int k) {
        ResBlock1W r;
        r.C = C; r.k = k;
        for (int i = 0; i < 3; ++i) {
            const std::string si = std::to_string(i);
            r.c1[i] = mk_conv(prefix + ".convs1." + si, C, C, k);
            r.c2[i] = mk_conv(prefix + ".convs2." + si, C, C, k);
            r.n1[i] = mk_adain(sb, prefix + ".adain1." + si, C);
            r.n2[i] = mk_adain(sb, prefix + ".adain2." + si, C);
            for (int which = 0; which < 2; ++which) {
                const HostT& a = get(prefix + (which ? ".alpha2." : ".alpha1.") + si, {1, C, 1});
                std::vector<float> ap(ru(C, 16), 1.f), ip(ru(C, 16), 1.f);
                for (int c = 0; c < C; ++c) { ap[c] = a.v[c]; ip[c] = 1.0f / a.v[c]; }
                (which ? r.a2[i] : r.a1[i]) = upload(ap);
                (which ? r.ia2[i] : r.ia1[i]) = upload(ip);
            }
        }
        return r;
    }
    LNW mk_ln(const std::string& wname, const std::string& bname, int C) {
        return LNW{upload(get(wname, {C}).v), upload(get(bname, {C}).v)};
    }

    void finalize() {
        REQUIRE(!finalized, K82_ESTATE, "k82_finalize called twice");
        const int D = cfg.bert_hidden, I = cfg.bert_inter, H = cfg.hidden_dim, E = cfg.bert_embed;
        // ---- ALBERT
        w_word = upload(get("bert.embeddings.word_embeddings.weight", {cfg.n_token, E}).v);
        w_pos = upload(get("bert.embeddings.position_embeddings.weight", {cfg.bert_max_pos, E}).v);
        {
            const HostT& t = get("bert.embeddings.token_type_embeddings.weight", {2, E});
            w_type0 = upload(std::vector<float>(t.v.begin(), t.v.begin() + E));
        }
        ln_emb = mk_ln("bert.embeddings.LayerNorm.weight", "bert.embeddings.LayerNorm.bias", E);
        map_in = mk_linear("bert.encoder.embedding_hidden_mapping_in", D, E);
        const std::string L = "bert.encoder.albert_layer_groups.0.albert_layers.0";
        {
            std::vector<float> W, b;
            for (const char* n : {"query", "key", "value"}) {
                const HostT& w = get(L + ".attention." + n + ".weight", {D, D});
                const HostT& bb = get(L + ".attention." + n + ".bias", {D});
                W.insert(W.end(), w.v.begin(), w.v.end());
                b.insert(b.end(), bb.v.begin(), bb.v.end());
            }
            qkv = mk_conv_raw(W, &b, 3 * D, D, 1);
        }
        attn_dense = mk_linear(L + ".attention.dense", D, D);
        ln_attn = mk_ln(L + ".attention.LayerNorm.weight", L + ".attention.LayerNorm.bias", D);
        ffn = mk_linear(L + ".ffn", I, D);
        ffn_out = mk_linear(L + ".ffn_output", D, I);
        ln_full = mk_ln(L + ".full_layer_layer_norm.weight", L + ".full_layer_layer_norm.bias", D);
        bert_enc = mk_linear("bert_encoder", H, D);
        // ---- predictor
        SfcBuilder sp, sd;
        for (int i = 0; i < 3; ++i) {
            dlstm[i] = mk_lstm("predictor.text_encoder.lstms." + std::to_string(2 * i), H + 128);
            const std::string pre = "predictor.text_encoder.lstms." + std::to_string(2 * i + 1);
            const HostT& w = get(pre + ".fc.weight", {2 * H, 128});
            const HostT& bb = get(pre + ".fc.bias", {2 * H});
            adaln_off[i] = sp.rows;
            sp.W.insert(sp.W.end(), w.v.begin(), w.v.end());
            sp.b.insert(sp.b.end(), bb.v.begin(), bb.v.end());
            sp.rows += 2 * H;
        }
        plstm = mk_lstm("predictor.lstm", H + 128);
        dur_proj = mk_linear("predictor.duration_proj.linear_layer", cfg.max_dur, H);
        shared = mk_lstm("predictor.shared", H + 128);
        const int fc[3][2] = {{H, H}, {H, H / 2}, {H / 2, H / 2}};
        for (int i = 0; i < 3; ++i) {
            f0blk[i] = mk_resblk1d(sp, "predictor.F0." + std::to_string(i), fc[i][0], fc[i][1], i == 1);
            nblk[i] = mk_resblk1d(sp, "predictor.N." + std::to_string(i), fc[i][0], fc[i][1], i == 1);
        }
        f0proj = mk_conv("predictor.F0_proj", 1, H / 2, 1);
        nproj = mk_conv("predictor.N_proj", 1, H / 2, 1);
        // ---- text encoder
        te_emb = upload(get("text_encoder.embedding.weight", {cfg.n_token, H}).v);
        for (int i = 0; i < 3; ++i) {
            const std::string pre = "text_encoder.cnn." + std::to_string(i);
            te_conv[i] = mk_conv(pre + ".0", H, H, cfg.text_kernel);
            te_ln[i] = mk_ln(pre + ".1.gamma", pre + ".1.beta", H);
        }
        telstm = mk_lstm("text_encoder.lstm", H);
        // ---- decoder
        enc = mk_resblk1d(sd, "decoder.encode", H + 2, 1024, false);
        for (int i = 0; i < 3; ++i) dec[i] = mk_resblk1d(sd, "decoder.decode." + std::to_string(i), 1024 + 2 + 64, 1024, false);
        dec[3] = mk_resblk1d(sd, "decoder.decode.3", 1024 + 2 + 64, 512, true);
        for (int which = 0; which < 2; ++which) {
            const std::string pre = which ? "decoder.N_conv" : "decoder.F0_conv";
            const HostT& w = get(pre + ".weight", {1, 1, 3});
            const HostT& b = get(pre + ".bias", {1});
            (which ? nconv_w : f0conv_w) = upload({w.v[0], w.v[1], w.v[2], b.v[0]});
        }
        asr_res = mk_conv("decoder.asr_res.0", 64, 512, 1);
        const std::string G = "decoder.generator";
        lin_w = upload(get(G + ".m_source.l_linear.weight", {1, K82_HARM}).v);
        lin_b = upload(get(G + ".m_source.l_linear.bias", {1}).v);
        {   // noise_convs.0: Conv1d(22,256,k12,s6,p3) -> 2-tap conv over the 132-row polyphase image
            const HostT& w = get(G + ".noise_convs.0.weight", {256, 22, 12});
            std::vector<float> W((size_t)256 * 132 * 2);
            for (int co = 0; co < 256; ++co)
                for (int ci = 0; ci < 22; ++ci)
                    for (int r = 0; r < 6; ++r)
                        for (int j = 0; j < 2; ++j)
                            W[((size_t)co * 132 + ci * 6 + r) * 2 + j] = w.v[((size_t)co * 22 + ci) * 12 + 6 * j + r];
            nconv0 = mk_conv_raw(W, &get(G + ".noise_convs.0.bias", {256}).v, 256, 132, 2);
            nconv0.pad = 0;
        }
        nconv1 = mk_conv(G + ".noise_convs.1", 128, 22, 1);
        ups0 = mk_convT(G + ".ups.0", 512, 256, 20, 10);
        ups1 = mk_convT(G + ".ups.1", 256, 128, 12, 6);
        nres[0] = mk_resblock1(sd, G + ".noise_res.0", 256, 7);
        nres[1] = mk_resblock1(sd, G + ".noise_res.1", 128, 11);
        const int rk[3] = {3, 7, 11};
        for (int i = 0; i < 6; ++i) rb[i] = mk_resblock1(sd, G + ".resblocks." + std::to_string(i), i < 3 ? 256 : 128, rk[i % 3]);
        conv_post = mk_conv(G + ".conv_post", 22, 128, 7);
        sfc_pros_w = upload(sp.W); sfc_pros_b = upload(sp.b); Gp = sp.rows;
        sfc_dec_w = upload(sd.W); sfc_dec_b = upload(sd.b); Gd = sd.rows;
        {
            hipDeviceProp_t prop;
            HIPCHK(hipGetDeviceProperties(&prop, device));
            // beyond CUs / 8 sentences the split grid runs in rounds of ~2.6 us per step each; the streaming kernel's
            // 9-14 us per step is better from about five rounds on
            cus = prop.multiProcessorCount;
            lstm_split_max_b = cus / 2;          // 4 sentences per workgroup: one resident round up to CUs / 2 sentences
            if (const char* e = std::getenv("K82_FUSED_STATS")) fused_stats = std::atoi(e) != 0;
            if (const char* e = std::getenv("K82_RING")) ring_off = std::atoi(e) == 0;
            if (const char* e = std::getenv("K82_T1_BF16")) t1_b16 = std::atoi(e) != 0;
            if (const char* e = std::getenv("K82_SPLITK")) splitk_on = std::atoi(e) != 0;
            if (const char* e = std::getenv("K82_DEC_KSPLIT")) dec_ksplit = std::atoi(e) != 0;
            if (const char* e = std::getenv("K82_KFLUSH_DIV")) { const int v = std::atoi(e); if (v == 1 || v == 2 || v == 4) kflush_div = v; }
            if (const char* e = std::getenv("K82_KSPLIT")) precise = std::atoi(e) != 0;
            if (const char* e = std::getenv("K82_BM256")) bm256_off = std::atoi(e) == 0;
            // K82_LSTM_SPLIT_MAX_B=0: streaming kernel everywhere -- for deployments that KNOW the GPU is shared
            if (const char* e = std::getenv("K82_LSTM_SPLIT_MAX_B")) lstm_split_max_b = std::max(0, std::min(4 * cus, std::atoi(e)));   // (above CUs / 2: rounds of the split grid; tuning)
            if (const char* e = std::getenv("K82_LSTM_SPIN_LIMIT")) lstm_spin_limit = std::max(0, std::atoi(e));   // test hook
        }
        if (const char* e = std::getenv("K82_CONV_SPIN_LIMIT")) conv_spin_limit = std::max(0, std::atoi(e));   // test hook
        lstm_err = (int*)wmalloc(2 * sizeof(int));
        conv_err = lstm_err + 1;
        HIPCHK(hipMemset(lstm_err, 0, 2 * sizeof(int)));
        // split-K launches have at most CUs 64 x 64 tiles and at most 16 K blocks (the k = 5 text-encoder convs)
        skcap = (size_t)cus * 4096 * 16 * kflush_div;
        skbuf = (float*)wmalloc(skcap * sizeof(float));
        host.clear();
        finalized = true;
    }

    // ------------------------------------------------------------------ conv helpers
    ConvP cp(const ConvW& w, const float* x, long xbs, int xpitch, float* y, long ybs, int ypitch, LenSpec lin,
             LenSpec lout, int nb, int Lmax, int dil = 1) {
        ConvP p;
        std::memset(&p, 0, sizeof(p));
        p.x = x; p.xbs = xbs; p.xpitch = xpitch;
        p.w = w.w; p.bias = w.bias; p.wf = w.wf;
        p.wb = bf16_now ? (precision == 2 ? w.wb2 : w.wb) : nullptr;
        p.wb_split = bf16_now && precision == 2 && w.wb2;
        p.y = y; p.ybs = ybs; p.ypitch = ypitch;
        p.lin = lin; p.lout = lout;
        p.cin = w.cin; p.cin_p = w.cin_p; p.cout = w.cout; p.cout_p = w.cout_p; p.taps = w.taps;
        p.dil = dil; p.pad = w.pad * dil;
        p.tr = TR_NONE; p.in_scale = 1.f; p.act = ACT_NONE; p.slope = 1.f;
        p.out_scale = 1.f; p.acc_scale = 1.f;
        p.up_s = w.up_s; p.up_cout_shift = w.up_shift; p.tapmask = w.tapmask;
        p.B = nb; p.Lmax_out = Lmax;
        p.dbg = (ring_off ? 1 : 0) | (bm256_off ? 2 : 0);
        p.cus = cus; p.spin_limit = conv_spin_limit; p.err = conv_err;
        // blocked summation on every fp32 layer outside the decoder / generator (durations, F0 / N, text encoder): fold every
        // ~128 products (conv_mfma_kernel.h KSPLIT).  (16 / kflush_div) / taps 16-channel tiles, kept even for the 32-channel small-launch tiles.
        // ... and, in the fp32 mode, on the decoder's 1024-channel convs (K up to 3312): a single sentence's decoder launches are
        // then split over K blocks like the token-rate ones (run_conv), and a batch sums the same blocks in one workgroup
        p.kflush = (precise && (!bf16_zone || (dec_zone && precision == 0 && dec_ksplit))) ? std::max(2, ((16 / kflush_div) / std::max(1, w.taps)) & ~1) : 0;
        return p;
    }
    // operand shapes the kernel and its grid assume (conv_mfma.hip launch_conv), checked before every launch
    static void check_conv(const ConvP& p) {
        REQUIRE((p.taps - 1) * p.dil <= 64, K82_EINVAL, "conv halo exceeds the LDS tile");
        REQUIRE(p.x && p.w && p.bias && (p.y || p.accbuf), K82_EINVAL, "conv operand missing");
        REQUIRE(p.cout_p % 128 != 0 || p.wf, K82_EINVAL, "conv fragment image missing");
        REQUIRE(p.cin_p % 16 == 0 && p.cout_p % 32 == 0 && p.cin <= p.cin_p && p.cout <= p.cout_p, K82_EINVAL, "conv channel padding");
        REQUIRE(p.xpitch % 4 == 0 && (!p.y || p.ypitch % 4 == 0) && (!p.res || p.res_shift || p.up_s || p.rpitch % 4 == 0) &&
                (!p.accbuf || p.apitch % 4 == 0), K82_EINVAL, "conv row pitch must be a multiple of 4 floats");
        REQUIRE(!p.accbuf || (p.res && !p.up_s && !p.res_shift), K82_EINVAL, "accumulating epilogue needs a residual and a plain layout");
        REQUIRE(p.act != ACT_SNAKE || (p.alpha && p.inv_alpha), K82_EINVAL, "snake needs alpha");
        REQUIRE(p.act != ACT_SNAKE || p.cout_p >= 64, K82_EINVAL, "no Snake conv with <= 32 output channels is built");
        REQUIRE(p.act != ACT_LEAKY || (p.slope >= 0.f && p.slope <= 1.f), K82_EINVAL, "LeakyReLU slope must be in [0, 1]");
        REQUIRE(p.tr != TR_NORM || p.norm, K82_EINVAL, "norm transform needs statistics");
        // bf16 activations in HBM exist in the bf16 ring kernels of the Snake convs only (conv_mfma.hip launch_ring's own conditions)
        REQUIRE(!(p.x_b16 || p.y_b16) || (p.wb && !p.wb_split && !(p.dbg & 1) && !p.tapmask && p.act == ACT_SNAKE && p.tr == TR_NORM && p.cin % 16 == 0 &&
                                          p.cin_p >= 32 && p.cout % 32 == 0 && (p.taps == 3 || p.taps == 7 || p.taps == 11) && !p.up_s && !p.res_shift &&
                                          (p.y_b16 ? (p.y && !p.res && !p.accbuf && !p.x_b16) : (p.res != nullptr))),
                K82_EINVAL, "bf16-activation conv outside the shapes the ring kernels are built for");
        REQUIRE(!p.stat || (p.y && !p.up_s && !p.res_shift && p.snt * 32 >= p.Lmax_out), K82_EINVAL, "statistics partials need a stored, plain-layout output");
        // the lean epilogue addresses a sentence's rows with 32-bit byte offsets
        const long kMaxElems = 1L << 30;
        REQUIRE((long)p.cout_p * p.ypitch < kMaxElems && (long)p.cout_p * p.rpitch < kMaxElems && (long)p.cout_p * p.apitch < kMaxElems &&
                (!p.stat || (long)p.cout_p * p.snt < (kMaxElems >> 1)), K82_EINVAL, "conv output of one sentence exceeds 4 GB");
    }
    void run_conv(const ConvP& p0, const int* host_base) {
        ConvP p = p0;
        // small launches of the token-rate / prosody layers (a single sentence, a short batch): split over K blocks -- same bits
        if (splitk_on && skbuf) {
            const int nblk = conv_splitk_blocks(p);
            const int pp = ru(p.Lmax_out, 4);
            if (nblk >= 2 && (size_t)nblk * p.B * p.cout_p * pp <= skcap) { p.part = skbuf; p.nblk = nblk; p.ppitch = pp; }
        }
        check_conv(p);
        const int fam = (p.cout_p % 128 == 0) ? FAM_CONV22 : FAM_CONV14;
        double work = 0;
        if (prof_on) work = conv_flops(p, p.lout.lens ? host_base : nullptr);
        timed(fam, work, [&] { launch_conv(p, st); });
    }
    float4* stats(const float* x, long xbs, int xpitch, LenSpec ls, int C, const float* gb, int gstride, int goff,
                  Arena& ar, int Lmax, double elems) {
        float4* nm = (float4*)ar.alloc((size_t)B * ru(C, 16) * sizeof(float4));
        // entries of the padded channels: (mean, rstd, gain, beta) = 0, i.e. "stage an exact zero" (ConvP::cin_pad_ok)
        if (ru(C, 16) != C) HIPCHK(hipMemsetAsync(nm, 0, (size_t)B * ru(C, 16) * sizeof(float4), st));
        timed(FAM_STATS, elems * 4.0, [&] { launch_adain_stats(x, xbs, xpitch, ls, C, B, gb, gstride, goff, nm, ru(C, 16), Lmax, st); });
        return nm;
    }

    // istftnet.AdainResBlk1d.  x [B][cin][xpitch] at length lx; out gets [cout] rows at length (up ? 2 : 1)*lx.
    void resblk1d(const ResBlk1dW& L, const float* x, long xbs, int xpitch, LenSpec lx, int Lmax, const int* hbase,
                  double sumL, float* out, long obs, int opitch, const float* gb, int gstride, Arena& ar,
                  bool fuse = true) {
        const int nb = B;
        float4* n1 = stats(x, xbs, xpitch, lx, L.cin, gb, gstride, L.n1.goff, ar, Lmax, sumL * L.cin);
        LenSpec lo = lx;
        int Lo = Lmax;
        const float* c1in = x; long c1bs = xbs; int c1pitch = xpitch;
        if (L.up) {
            lo.lmul *= 2; lo.ladd *= 2; Lo = 2 * Lmax;
            const int pp = ru(Lo, 4);
            float* pooled = ar.f((size_t)nb * L.c1.cin_p * pp);
            timed(FAM_OTHER, 0, [&] { launch_pool_up2(x, xbs, xpitch, n1, ru(L.cin, 16), 0.2f, L.poolw, L.poolb, pooled, (long)L.c1.cin_p * pp, pp, lx, L.cin, nb, Lmax, st); });
            c1in = pooled; c1bs = (long)L.c1.cin_p * pp; c1pitch = pp;
        }
        const int tp = ru(Lo, 4);
        float* t1 = ar.f((size_t)nb * L.cout * tp);
        ConvP p1 = cp(L.c1, c1in, c1bs, c1pitch, t1, (long)L.cout * tp, tp, lo, lo, nb, Lo);
        if (!L.up) { p1.tr = TR_NORM; p1.norm = n1; p1.nstride = ru(L.cin, 16); p1.act = ACT_LEAKY; p1.slope = 0.2f; p1.cin_pad_ok = 1; }
        float4* n2;
        // (not in the F0 / N stacks: F0 sets the phase of the harmonic source, where a rounding of rstd -- harmless for any
        // stage tolerance -- moves the waveform; they keep the all-fp64 sweep)
        if (fused_stats && fuse) {                  // conv1's epilogue leaves the partials, no second pass over t1
            const int snt = (Lo + 31) / 32;
            float2* part = (float2*)ar.alloc((size_t)nb * L.cout * snt * sizeof(float2));
            p1.stat = part; p1.sbs = (long)L.cout * snt; p1.snt = snt;
            run_conv(p1, hbase);
            n2 = (float4*)ar.alloc((size_t)nb * ru(L.cout, 16) * sizeof(float4));
            timed(FAM_STATS, sumL * (L.up ? 2 : 1) * L.cout / 4.0, [&] { launch_adain_finalize(part, (long)L.cout * snt, snt, lo, L.cout, nb, gb, gstride, L.n2.goff, n2, ru(L.cout, 16), st); });
        } else {
            run_conv(p1, hbase);
            n2 = stats(t1, (long)L.cout * tp, tp, lo, L.cout, gb, gstride, L.n2.goff, ar, Lo, sumL * (L.up ? 2 : 1) * L.cout);
        }
        const float* res = x; long rbs = xbs; int rpitch = xpitch;
        if (L.has_sc) {
            const int sp_ = ru(Lmax, 4);
            float* sc = ar.f((size_t)nb * L.cout * sp_);
            ConvP ps = cp(L.sc, x, xbs, xpitch, sc, (long)L.cout * sp_, sp_, lx, lx, nb, Lmax);
            run_conv(ps, hbase);
            res = sc; rbs = (long)L.cout * sp_; rpitch = sp_;
        }
        ConvP p2 = cp(L.c2, t1, (long)L.cout * tp, tp, out, obs, opitch, lo, lo, nb, Lo);
        p2.tr = TR_NORM; p2.norm = n2; p2.nstride = ru(L.cout, 16); p2.act = ACT_LEAKY; p2.slope = 0.2f;
        p2.res = res; p2.rbs = rbs; p2.rpitch = rpitch; p2.res_shift = L.up ? 1 : 0;
        p2.out_scale = 0.70710678118654752f;
        run_conv(p2, hbase);
    }

    // istftnet.AdaINResBlock1.  x: block input (never written unless xr == x); xr: running tensor;
    // accbuf != nullptr: the last conv accumulates its result * acc_scale into accbuf instead of storing.
    void resblock1(const ResBlock1W& L, const float* x, float* xr, float* t1, long bs, int pitch, LenSpec ls, int Lmax,
                   const int* hbase, double sumL, float* accbuf, int acc_init, float acc_scale, Arena& ar,
                   float4** share = nullptr) {
        const int dil[3] = {1, 3, 5};
        // the statistics of every tensor a conv of this block writes come out of that conv's epilogue (one partial per
        // row and 32 columns); only the block input is swept by the statistics kernel
        const int snt = (Lmax + 31) / 32;
        const bool fs = fused_stats;
        float2* part = fs ? (float2*)ar.alloc((size_t)B * L.C * snt * sizeof(float2)) : nullptr;
        auto fused = [&](int goff) {
            if (!fs) return (float4*)nullptr;
            float4* nm = (float4*)ar.alloc((size_t)B * ru(L.C, 16) * sizeof(float4));
            timed(FAM_STATS, sumL * L.C / 4.0, [&] { launch_adain_finalize(part, (long)L.C * snt, snt, ls, L.C, B, gbd, Gd, goff, nm, ru(L.C, 16), st); });
            return nm;
        };
        float4* n1_next = nullptr;
        for (int i = 0; i < 3; ++i) {
            const float* cur = i == 0 ? x : xr;
            float4* n1 = n1_next;
            if (i == 0 && share && *share) {            // a sibling block already swept this input
                n1 = (float4*)ar.alloc((size_t)B * ru(L.C, 16) * sizeof(float4));
                const float4* src = *share;
                timed(FAM_STATS, 0, [&] { launch_adain_regamma(src, n1, ru(L.C, 16), L.C, B, gbd, Gd, L.n1[0].goff, st); });
            } else if (i == 0 || !fs) {
                n1 = stats(cur, bs, pitch, ls, L.C, gbd, Gd, L.n1[i].goff, ar, Lmax, sumL * L.C);
                if (i == 0 && share) *share = n1;
            }
            ConvP p1 = cp(L.c1[i], cur, bs, pitch, t1, bs, pitch, ls, ls, B, Lmax, dil[i]);
            if (fs) { p1.stat = part; p1.sbs = (long)L.C * snt; p1.snt = snt; }
            // bf16 mode: t1 -- written by this conv, read by the next one, by nothing else -- crosses HBM as bf16 channel pairs
            const bool tb = t1_b16 && fs && bf16_now && precision == 1 && !ring_off && L.C % 32 == 0 && (L.k == 3 || L.k == 7 || L.k == 11);
            p1.y_b16 = tb ? 1 : 0;
            p1.tr = TR_NORM; p1.norm = n1; p1.nstride = ru(L.C, 16);
            p1.act = ACT_SNAKE; p1.alpha = L.a1[i]; p1.inv_alpha = L.ia1[i];
            run_conv(p1, hbase);
            float4* n2 = fs ? fused(L.n2[i].goff) : stats(t1, bs, pitch, ls, L.C, gbd, Gd, L.n2[i].goff, ar, Lmax, sumL * L.C);
            const bool last = i == 2 && accbuf;
            ConvP p2 = cp(L.c2[i], t1, bs, pitch, last ? nullptr : xr, bs, pitch, ls, ls, B, Lmax, 1);
            p2.tr = TR_NORM; p2.norm = n2; p2.nstride = ru(L.C, 16);
            p2.act = ACT_SNAKE; p2.alpha = L.a2[i]; p2.inv_alpha = L.ia2[i];
            p2.res = cur; p2.rbs = bs; p2.rpitch = pitch;
            p2.x_b16 = tb ? 1 : 0;
            if (last) { p2.accbuf = accbuf; p2.abs_ = bs; p2.apitch = pitch; p2.acc_init = acc_init; p2.acc_scale = acc_scale; }
            if (i < 2 && fs) { p2.stat = part; p2.sbs = (long)L.C * snt; p2.snt = snt; }
            run_conv(p2, hbase);
            if (i < 2) n1_next = fused(L.n1[i + 1].goff);
        }
    }

    void lstm(const LstmW& L, const float* x, long xbs, int xpitch, int cin_rows, LenSpec ls, int Lmax, const int* hbase,
              float* y, long ybs, int ypitch, Arena& ar, double steps) {
        (void)cin_rows;
        const int gp = ru(Lmax, 4);
        float* gx = ar.f((size_t)B * 2048 * gp);
        ConvP p = cp(L.ih, x, xbs, xpitch, gx, (long)2048 * gp, gp, ls, ls, B, Lmax);
        run_conv(p, hbase);
        if (use_split(B)) {               // few sentences: the weight-resident four-workgroup recurrence (same bits)
            float* hx = ar.f((size_t)ru(B, 4) * 2048);
            split_used = true;
            timed(FAM_LSTM, steps, [&] { launch_lstm_split(gx, (long)2048 * gp, gp, L.whh, y, ybs, ypitch, ls, B, hx, lstm_err, cus, lstm_spin_limit, st); });
        } else {
            timed(FAM_LSTM, steps, [&] { launch_lstm(gx, (long)2048 * gp, gp, L.whh, y, ybs, ypitch, ls, B, st); });
        }
    }

    // the same on the packed token axis: the input projection sees ONE long row (dense tiles), the recurrences their own columns
    void lstm_packed(const LstmW& L, const float* x, LenSpec lt, float* y, Arena& ar, double steps) {
        float* gx = ar.f((size_t)2048 * Pp);
        const LenSpec lp{nullptr, 0, Ptot, nullptr};
        run_conv(cp(L.ih, x, 0, Pp, gx, 0, Pp, lp, lp, 1, Ptot), nullptr);
        if (use_split(B)) {
            float* hx = ar.f((size_t)ru(B, 4) * 2048);
            split_used = true;
            timed(FAM_LSTM, steps, [&] { launch_lstm_split(gx, 0, Pp, L.whh, y, 0, Pp, lt, B, hx, lstm_err, cus, lstm_spin_limit, st); });
        } else {
            timed(FAM_LSTM, steps, [&] { launch_lstm(gx, 0, Pp, L.whh, y, 0, Pp, lt, B, st); });
        }
    }

    // ------------------------------------------------------------------ phase 1
    template <class T> T* to_device(const T* src, size_t n, Arena& ar) {
        T* d = (T*)ar.alloc(n * sizeof(T));
        hipPointerAttribute_t at;
        hipError_t e = hipPointerGetAttributes(&at, src);
        const bool on_dev = (e == hipSuccess && at.type == hipMemoryTypeDevice);
        if (e != hipSuccess) (void)hipGetLastError();
        HIPCHK(hipMemcpyAsync(d, src, n * sizeof(T), on_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
        return d;
    }

    static bool is_device_ptr(const void* p) {
        hipPointerAttribute_t at;
        const hipError_t e = hipPointerGetAttributes(&at, p);
        if (e != hipSuccess) { (void)hipGetLastError(); return false; }
        return at.type == hipMemoryTypeDevice;
    }
    void durations(const int64_t* ids, const int32_t* lens, int nb, int tmax, const float* ref_s, const float* speed,
                   const int32_t* forced, int32_t* pred_dur, int64_t* n_frames) {
        durations_once(ids, lens, nb, tmax, ref_s, speed, forced, pred_dur, n_frames);
        if (lstm_timed_out()) {           // same inputs, streaming BiLSTM kernel: same bits as an undisturbed split run
            durations_once(ids, lens, nb, tmax, ref_s, speed, forced, pred_dur, n_frames);
            REQUIRE(!lstm_timed_out(), K82_EHIP, "LSTM recurrence failed on the streaming kernel as well");
        }
    }
    void durations_once(const int64_t* ids, const int32_t* lens, int nb, int tmax, const float* ref_s, const float* speed,
                        const int32_t* forced, int32_t* pred_dur, int64_t* n_frames) {
        REQUIRE(finalized, K82_ESTATE, "k82_finalize has not been called");
        REQUIRE(nb >= 1 && tmax >= 2 && tmax <= K82_MAX_TOKENS, K82_EINVAL, "B must be >= 1 and 2 <= Tmax <= 512");
        REQUIRE(ids && lens && ref_s, K82_EINVAL, "ids, lens and ref_s must not be NULL");
        for (int b = 0; b < nb; ++b) REQUIRE(lens[b] >= 2 && lens[b] <= tmax, K82_EINVAL, "lens[b] must be in [2, Tmax]");
        if (!is_device_ptr(ids))          // host ids index the embedding tables: reject what would read outside them
            for (int b = 0; b < nb; ++b)
                for (int t = 0; t < lens[b]; ++t) {
                    const int64_t v = ids[(size_t)b * tmax + t];
                    REQUIRE(v >= 0 && v < cfg.n_token, K82_EINVAL, "token id " + std::to_string(v) + " outside [0, n_token) at sentence " + std::to_string(b) + ", position " + std::to_string(t));
                }
        begin_pass();
        std::vector<float> sp(nb, 1.0f);
        if (speed) for (int b = 0; b < nb; ++b) { REQUIRE(speed[b] > 0.f && std::isfinite(speed[b]), K82_EINVAL, "speed must be positive"); sp[b] = speed[b]; }
        have_phase1 = false;
        aT.reset();
        B = nb; Tmax = tmax; Tp = ru(tmax, 32);
        lens_h.assign(lens, lens + nb);
        // packed token axis (kernels.h LenSpec::off): sentence b owns the columns [toff[b], toff[b] + lens[b])
        toff_h.resize(nb);
        long ptot = 0;
        for (int b = 0; b < nb; ++b) { toff_h[b] = (int)ptot; ptot += lens[b]; }
        REQUIRE(ptot <= 400000, K82_EINVAL, "more than 400000 tokens in one batch: split it (32-bit row offsets in the conv epilogue)");
        Ptot = (int)ptot; Pp = ru(Ptot, 32);
        const int P = Pp;
        ids_d = to_device(ids, (size_t)nb * tmax, aT);
        lens_d = to_device(lens_h.data(), (size_t)nb, aT);
        toff_d = to_device(toff_h.data(), (size_t)nb, aT);
        refs_d = to_device(ref_s, (size_t)nb * K82_STYLE_DIM, aT);
        speed_d = to_device(sp.data(), (size_t)nb, aT);
        HIPCHK(hipStreamSynchronize(st));   // sp is a stack temporary
        const LenSpec lt{lens_d, 1, 0, toff_d};              // the per-sentence kernels: own columns of the packed axis
        const LenSpec lp{nullptr, 0, Ptot, nullptr};         // the Linears (1-tap convs): the batch is one long row
        const double sumT = (double)Ptot;
        const int D = cfg.bert_hidden, I = cfg.bert_inter;

        gbp = aT.f((size_t)nb * Gp);
        gbd = aT.f((size_t)nb * Gd);
        timed(FAM_OTHER, 0, [&] { launch_style_fc(sfc_pros_w, sfc_pros_b, Gp, refs_d, 128, gbp, Gp, nb, st); });
        timed(FAM_OTHER, 0, [&] { launch_style_fc(sfc_dec_w, sfc_dec_b, Gd, refs_d, 0, gbd, Gd, nb, st); });

        // ---- ALBERT
        float* emb = aT.f((size_t)128 * P);
        timed(FAM_OTHER, 0, [&] { launch_albert_embed(ids_d, tmax, w_word, w_pos, w_type0, ln_emb.w, ln_emb.b, 1e-12f, emb, 0, P, lt, nb, cfg.n_token, st); });
        float* h = aT.f((size_t)D * P);
        float* h1 = aT.f((size_t)D * P);
        float* tmp = aT.f((size_t)D * P);
        float* ctx = aT.f((size_t)D * P);
        float* qkvb = aT.f((size_t)3 * D * P);
        float* ff = aT.f((size_t)I * P);
        auto lin = [&](const ConvW& w, const float* x, float* y) { return cp(w, x, 0, P, y, 0, P, lp, lp, 1, Ptot); };
        run_conv(lin(map_in, emb, h), nullptr);
        for (int layer = 0; layer < cfg.bert_layers; ++layer) {
            run_conv(lin(qkv, h, qkvb), nullptr);
            double aw = 0;
            for (int b = 0; b < nb; ++b) aw += 4.0 * D * (double)lens[b] * lens[b];
            timed(FAM_ATTN, aw, [&] { launch_attention(qkvb, 0, P, ctx, 0, P, lt, cfg.bert_heads, nb, tmax, st); });
            ConvP pd = lin(attn_dense, ctx, tmp);
            pd.res = h; pd.rbs = 0; pd.rpitch = P;
            run_conv(pd, nullptr);
            timed(FAM_OTHER, 0, [&] { launch_layernorm_cols(tmp, nullptr, h1, 0, 0, P, lt, D, nb, tmax, 1e-12f, ln_attn.w, ln_attn.b, nullptr, 0, 0, 1.0f, st); });
            ConvP pf = lin(ffn, h1, ff);
            pf.out_act = OACT_GELU_NEW;
            run_conv(pf, nullptr);
            ConvP po = lin(ffn_out, ff, tmp);
            po.res = h1; po.rbs = 0; po.rpitch = P;
            run_conv(po, nullptr);
            timed(FAM_OTHER, 0, [&] { launch_layernorm_cols(tmp, nullptr, h, 0, 0, P, lt, D, nb, tmax, 1e-12f, ln_full.w, ln_full.b, nullptr, 0, 0, 1.0f, st); });
        }
        tap_packed("bert_out", h, D);

        // ---- bert_encoder + DurationEncoder (rows a6, a7)
        const int H = cfg.hidden_dim;
        dcat = aT.f((size_t)640 * P);
        run_conv(lin(bert_enc, h, dcat), nullptr);
        tap_packed("d_en", dcat, H);
        timed(FAM_OTHER, 0, [&] { launch_fill_rows(dcat, 0, P, H, 128, refs_d, K82_STYLE_DIM, 128, lt, nb, tmax, st); });
        float* hl = aT.f((size_t)H * P);
        for (int l = 0; l < 3; ++l) {
            lstm_packed(dlstm[l], dcat, lt, hl, aT, sumT);
            // AdaLayerNorm writes rows 0..511 of dcat; the style rows 512..639 stay
            timed(FAM_OTHER, 0, [&] { launch_layernorm_cols(hl, nullptr, dcat, 0, 0, P, lt, H, nb, tmax, 1e-5f, nullptr, nullptr, gbp, Gp, adaln_off[l], 1.0f, st); });
        }
        tap_packed("d", dcat, 640);
        // ---- duration head (row a8)
        lstm_packed(plstm, dcat, lt, hl, aT, sumT);
        tap_packed("dur_lstm", hl, H);
        float* logits = aT.f((size_t)cfg.max_dur * P);
        run_conv(lin(dur_proj, hl, logits), nullptr);
        dur_d = (int*)aT.alloc((size_t)nb * tmax * sizeof(int));
        raw_d = aT.f((size_t)nb * tmax);
        timed(FAM_OTHER, 0, [&] { launch_durations(logits, 0, P, cfg.max_dur, speed_d, lens_d, nb, tmax, dur_d, raw_d, st, toff_d); });
        tap("duration", raw_d, 1, tmax, tmax, nb);
        dur_h.resize((size_t)nb * tmax);
        if (forced) {
            for (int b = 0; b < nb; ++b)
                for (int t = 0; t < tmax; ++t) {
                    const int v = t < lens[b] ? forced[(size_t)b * tmax + t] : 0;
                    REQUIRE(v >= 0 && v <= 4096, K82_EINVAL, "forced_dur out of range");
                    dur_h[(size_t)b * tmax + t] = v;
                }
            HIPCHK(hipMemcpyAsync(dur_d, dur_h.data(), dur_h.size() * sizeof(int), hipMemcpyHostToDevice, st));
            HIPCHK(hipStreamSynchronize(st));
        } else {
            HIPCHK(hipMemcpyAsync(dur_h.data(), dur_d, dur_h.size() * sizeof(int), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
        }
        if (pred_dur) std::copy(dur_h.begin(), dur_h.end(), pred_dur);
        if (n_frames)
            for (int b = 0; b < nb; ++b) {
                int64_t a = 0;
                for (int t = 0; t < lens[b]; ++t) a += dur_h[(size_t)b * tmax + t];
                n_frames[b] = a;
            }
        aT_mark_cur = aT.cur; aT_mark_off = aT.off;
        have_phase1 = true;
    }

    // ------------------------------------------------------------------ phase 2
    void decode(const k82_noise* noise, float* audio, int64_t capacity, int64_t* audio_off) {
        decode_once(noise, audio, capacity, audio_off);
        if (lstm_timed_out()) {           // the token-rate state of phase 1 is intact: run phase 2 again, streaming kernel
            decode_once(noise, audio, capacity, audio_off);
            REQUIRE(!lstm_timed_out(), K82_EHIP, "LSTM recurrence failed on the streaming kernel as well");
        }
    }
    void decode_once(const k82_noise* noise, float* audio, int64_t capacity, int64_t* audio_off) {
        REQUIRE(have_phase1, K82_ESTATE, "k82_decode needs a preceding successful k82_durations on this handle");
        REQUIRE(audio_off, K82_EINVAL, "audio_off must not be NULL");
        const int nb = B, tmax = Tmax, Tp_ = Tp, H = cfg.hidden_dim;
        std::vector<int> nfr_h(nb);
        std::vector<int64_t> aoff(nb + 1, 0);
        int Fmax = 0;
        double sumF = 0;
        for (int b = 0; b < nb; ++b) {
            int a = 0;
            for (int t = 0; t < lens_h[b]; ++t) a += dur_h[(size_t)b * tmax + t];
            REQUIRE(a >= 1, K82_EINVAL, "a sentence has zero frames");
            nfr_h[b] = a; Fmax = std::max(Fmax, a); sumF += a;
            aoff[b + 1] = aoff[b] + (int64_t)K82_FRAME_SAMPLES * a;
        }
        REQUIRE(600.0 * Fmax < 16000000.0, K82_EINVAL, "sentence longer than 2^24 samples");
        std::copy(aoff.begin(), aoff.end(), audio_off);
        if (capacity < aoff[nb] || !audio) throw K82Error(K82_ENOSPC, "audio buffer too small: need " + std::to_string(aoff[nb]) + " floats");
        begin_pass();
        aF.reset();
        aT.cur = aT_mark_cur; aT.off = aT_mark_off;
        const int* hb = nfr_h.data();
        const int PF = ru(Fmax, 4), P2 = ru(2 * Fmax, 4), P20 = ru(20 * Fmax + 1, 4), P120 = ru(120 * Fmax + 1, 4);
        const long P600 = 600L * Fmax;
        int* nfr_d = (int*)aF.alloc(nb * sizeof(int));
        int* idx = (int*)aF.alloc((size_t)nb * PF * sizeof(int));
        int64_t* aoff_d = (int64_t*)aF.alloc((nb + 1) * sizeof(int64_t));
        HIPCHK(hipMemcpyAsync(aoff_d, aoff.data(), (nb + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
        timed(FAM_OTHER, 0, [&] { launch_alignment(dur_d, tmax, lens_d, nb, idx, PF, nfr_d, st); });
        const LenSpec lt{lens_d, 1, 0}, lF{nfr_d, 1, 0}, l2F{nfr_d, 2, 0}, l20{nfr_d, 20, 0}, l20p{nfr_d, 20, 1}, l120{nfr_d, 120, 1};

        // ---- length regulation + F0/N predictor (rows a9, a10)
        float* en = aF.f((size_t)nb * 640 * PF);
        double sumTok = 0;
        for (int b = 0; b < nb; ++b) sumTok += lens_h[b];
        // length regulation: algorithmic bytes = one read of the token rows + one write of the frame rows
        timed(FAM_GATHER, 4.0 * 640 * (sumTok + sumF), [&] { launch_gather(dcat, 0, Pp, 640, idx, PF, nfr_d, en, (long)640 * PF, PF, nb, Fmax, st, toff_d); });   // (dcat: packed token axis)
        tap("en", en, 640, PF, (long)640 * PF, nb);
        float* sh = aF.f((size_t)nb * H * PF);
        lstm(shared, en, (long)640 * PF, PF, 640, lF, Fmax, hb, sh, (long)H * PF, PF, aF, sumF);
        tap("shared", sh, H, PF, (long)H * PF, nb);
        float* f0 = aF.f((size_t)nb * P2);
        float* nn = aF.f((size_t)nb * P2);
        for (int which = 0; which < 2; ++which) {
            const ResBlk1dW* blk = which ? nblk : f0blk;
            float* a = aF.f((size_t)nb * H * PF);
            resblk1d(blk[0], sh, (long)H * PF, PF, lF, Fmax, hb, sumF, a, (long)H * PF, PF, gbp, Gp, aF, false);
            tap(which ? "n_blk0" : "f0_blk0", a, H, PF, (long)H * PF, nb);
            float* bq = aF.f((size_t)nb * (H / 2) * P2);
            resblk1d(blk[1], a, (long)H * PF, PF, lF, Fmax, hb, sumF, bq, (long)(H / 2) * P2, P2, gbp, Gp, aF, false);
            tap(which ? "n_blk1" : "f0_blk1", bq, H / 2, P2, (long)(H / 2) * P2, nb);
            float* c = aF.f((size_t)nb * (H / 2) * P2);
            resblk1d(blk[2], bq, (long)(H / 2) * P2, P2, l2F, 2 * Fmax, hb, 2 * sumF, c, (long)(H / 2) * P2, P2, gbp, Gp, aF, false);
            tap(which ? "n_blk2" : "f0_blk2", c, H / 2, P2, (long)(H / 2) * P2, nb);
            run_conv(cp(which ? nproj : f0proj, c, (long)(H / 2) * P2, P2, which ? nn : f0, P2, P2, l2F, l2F, nb, 2 * Fmax), hb);
        }
        apply_forced("F0", f0, P2, 1, 2 * nfr_h[0]);
        apply_forced("N", nn, P2, 1, 2 * nfr_h[0]);
        tap("F0", f0, 1, P2, P2, nb);
        tap("N", nn, 1, P2, P2, nb);

        // ---- text encoder (row a11), token rate
        float* te = aT.f((size_t)nb * H * Tp_);
        float* te2 = aT.f((size_t)nb * H * Tp_);
        const long tbs = (long)H * Tp_;
        timed(FAM_OTHER, 0, [&] { launch_embed(ids_d, tmax, te_emb, H, te, tbs, Tp_, lt, nb, cfg.n_token, st); });
        for (int i = 0; i < 3; ++i) {
            run_conv(cp(te_conv[i], te, tbs, Tp_, te2, tbs, Tp_, lt, lt, nb, tmax), lens_h.data());
            timed(FAM_OTHER, 0, [&] { launch_layernorm_cols(te2, nullptr, te, tbs, tbs, Tp_, lt, H, nb, tmax, 1e-5f, te_ln[i].w, te_ln[i].b, nullptr, 0, 0, 0.2f, st); });
        }
        double sumT = 0;
        for (int b = 0; b < nb; ++b) sumT += lens_h[b];
        lstm(telstm, te, tbs, Tp_, H, lt, tmax, lens_h.data(), te2, tbs, Tp_, aT, sumT);
        tap("t_en", te2, H, Tp_, tbs, nb);

        // ---- decoder front (row a12)
        const int C0 = ru(H + 2, 16), C1 = ru(1024 + 64 + 2, 16);
        float* cat0 = aF.f((size_t)nb * C0 * PF);
        const long c0bs = (long)C0 * PF;
        timed(FAM_GATHER, 4.0 * H * (sumT + sumF), [&] { launch_gather(te2, tbs, Tp_, H, idx, PF, nfr_d, cat0, c0bs, PF, nb, Fmax, st); });
        tap("asr", cat0, H, PF, c0bs, nb);
        timed(FAM_OTHER, 0, [&] { launch_conv_k3s2(f0, P2, f0conv_w, cat0 + (long)H * PF, c0bs, nfr_d, nb, Fmax, st); });
        timed(FAM_OTHER, 0, [&] { launch_conv_k3s2(nn, P2, nconv_w, cat0 + (long)(H + 1) * PF, c0bs, nfr_d, nb, Fmax, st); });
        float* catA = aF.f((size_t)nb * C1 * PF);
        float* catB = aF.f((size_t)nb * C1 * PF);
        const long c1bs = (long)C1 * PF;
        // bf16 MFMA mode (k82_set_precision): from here on -- decoder and generator convs -- the 2x2-shaped
        // convs take their bf16 weight image; everything before (durations, F0/N curves, text encoder) and
        // conv_post (1x4 shape) stay on the exact-fp32 instruction.  The guard clears the flag on any exit.
        struct Bf16Scope { bool& f; bool& z; bool& d; ~Bf16Scope() { f = false; z = false; d = false; } } bf16_scope{bf16_now, bf16_zone, dec_zone};
        bf16_now = precision != 0;
        bf16_zone = true;
        dec_zone = true;
        resblk1d(enc, cat0, c0bs, PF, lF, Fmax, hb, sumF, catA, c1bs, PF, gbd, Gd, aF);
        tap("dec_encode", catA, 1024, PF, c1bs, nb);
        for (float* cat : {catA, catB}) {
            run_conv(cp(asr_res, cat0, c0bs, PF, cat + (long)1024 * PF, c1bs, PF, lF, lF, nb, Fmax), hb);
            timed(FAM_OTHER, 0, [&] { launch_copy_rows(cat0 + (long)H * PF, c0bs, PF, cat + (long)1088 * PF, c1bs, PF, 2, lF, nb, Fmax, st); });
        }
        resblk1d(dec[0], catA, c1bs, PF, lF, Fmax, hb, sumF, catB, c1bs, PF, gbd, Gd, aF);
        resblk1d(dec[1], catB, c1bs, PF, lF, Fmax, hb, sumF, catA, c1bs, PF, gbd, Gd, aF);
        resblk1d(dec[2], catA, c1bs, PF, lF, Fmax, hb, sumF, catB, c1bs, PF, gbd, Gd, aF);
        float* xdec = aF.f((size_t)nb * 512 * P2);
        resblk1d(dec[3], catB, c1bs, PF, lF, Fmax, hb, sumF, xdec, (long)512 * P2, P2, gbd, Gd, aF);
        tap("dec_out", xdec, 512, P2, (long)512 * P2, nb);
        dec_zone = false;

        // ---- harmonic source + STFT (rows a14, a15)
        float* ph = aF.f((size_t)nb * K82_HARM * P2);
        float* hsrc = aF.f((size_t)nb * P600);
        const float* nz = nullptr;
        const uint64_t* seeds_d = nullptr;
        if (noise && noise->noise) nz = to_device(noise->noise, (size_t)aoff[nb] * K82_HARM, aF);
        else if (noise && noise->seeds) seeds_d = to_device(noise->seeds, (size_t)nb, aF);
        timed(FAM_SRC, 0, [&] { launch_src_phase(f0, P2, nfr_d, nb, ph, (long)K82_HARM * P2, P2, st); });
        timed(FAM_SRC, 0, [&] { launch_src_har(f0, P2, ph, (long)K82_HARM * P2, P2, nfr_d, nz, aoff_d, seeds_d, lin_w, lin_b, hsrc, P600, nb, Fmax, st); });
        tap("src_ph", ph, K82_HARM, P2, (long)K82_HARM * P2, nb);
        tap("har_source", hsrc, 1, (int)P600, P600, nb);
        float* har = aF.f((size_t)nb * 22 * P120);
        float* harp = aF.f((size_t)nb * 144 * P20);
        HIPCHK(hipMemsetAsync(harp, 0, (size_t)nb * 144 * P20 * sizeof(float), st));
        timed(FAM_SRC, 0, [&] { launch_stft(hsrc, P600, nfr_d, nb, Fmax, har, (long)22 * P120, P120, harp, (long)144 * P20, P20, st); });
        if (forced.count("har")) {
            apply_forced("har", har, P120, 22, 120 * nfr_h[0] + 1);
            // the stride-6 polyphase image the first noise conv reads: harp[ci*6 + r][q] = har[ci][6q + r - 3], 0 outside
            const HostT& t = forced["har"];
            const int L = 120 * nfr_h[0] + 1, Q = 20 * nfr_h[0] + 1;
            std::vector<float> hp((size_t)132 * Q, 0.f);
            for (int ci = 0; ci < 22; ++ci)
                for (int r = 0; r < 6; ++r)
                    for (int q = 0; q < Q; ++q) {
                        const int c = 6 * q + r - 3;
                        if (c >= 0 && c < L) hp[((size_t)ci * 6 + r) * Q + q] = t.v[(size_t)ci * L + c];
                    }
            HIPCHK(hipMemcpy2DAsync(harp, (size_t)P20 * sizeof(float), hp.data(), (size_t)Q * sizeof(float), (size_t)Q * sizeof(float), 132, hipMemcpyHostToDevice, st));
            HIPCHK(hipStreamSynchronize(st));               // hp is a stack temporary
        }
        tap("har", har, 22, P120, (long)22 * P120, nb);

        // ---- generator (rows a16, a17)
        const long b0 = (long)256 * P20, b1 = (long)128 * P120;
        float* xsrc0 = aF.f((size_t)nb * b0);
        float* x0 = aF.f((size_t)nb * b0);
        float* xr0 = aF.f((size_t)nb * std::max(b0, (long)ups0.cout * P2));   // also holds the phase rows of ups0 before the resblocks use it
        float* t10 = aF.f((size_t)nb * b0);
        float* xs0 = aF.f((size_t)nb * b0);
        // a noise conv also leaves the statistics of its output for the first AdaIN of the noise resblock
        auto noise_conv = [&](ConvP pn, int C, LenSpec ls, int Lmax, int goff, double elems) -> float4* {
            if (!fused_stats) { run_conv(pn, hb); return nullptr; }
            const int snt = (Lmax + 31) / 32;
            float2* part = (float2*)aF.alloc((size_t)nb * C * snt * sizeof(float2));
            pn.stat = part; pn.sbs = (long)C * snt; pn.snt = snt;
            run_conv(pn, hb);
            float4* nm = (float4*)aF.alloc((size_t)nb * ru(C, 16) * sizeof(float4));
            timed(FAM_STATS, elems / 4.0, [&] { launch_adain_finalize(part, (long)C * snt, snt, ls, C, nb, gbd, Gd, goff, nm, ru(C, 16), st); });
            return nm;
        };
        float4* nsh0 = noise_conv(cp(nconv0, harp, (long)144 * P20, P20, xsrc0, b0, P20, l20p, l20, nb, 20 * Fmax), 256, l20, 20 * Fmax,
                                  nres[0].n1[0].goff, 20 * sumF * 256);
        tap("gen0_noise_conv", xsrc0, 256, P20, b0, nb);
        resblock1(nres[0], xsrc0, xsrc0, t10, b0, P20, l20, 20 * Fmax, hb, 20 * sumF, nullptr, 0, 1.f, aF, &nsh0);
        tap("gen0_xsource", xsrc0, 256, P20, b0, nb);
        {   // ConvTranspose1d as a conv over 10 x 256 phase rows (wide epilogue), then the phase interleave + noise add
            float* yp = xr0;
            ConvP pu = cp(ups0, xdec, (long)512 * P2, P2, yp, (long)ups0.cout * P2, P2, l2F, l2F, nb, 2 * Fmax);
            pu.tr = TR_SCALE; pu.in_scale = 1.f; pu.act = ACT_LEAKY; pu.slope = 0.1f;
            pu.up_s = 0; pu.up_cout_shift = 0;
            run_conv(pu, hb);
            timed(FAM_OTHER, 0, [&] { REQUIRE(launch_phase_interleave(yp, (long)ups0.cout * P2, P2, x0, b0, P20, xsrc0, b0, P20, l2F, 256, 10, 0, nb, 2 * Fmax, st) == 0, K82_EINVAL, "transposed-conv rate without an interleave kernel"); });
        }
        float4* sh0 = nullptr;
        for (int j = 0; j < 3; ++j) resblock1(rb[j], x0, xr0, t10, b0, P20, l20, 20 * Fmax, hb, 20 * sumF, xs0, j == 0, 1.0f / 3.0f, aF, &sh0);
        tap("gen0_out", xs0, 256, P20, b0, nb);

        float* xsrc1 = aF.f((size_t)nb * b1);
        float* x1 = aF.f((size_t)nb * b1);
        float* xr1 = aF.f((size_t)nb * std::max(b1, (long)ups1.cout * P20));  // also holds the phase rows of ups1
        float* t11 = aF.f((size_t)nb * b1);
        float* xs1 = aF.f((size_t)nb * b1);
        float4* nsh1 = noise_conv(cp(nconv1, har, (long)22 * P120, P120, xsrc1, b1, P120, l120, l120, nb, 120 * Fmax + 1), 128, l120,
                                  120 * Fmax + 1, nres[1].n1[0].goff, (120 * sumF + nb) * 128);
        resblock1(nres[1], xsrc1, xsrc1, t11, b1, P120, l120, 120 * Fmax + 1, hb, 120 * sumF + nb, nullptr, 0, 1.f, aF, &nsh1);
        tap("gen1_xsource", xsrc1, 128, P120, b1, nb);
        {
            float* yp = xr1;
            ConvP pu = cp(ups1, xs0, b0, P20, yp, (long)ups1.cout * P20, P20, l20, l20, nb, 20 * Fmax);
            pu.act = ACT_LEAKY; pu.slope = 0.1f;
            pu.up_s = 0; pu.up_cout_shift = 0;
            run_conv(pu, hb);
            timed(FAM_OTHER, 0, [&] { REQUIRE(launch_phase_interleave(yp, (long)ups1.cout * P20, P20, x1, b1, P120, xsrc1, b1, P120, l20, 128, 6, 1, nb, 20 * Fmax, st) == 0, K82_EINVAL, "transposed-conv rate without an interleave kernel"); });
            timed(FAM_OTHER, 0, [&] { launch_reflect_fix(x1, b1, P120, xsrc1, b1, P120, 128, nb, st); });
        }
        float4* sh1 = nullptr;
        for (int j = 0; j < 3; ++j) resblock1(rb[3 + j], x1, xr1, t11, b1, P120, l120, 120 * Fmax + 1, hb, 120 * sumF + nb, xs1, j == 0, 1.0f / 3.0f, aF, &sh1);
        tap("gen1_out", xs1, 128, P120, b1, nb);

        float* cpo = aF.f((size_t)nb * 22 * P120);
        {
            ConvP pc = cp(conv_post, xs1, b1, P120, cpo, (long)22 * P120, P120, l120, l120, nb, 120 * Fmax + 1);
            pc.act = ACT_LEAKY; pc.slope = 0.01f;
            run_conv(pc, hb);
        }
        tap("conv_post", cpo, 22, P120, (long)22 * P120, nb);

        // ---- iSTFT (row a18)
        hipPointerAttribute_t at;
        hipError_t e = hipPointerGetAttributes(&at, audio);
        const bool on_dev = (e == hipSuccess && at.type == hipMemoryTypeDevice);
        if (e != hipSuccess) (void)hipGetLastError();
        float* adev = on_dev ? audio : aF.f((size_t)aoff[nb]);
        timed(FAM_ISTFT, sumF * 120.0 * (22 + 5) * 4.0, [&] { launch_istft(cpo, (long)22 * P120, P120, nfr_d, nb, Fmax, adev, aoff_d, st); });
        if (!on_dev) HIPCHK(hipMemcpyAsync(audio, adev, (size_t)aoff[nb] * sizeof(float), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipGetLastError());
        aF.fold();
    }
};

// ============================================================================== C-ABI
namespace {
template <class F> int guarded(k82_handle* h, void* stream, F&& f) {
    if (!h) return K82_EINVAL;
    std::lock_guard<std::mutex> lk(h->mu);
    try {
        HIPCHK(hipSetDevice(h->device));
        h->st = stream ? (hipStream_t)stream : h->own_stream;
        f();
        return K82_OK;
    } catch (const K82Error& e) {
        h->err = e.what();
        return e.code;
    } catch (const std::bad_alloc&) {
        h->err = "host allocation failed";
        return K82_ENOMEM;
    } catch (const std::exception& e) {
        h->err = e.what();
        return K82_EINVAL;
    }
}
}  // namespace

extern "C" {

int k82_create(const k82_config* cfg, int device, k82_handle** out) {
    if (!cfg || !out) { g_create_error = "cfg and out must not be NULL"; return K82_EINVAL; }
    const k82_config want = {178, 512, 128, 3, 50, 5, 768, 12, 2048, 12, 512, 128};
    if (std::memcmp(cfg, &want, sizeof(want)) != 0 && !(cfg->n_token >= 1 && cfg->hidden_dim == 512 && cfg->style_dim == 128 &&
        cfg->n_layer == 3 && cfg->max_dur >= 1 && cfg->max_dur <= 64 && cfg->text_kernel == 5 && cfg->bert_hidden == 768 &&
        cfg->bert_heads == 12 && cfg->bert_inter == 2048 && cfg->bert_layers >= 1 && cfg->bert_max_pos >= 512 && cfg->bert_embed == 128)) {
        g_create_error = "unsupported architecture: the engine implements hexgrad/Kokoro-82M (hidden 512, style 128, 3 layers, ALBERT 768/12/2048)";
        return K82_EINVAL;
    }
    k82_handle* h = nullptr;
    try {
        h = new k82_handle();
        h->cfg = *cfg;
        h->device = device;
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
        REQUIRE(source_init_constants() == 0, K82_EHIP, "hipMemcpyToSymbol failed");
        h->scratch = (unsigned char*)scratch_address(device);      // set-up time, device current: never on the no-alloc path
        *out = h;
        return K82_OK;
    } catch (const K82Error& e) {
        g_create_error = e.what();
        delete h;
        return e.code;
    } catch (const std::exception& e) {
        g_create_error = e.what();
        delete h;
        return K82_ENOMEM;
    }
}

// A second execution context on the weights of `parent`: own stream, own workspace arenas, own phase-1 state, own error
// words -- nothing is uploaded again.  Calls on the two handles do not serialise against each other, so a host thread per
// handle overlaps one batch's token-rate phase (small, latency-bound launches and its device -> host sync) with another
// batch's frame-rate phase on the same GPU.  Same kernels, same launch rules: a batch gives the same bits on either handle.
int k82_fork(k82_handle* parent, k82_handle** out) {
    if (!parent || !out) { g_create_error = "parent and out must not be NULL"; return K82_EINVAL; }
    k82_handle* h = nullptr;
    std::lock_guard<std::mutex> lk(parent->mu);
    try {
        REQUIRE(parent->finalized, K82_ESTATE, "k82_fork needs a finalized handle");
        REQUIRE(!parent->parent, K82_ESTATE, "k82_fork: fork the handle that owns the weights, not one of its forks");
        h = new k82_handle();
        h->cfg = parent->cfg;
        h->device = parent->device;
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
        static_cast<K82Weights&>(*h) = static_cast<const K82Weights&>(*parent);
        h->scratch = parent->scratch;
        h->cus = parent->cus;
        h->kflush_div = parent->kflush_div; h->dec_ksplit = parent->dec_ksplit; h->splitk_on = parent->splitk_on; h->skcap = parent->skcap;
        h->skbuf = (float*)h->wmalloc(h->skcap * sizeof(float));
        h->t1_b16 = parent->t1_b16; h->ring_off = parent->ring_off; h->bm256_off = parent->bm256_off; h->fused_stats = parent->fused_stats; h->precise = parent->precise;
        h->lstm_split_max_b = parent->lstm_split_max_b; h->lstm_spin_limit = parent->lstm_spin_limit; h->conv_spin_limit = parent->conv_spin_limit;
        h->precision = parent->precision;
        h->lstm_err = (int*)h->wmalloc(2 * sizeof(int));        // (wallocs of a fork hold only its own words)
        h->conv_err = h->lstm_err + 1;
        HIPCHK(hipMemset(h->lstm_err, 0, 2 * sizeof(int)));
        h->finalized = true;
        h->parent = parent;
        parent->forks.fetch_add(1);
        *out = h;
        return K82_OK;
    } catch (const K82Error& e) {
        g_create_error = e.what();
        parent->err = e.what();
        if (h) { for (void* p : h->wallocs) (void)hipFree(p); if (h->own_stream) (void)hipStreamDestroy(h->own_stream); }
        delete h;
        return e.code;
    } catch (const std::exception& e) {
        g_create_error = e.what();
        delete h;
        return K82_ENOMEM;
    }
}

int k82_destroy(k82_handle* h) {
    if (!h) return K82_EINVAL;
    if (h->forks.load() > 0) {           // its forks read this handle's weights
        std::lock_guard<std::mutex> lk(h->mu);
        h->err = "k82_destroy: " + std::to_string(h->forks.load()) + " handle(s) made by k82_fork are still alive; destroy them first";
        return K82_ESTATE;
    }
    {
        std::lock_guard<std::mutex> lk(h->mu);
        if (h->parent) h->parent->forks.fetch_sub(1);
        (void)hipSetDevice(h->device);
        (void)hipDeviceSynchronize();
        for (void* p : h->wallocs) (void)hipFree(p);
        h->aT.release();
        h->aF.release();
        for (auto e : h->evpool) (void)hipEventDestroy(e);
        if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    }
    delete h;
    return K82_OK;
}

const char* k82_last_error(const k82_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int k82_set_tensor(k82_handle* h, const char* name, const float* data, const int64_t* shape, int ndim) {
    return guarded(h, nullptr, [&] {
        REQUIRE(name && data && shape && ndim >= 1 && ndim <= 4, K82_EINVAL, "bad k82_set_tensor arguments");
        REQUIRE(!h->finalized, K82_ESTATE, "weights are already finalized");
        HostT t;
        size_t n = 1;
        for (int i = 0; i < ndim; ++i) { REQUIRE(shape[i] >= 1, K82_EINVAL, "bad shape"); t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
        t.v.assign(data, data + n);
        h->host[name] = std::move(t);
    });
}

int k82_finalize(k82_handle* h) {
    return guarded(h, nullptr, [&] { h->finalize(); });
}

int k82_durations(k82_handle* h, const int64_t* ids, const int32_t* lens, int B, int Tmax, const float* ref_s,
                  const float* speed, const int32_t* forced_dur, int32_t* pred_dur, int64_t* n_frames, void* stream) {
    return guarded(h, stream, [&] { h->durations(ids, lens, B, Tmax, ref_s, speed, forced_dur, pred_dur, n_frames); });
}

int k82_decode(k82_handle* h, const k82_noise* noise, float* audio, int64_t capacity, int64_t* audio_off, void* stream) {
    return guarded(h, stream, [&] { h->decode(noise, audio, capacity, audio_off); });
}

int k82_synthesize(k82_handle* h, const int64_t* ids, const int32_t* lens, int B, int Tmax, const float* ref_s,
                   const float* speed, const k82_noise* noise, int32_t* pred_dur, float* audio, int64_t capacity,
                   int64_t* audio_off, void* stream) {
    return guarded(h, stream, [&] {
        h->durations(ids, lens, B, Tmax, ref_s, speed, nullptr, pred_dur, nullptr);
        h->decode(noise, audio, capacity, audio_off);
    });
}

int k82_pcm16(k82_handle* h, const float* src, int16_t* dst, int64_t n, void* stream) {
    return guarded(h, stream, [&] {
        REQUIRE(src && dst && n >= 0, K82_EINVAL, "bad k82_pcm16 arguments");
        h->aF.reset();
        hipPointerAttribute_t at;
        hipError_t e = hipPointerGetAttributes(&at, src);
        const bool sdev = (e == hipSuccess && at.type == hipMemoryTypeDevice);
        if (e != hipSuccess) (void)hipGetLastError();
        e = hipPointerGetAttributes(&at, dst);
        const bool ddev = (e == hipSuccess && at.type == hipMemoryTypeDevice);
        if (e != hipSuccess) (void)hipGetLastError();
        const float* s = src;
        if (!sdev) {
            float* t = h->aF.f((size_t)n);
            HIPCHK(hipMemcpyAsync(t, src, (size_t)n * sizeof(float), hipMemcpyHostToDevice, h->st));
            s = t;
        }
        int16_t* d = ddev ? dst : (int16_t*)h->aF.alloc((size_t)n * sizeof(int16_t));
        launch_pcm16(s, d, n, h->st);
        if (!ddev) HIPCHK(hipMemcpyAsync(dst, d, (size_t)n * sizeof(int16_t), hipMemcpyDeviceToHost, h->st));
        HIPCHK(hipStreamSynchronize(h->st));
    });
}

int k82_pcm16_static(k82_handle* h, const float* src, int16_t* dst, int64_t n, void* stream) {
    return guarded(h, stream, [&] {
        REQUIRE(src && dst && n >= 0, K82_EINVAL, "bad k82_pcm16_static arguments");
        REQUIRE(n <= kPcmStaticMax, K82_EINVAL, "k82_pcm16_static converts at most " + std::to_string(kPcmStaticMax) +
                                                    " samples per call (static staging buffer); got " + std::to_string(n));
        REQUIRE(h->scratch, K82_ESTATE, "static scratch address was not resolved at k82_create");
        if (n == 0) return;
        const bool sdev = k82_handle::is_device_ptr(src), ddev = k82_handle::is_device_ptr(dst);
        std::lock_guard<std::mutex> lk(g_scratch_mu[h->device]);      // other handles of this process on this GPU
        float* sf = (float*)h->scratch;                               // [0, 16 MiB): float staging
        int16_t* sd = (int16_t*)(h->scratch + (size_t)kPcmStaticMax * sizeof(float));   // then the int16 staging
        const float* s = src;
        if (!sdev) {
            HIPCHK(hipMemcpyAsync(sf, src, (size_t)n * sizeof(float), hipMemcpyHostToDevice, h->st));
            s = sf;
        }
        int16_t* d = ddev ? dst : sd;
        launch_pcm16(s, d, n, h->st);
        if (!ddev) HIPCHK(hipMemcpyAsync(dst, d, (size_t)n * sizeof(int16_t), hipMemcpyDeviceToHost, h->st));
        HIPCHK(hipStreamSynchronize(h->st));
        HIPCHK(hipGetLastError());
    });
}

int k82_prof_enable(k82_handle* h, int enable) {
    return guarded(h, nullptr, [&] {
        h->prof_drain();
        h->prof_on = enable != 0;
        if (enable) for (int i = 0; i < NFAM; ++i) { h->fam_ms[i] = 0; h->fam_n[i] = 0; h->fam_work[i] = 0; }
    });
}

int k82_prof_read(k82_handle* h, const char* name, double* ms, int64_t* launches, double* work) {
    return guarded(h, nullptr, [&] {
        REQUIRE(name, K82_EINVAL, "name must not be NULL");
        h->prof_drain();
        double m = 0, w = 0; int64_t n = 0; bool found = false;
        if (std::strcmp(name, "lstm_fallbacks") == 0) { n = h->lstm_fallbacks; found = true; }
        // launches of the large-tile conv families since the process started (bit-identity tests assert that they ran)
        const char* const shapes[CONV_NSHAPE] = {"conv_shape_bm256", "conv_shape_ring_wn2", "conv_shape_ring", "conv_shape_lds3", "conv_shape_ksplit", "conv_shape_splitk"};
        for (int i = 0; i < CONV_NSHAPE; ++i)
            if (std::strcmp(name, shapes[i]) == 0) { n = conv_shape_launches(i); found = true; }
        // workspace of the largest pass so far / capacity held now, in bytes (sizing of k82_reserve, tests)
        if (std::strcmp(name, "arena_token") == 0) { w = (double)std::max(h->aT.high, h->aT.pass_bytes); n = (int64_t)h->aT.capacity(); found = true; }
        if (std::strcmp(name, "arena_frame") == 0) { w = (double)std::max(h->aF.high, h->aF.pass_bytes); n = (int64_t)h->aF.capacity(); found = true; }
        for (int i = 0; i < NFAM; ++i) {
            const bool match = std::strcmp(name, kFamNames[i]) == 0 || (std::strcmp(name, "conv_mfma") == 0 && i <= FAM_CONV14);
            if (match) { m += h->fam_ms[i]; n += h->fam_n[i]; w += h->fam_work[i]; found = true; }
        }
        REQUIRE(found, K82_ENOENT, std::string("unknown kernel family: ") + name);
        if (ms) *ms = m;
        if (launches) *launches = n;
        if (work) *work = w;
    });
}

int k82_debug_enable(k82_handle* h, int enable) {
    return guarded(h, nullptr, [&] { h->debug_on = enable != 0; if (!enable) h->dbg.clear(); });
}
int k82_debug_info(k82_handle* h, const char* name, int64_t dims[3]) {
    return guarded(h, nullptr, [&] {
        auto it = h->dbg.find(name ? name : "");
        REQUIRE(it != h->dbg.end(), K82_ENOENT, std::string("no debug tensor named ") + (name ? name : "(null)"));
        for (int i = 0; i < 3; ++i) dims[i] = it->second.dims[i];
    });
}
int k82_debug_copy(k82_handle* h, const char* name, float* dst) {
    return guarded(h, nullptr, [&] {
        auto it = h->dbg.find(name ? name : "");
        REQUIRE(it != h->dbg.end() && dst, K82_ENOENT, "no such debug tensor");
        std::memcpy(dst, it->second.v.data(), it->second.v.size() * sizeof(float));
    });
}

int k82_debug_force(k82_handle* h, const char* name, const float* data, int64_t rows, int64_t cols) {
    return guarded(h, nullptr, [&] {
        REQUIRE(name && (!std::strcmp(name, "F0") || !std::strcmp(name, "N") || !std::strcmp(name, "har")), K82_ENOENT,
                "k82_debug_force knows the stages F0, N and har");
        if (!data) { h->forced.erase(name); return; }
        REQUIRE(rows >= 1 && cols >= 1, K82_EINVAL, "bad shape");
        HostT t;
        t.shape = {rows, cols};
        t.v.assign(data, data + (size_t)rows * cols);
        h->forced[name] = std::move(t);
    });
}

int k82_reserve(k82_handle* h, int B, int Tmax, int64_t frames) {
    return guarded(h, nullptr, [&] {
        REQUIRE(h->finalized, K82_ESTATE, "k82_finalize has not been called");
        REQUIRE(B >= 1 && Tmax >= 2 && Tmax <= K82_MAX_TOKENS && frames >= 1 && frames <= 50L * K82_MAX_TOKENS, K82_EINVAL,
                "k82_reserve: B >= 1, 2 <= Tmax <= 512, 1 <= frames <= 50 * 512");
        // Measured workspace (k82_prof_read "arena_token" / "arena_frame" over batches from 1 x 8 to 64 x 256 tokens):
        // the token-rate pass takes 61-84 KB per padded token (B * roundup(Tmax, 32)), the frame-rate pass 0.584-0.599 MB
        // per padded frame (B * frames) -- tools/arena_probe.py; 5 % and a few MB on top.  A pass that still outgrows the reservation
        // grows the arena on demand as before.
        const size_t tok = (size_t)B * (size_t)ru(Tmax, 32), fr = (size_t)B * (size_t)frames;
        h->have_phase1 = false;
        h->aT.reserve((size_t)(1.05 * 86.0e3 * (double)tok + 8.0e6));
        h->aF.reserve((size_t)(1.05 * 0.60e6 * (double)fr + 8.0e6));
    });
}

int k82_set_precision(k82_handle* h, int mode) {
    return guarded(h, nullptr, [&] {
        REQUIRE(mode >= 0 && mode <= 2, K82_EINVAL, "precision mode must be 0 (fp32), 1 (bf16 decoder/generator MFMA) or 2 (bf16x3 split)");
        h->precision = mode;
    });
}

int k82_op_conv1d(k82_handle* h, const float* x, const float* w, const float* bias, float* y, int B, int cin, int cout,
                  int L, int taps, int dil, int act, float slope, const float* alpha, int iters, float* ms) {
    return guarded(h, nullptr, [&] {
        REQUIRE(x && w && y && B >= 1 && cin >= 1 && cout >= 1 && L >= 1 && taps >= 1 && (taps & 1) && dil >= 1, K82_EINVAL, "bad k82_op_conv1d arguments");
        REQUIRE((taps - 1) * dil <= 64, K82_EINVAL, "halo too large");
        std::vector<float> W(w, w + (size_t)cout * cin * taps);
        std::vector<float> bv;
        if (bias) bv.assign(bias, bias + cout);
        const size_t mark = h->wallocs.size();
        ConvW cw = h->mk_conv_raw(W, bias ? &bv : nullptr, cout, cin, taps);
        const int P = ru(L, 4);
        h->aF.reset();
        h->B = B;
        h->have_phase1 = false;
        float* xd = h->aF.f((size_t)B * cw.cin_p * P);
        float* yd = h->aF.f((size_t)B * cout * P);
        HIPCHK(hipMemsetAsync(xd, 0, (size_t)B * cw.cin_p * P * sizeof(float), h->st));
        for (int b = 0; b < B; ++b)
            HIPCHK(hipMemcpy2DAsync(xd + (size_t)b * cw.cin_p * P, P * sizeof(float), x + (size_t)b * cin * L, L * sizeof(float), L * sizeof(float), cin, hipMemcpyHostToDevice, h->st));
        float *ad = nullptr, *iad = nullptr;
        if (act == ACT_SNAKE) {
            REQUIRE(alpha, K82_EINVAL, "snake needs alpha");
            std::vector<float> a(cw.cin_p, 1.f), ia(cw.cin_p, 1.f);
            for (int c = 0; c < cin; ++c) { a[c] = alpha[c]; ia[c] = 1.0f / alpha[c]; }
            ad = h->aF.f(cw.cin_p); iad = h->aF.f(cw.cin_p);
            HIPCHK(hipMemcpyAsync(ad, a.data(), cw.cin_p * sizeof(float), hipMemcpyHostToDevice, h->st));
            HIPCHK(hipMemcpyAsync(iad, ia.data(), cw.cin_p * sizeof(float), hipMemcpyHostToDevice, h->st));
            HIPCHK(hipStreamSynchronize(h->st));
        }
        const LenSpec ls{nullptr, 0, L};
        ConvP p = h->cp(cw, xd, (long)cw.cin_p * P, P, yd, (long)cout * P, P, ls, ls, B, L, dil);
        p.act = act; p.slope = slope; p.alpha = ad; p.inv_alpha = iad;
        p.wb = h->precision == 1 ? cw.wb : h->precision == 2 ? cw.wb2 : nullptr;    // the op hook follows k82_set_precision directly
        p.wb_split = h->precision == 2 && cw.wb2;
        k82_handle::check_conv(p);
        launch_conv(p, h->st);
        if (iters > 1 || ms) {
            hipEvent_t a, b;
            HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
            const int n = std::max(iters, 1);
            HIPCHK(hipEventRecord(a, h->st));
            for (int i = 0; i < n; ++i) launch_conv(p, h->st);
            HIPCHK(hipEventRecord(b, h->st));
            HIPCHK(hipEventSynchronize(b));
            float t = 0.f;
            HIPCHK(hipEventElapsedTime(&t, a, b));
            if (ms) *ms = t / n;
            (void)hipEventDestroy(a); (void)hipEventDestroy(b);
        }
        for (int b = 0; b < B; ++b)
            HIPCHK(hipMemcpy2DAsync(y + (size_t)b * cout * L, L * sizeof(float), yd + (size_t)b * cout * P, P * sizeof(float), L * sizeof(float), cout, hipMemcpyDeviceToHost, h->st));
        HIPCHK(hipStreamSynchronize(h->st));
        HIPCHK(hipGetLastError());
        for (size_t i = mark; i < h->wallocs.size(); ++i) (void)hipFree(h->wallocs[i]);
        h->wallocs.resize(mark);
    });
}

}  // extern "C"
