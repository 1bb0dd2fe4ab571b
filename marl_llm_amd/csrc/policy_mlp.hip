// Fused policy MLP for the device-resident rollout: the reference's actor
// (/root/reference/marl_llm/algorithm/utils/networks.py:6-44 -- fc1..fc4, leaky-ReLU x3, tanh out; hidden_dim = 180,
// /root/reference/marl_llm/cfg/assembly_cfg.py:185) evaluated on the env's [E*N, 192] fp32 observation rows in ONE kernel:
// bf16 MFMA (v_mfma_f32_32x32x16_bf16, fp32 accumulate), activations never leave the registers between layers.
//
// Orientation: every layer is computed transposed, H^T = W . X^T, with the 32 batch rows of a wavefront's tile on the lanes
// (MFMA column) and the output features in the 16 accumulator registers x 2 lane halves (MFMA row).  The accumulator tile
// of layer l is then directly the B operand of layer l+1 (the product sums over its ROW index): registers 8s..8s+7 of
// feature tile kt, converted pairwise to bf16, are the fragment of k-step (kt, s); inside a step the k order is permuted
// (element j of lane half h is feature 32 kt + 16 s + 8 (j >> 2) + 4 h + (j & 3)), so the weights are packed on the host
// in exactly that order -- no transposes and no cross-lane movement of activations in the whole network (LDS only holds
// the weights).
// Fragment maps (cdna_hip_programming.md section 3): A[row = l & 31][k = 8 (l >> 5) + j], B[k = 8 (l >> 5) + j][col = l & 31],
// C/D col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5).
//
// Numerics: weights and layer inputs are rounded to bf16 (round-to-nearest-even), sums and biases are fp32: the same
// contract as torch.autocast(bfloat16) on the reference module.  tests/test_gpu_policy.py states the tolerance.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "swarm_policy.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;

constexpr int kKP = 192;          // padded input / hidden width: 12 k-steps of 16, 6 feature tiles of 32
constexpr int kKS = kKP / 16;     // k-steps per layer
constexpr int kMT = kKP / 32;     // feature tiles per hidden layer
// wavefronts (32-row tiles) per workgroup = rows that share one pass of the weight stream.  bf16: four (eight: +3 %);
// bf16x3 -- twice the weight stream -- eight (155 / 179 us on 262144 bf16 / fp32 rows instead of 200 / 221 with four)
constexpr int waves_of(bool x3) { return x3 ? 8 : 4; }

struct MlpParams {
    const bf8 *w1, *w2, *w3, *w4;          // packed fragments: [feature tile][k-step][lane] x 8 bf16
    const bf8 *l1, *l2, *l3, *l4;          // the same layout, LOW parts w - bf16(w) (precision mode bf16x3 only)
    const float *b1, *b2, *b3, *b4;        // padded biases (kKP, kKP, kKP, 32)
    int in_dim, act_dim;
    long long rows;
    // exploration noise of the rollout (agents.py:93-96: action += scale * N(0, 1); clamp to [-1, 1]), fused into the epilogue:
    // counter-based generator keyed by (seed, step, row), two normals per hash (Box-Muller)
    float noise_scale;
    unsigned long long noise_key;      // mix64(seed, step), host-side
};

__host__ __device__ inline unsigned long long pmix64(unsigned long long z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ int feat_of(int mt, int reg, int h) { return 32 * mt + (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// A workgroup = kWaves wavefronts, one 32-row tile each.  The packed weights stream through LDS in CHUNKS of three feature
// tiles (36 fragments = 36 KB; the output layer's 12 fragments are the last chunk), two LDS buffers: while the waves run
// the MFMAs of chunk c out of one buffer, every thread already holds chunk c+1 in registers (9 x 16 B, global loads issued
// before the MFMA loop) and writes it to the other buffer afterwards -- one barrier per chunk, the L2 latency of the weight
// stream hidden behind the matrix cores.  (Measured at 262144 rows: weights straight from L2 per wave 248 us; whole layers
// staged with the waves waiting 136 us, of which 47 us staging and 50 us observation loads.)
// Biases: layer 1's initialise the accumulators; for the later layers the (padded) hidden feature kOne is held at 1.0 --
// bias 1 / zero weights in layer 1, a 1.0 on the diagonal afterwards -- and the biases sit in that column of the packed
// weights (bf16, as under torch.autocast), so no bias loads in the chain.
constexpr int kChunkFrags = 3 * kKS;                                        // 36 fragments
constexpr int kChunks = 7;                                                  // 2 per hidden layer + the output layer
constexpr int pre_of(int waves) { return (kChunkFrags * 64 + 64 * waves - 1) / (64 * waves); }   // 16-byte pieces per thread and chunk
constexpr int kSmemBytes = 2 * kChunkFrags * 64 * 16;                       // 72 KB

// TPW = row tiles (32 rows each) per wave.  With one tile every MFMA needs its own 1 KB weight-fragment read from LDS (four
// SIMDs at full MFMA rate would ask for exactly the LDS bandwidth, 128 B/clk); with two tiles each read feeds two MFMAs at
// the price of ~390 VGPRs, i.e. one wave per SIMD -- measured slower, so TPW = 1 is what runs (see swarm_policy_forward).
// X3 ("bf16x3"): every operand is split into a bf16 high part and a bf16 low part (x = hi + lo to 16 significant bits) and a
// product is three MFMAs, w_hi x_hi + w_hi x_lo + w_lo x_hi, accumulated in fp32 -- the rollout then follows the reference's
// fp32 actor to ~1e-4 instead of bf16's 4e-2.  The weight stream alternates high and low chunks (same chunk size, same two
// LDS buffers: twice as many chunk steps); a high chunk meets both activation parts, a low chunk the high part only.
template <bool IN_BF16, int TPW, bool X3>
__global__ void __launch_bounds__(64 * waves_of(X3), (TPW == 1 && !X3) ? 2 : 1)
k_policy_mlp(const MlpParams P, const void *__restrict__ obs_, float *__restrict__ act)
{
    constexpr int kWaves = waves_of(X3), kPre = pre_of(kWaves);
    extern __shared__ __align__(16) unsigned char smem_raw[];
    bf8 *const buf[2] = {reinterpret_cast<bf8 *>(smem_raw), reinterpret_cast<bf8 *>(smem_raw) + kChunkFrags * 64};
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long long row0 = ((long long)blockIdx.x * kWaves + wave) * (32 * TPW) + r;    // this lane's row in tile 0
    const bool wave_on = row0 - r < P.rows;                                             // idle waves still stage and take the barriers

    bf8 pre[kPre];
    // chunk step cc: X3 interleaves (chunk c, high) and (chunk c, low)
    constexpr int kSteps = X3 ? 2 * kChunks : kChunks;
    auto issue = [&](int cc) {                                              // chunk step cc of the weight stream -> registers
        const int c = X3 ? cc >> 1 : cc;
        const bool low = X3 && (cc & 1);
        const bf8 *b1_ = low ? P.l1 : P.w1, *b2_ = low ? P.l2 : P.w2, *b3_ = low ? P.l3 : P.w3, *b4_ = low ? P.l4 : P.w4;
        const bf8 *w = c < 2 ? b1_ + (size_t)c * kChunkFrags * 64 : c < 4 ? b2_ + (size_t)(c - 2) * kChunkFrags * 64
                     : c < 6 ? b3_ + (size_t)(c - 4) * kChunkFrags * 64 : b4_;
        const int n = (c < 6 ? kChunkFrags : kKS) * 64;
#pragma unroll
        for (int k = 0; k < kPre; ++k) { const int q = tid + k * 64 * kWaves; if (q < n) pre[k] = w[q]; }
    };
    auto commit = [&](int cc) {                                             // registers -> LDS buffer of chunk step cc
        const int c = X3 ? cc >> 1 : cc;
        const int n = (c < 6 ? kChunkFrags : kKS) * 64;
#pragma unroll
        for (int k = 0; k < kPre; ++k) { const int q = tid + k * 64 * kWaves; if (q < n) buf[cc & 1][q] = pre[k]; }
    };

    issue(0);
    // the 12 B fragments of layer 1, straight from the observation row (natural k order); later `pk` holds the packed
    // activations of the previous layer (k-step = 2 * feature tile + s)
    bf8 pk[TPW][kKS];
    bf8 pl[X3 ? TPW : 1][X3 ? kKS : 1];                                     // X3: the low parts of the activations
    f16v acc[TPW][kMT];
    if (wave_on) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const long long row = row0 + 32 * t;
            const size_t xoff = (size_t)(row < P.rows ? row : P.rows - 1) * P.in_dim;
            const float *x = static_cast<const float *>(obs_) + xoff;
            const __bf16 *xb = static_cast<const __bf16 *>(obs_) + xoff;
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) {
                const int k0 = 16 * ks + 8 * h;
                if constexpr (IN_BF16) {                                    // the fragment as it lies in memory (in_dim % 8 == 0)
                    bf8 z;
#pragma unroll
                    for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.0f;
                    pk[t][ks] = k0 < P.in_dim ? *reinterpret_cast<const bf8 *>(xb + k0) : z;
                    if constexpr (X3) pl[t][ks] = z;                        // bf16 rows have no low part
                    continue;
                }
                float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
                if (k0 < P.in_dim) lo = *reinterpret_cast<const float4 *>(x + k0);          // in_dim is a multiple of 4
                if (k0 + 4 < P.in_dim) hi = *reinterpret_cast<const float4 *>(x + k0 + 4);
                const float xv[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const __bf16 hj = (__bf16)xv[j];
                    pk[t][ks][j] = hj;
                    if constexpr (X3) pl[t][ks][j] = (__bf16)(xv[j] - (float)hj);
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const float bv = P.b1[feat_of(mt, reg, h)];
#pragma unroll
                for (int t = 0; t < TPW; ++t) acc[t][mt][reg] = bv;
            }
    }
    commit(0);
    __syncthreads();

#pragma unroll
    for (int cc = 0; cc < kSteps; ++cc) {
        const int c = X3 ? cc >> 1 : cc;
        const bool low = X3 && (cc & 1);                                    // this step holds LOW weight parts
        const bool last_part = !X3 || low;                                  // the chunk's sums are complete after this step
        if (cc + 1 < kSteps) issue(cc + 1);
        const bf8 *wl = buf[cc & 1];
        if (wave_on) {
            if (c < 6) {
                const int half = c & 1;
#pragma unroll
                for (int ks = 0; ks < kKS; ++ks)
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        const bf8 a = wl[(m * kKS + ks) * 64 + lane];
#pragma unroll
                        for (int t = 0; t < TPW; ++t) {
                            acc[t][3 * half + m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pk[t][ks], acc[t][3 * half + m], 0, 0, 0);
                            if constexpr (X3) {
                                if (!low) acc[t][3 * half + m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pl[t][ks], acc[t][3 * half + m], 0, 0, 0);
                            }
                        }
                    }
                if (half == 1 && last_part) {                               // layer complete: leaky ReLU (slope 0.01, networks.py:40-42), pack
#pragma unroll
                    for (int t = 0; t < TPW; ++t) {
#pragma unroll
                        for (int kt = 0; kt < kMT; ++kt)
#pragma unroll
                            for (int s = 0; s < 2; ++s)
#pragma unroll
                                for (int j = 0; j < 8; ++j) {
                                    const float v = acc[t][kt][8 * s + j];
                                    const float r_ = fmaxf(v, 0.01f * v);
                                    const __bf16 hj = (__bf16)r_;
                                    pk[t][2 * kt + s][j] = hj;
                                    if constexpr (X3) pl[t][2 * kt + s][j] = (__bf16)(r_ - (float)hj);
                                }
#pragma unroll
                        for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
                            for (int reg = 0; reg < 16; ++reg) acc[t][mt][reg] = 0.0f;
                    }
                }
            } else {
                // output layer: one feature tile; action k is accumulator register k of the lower lane half (row = reg, h = 0)
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    f16v o = acc[t][0];                                     // zeros (X3: the high step's sums on the low step)
#pragma unroll
                    for (int ks = 0; ks < kKS; ++ks) {
                        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[ks * 64 + lane], pk[t][ks], o, 0, 0, 0);
                        if constexpr (X3) {
                            if (!low) o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[ks * 64 + lane], pl[t][ks], o, 0, 0, 0);
                        }
                    }
                    if constexpr (X3) { if (!low) { acc[t][0] = o; continue; } }
                    const long long row = row0 + 32 * t;
                    if (h == 0 && row < P.rows) {
                        float *y = act + (size_t)row * P.act_dim;
                        float z[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                        if (P.noise_scale > 0.0f) {
                            unsigned long long hk = pmix64(P.noise_key ^ (unsigned long long)row);
#pragma unroll
                            for (int k = 0; k < 4; k += 2) {
                                if (k < P.act_dim) {
                                    const float u1 = (float)((unsigned)(hk >> 40) + 1u) * 5.9604644775390625e-08f;      // (0, 1]
                                    const float u2 = (float)((unsigned)(hk >> 16) & 0xFFFFFFu) * 5.9604644775390625e-08f; // [0, 1)
                                    const float rad = sqrtf(-2.0f * logf(u1));
                                    float sn, cs;
                                    sincosf(6.283185307179586f * u2, &sn, &cs);
                                    z[k] = rad * cs; z[k + 1] = rad * sn;
                                    hk = pmix64(hk + 0x9E3779B97F4A7C15ull);
                                }
                            }
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (k < P.act_dim) {
                                float v = tanhf(o[k]);                      // networks.py:43 tanh output
                                if (P.noise_scale > 0.0f) v = fminf(fmaxf(v + P.noise_scale * z[k], -1.0f), 1.0f);
                                y[k] = v;
                            }
                    }
                }
            }
        }
        if (cc + 1 < kSteps) {
            commit(cc + 1);                                                 // the other buffer: last read in step cc - 1, behind the barrier
            __syncthreads();
        }
    }
}

uint16_t bf16_rne(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) return (uint16_t)((u >> 16) | 0x40);   // NaN
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

thread_local std::string g_policy_error;

}  // namespace

struct swarm_policy {
    int device, in_dim, hidden, act_dim;
    int precision;                 // 0 = bf16 (default), 1 = bf16x3
    void *d_blob;
    MlpParams p;
    bool smem_set;
};

extern "C" {

const char *swarm_policy_last_error(void) { return g_policy_error.c_str(); }

int swarm_policy_create(const float *w1, const float *b1, const float *w2, const float *b2, const float *w3, const float *b3,
                        const float *w4, const float *b4, int in_dim, int hidden, int act_dim, int device, swarm_policy_t **out)
{
    if (!out) return SWARM_POLICY_ERR_INVALID;
    *out = nullptr;
    if (!w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !w4 || !b4) { g_policy_error = "swarm_policy_create: null weight pointer"; return SWARM_POLICY_ERR_INVALID; }
    if (in_dim < 4 || in_dim > kKP || (in_dim & 3) || hidden < 1 || hidden >= kKP || act_dim < 1 || act_dim > 4) {
        g_policy_error = "swarm_policy_create: supported shapes are in_dim <= 192 (multiple of 4), hidden <= 191, act_dim <= 4";
        return SWARM_POLICY_ERR_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_policy_error = "swarm_policy_create: no HIP device (there is no CPU path)"; return SWARM_POLICY_ERR_HIP; }
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    if (device >= ndev || hipSetDevice(device) != hipSuccess) { g_policy_error = "swarm_policy_create: bad device"; return SWARM_POLICY_ERR_HIP; }

    // blob: [w1 | w2 | w3 | w4] fragments (8 bf16 per lane), then [b1 | b2 | b3 | b4] fp32
    const size_t frag = 64 * 8;                                   // bf16 elements per fragment
    const size_t n_hid = (size_t)kMT * kKS * frag, n_out = (size_t)kKS * frag;
    const size_t w_elems = 3 * n_hid + n_out;
    const size_t bytes = 2 * w_elems * 2 + (3 * kKP + 32) * 4;                // high parts, low parts, biases
    std::vector<unsigned char> blob(bytes, 0);
    uint16_t *wp = reinterpret_cast<uint16_t *>(blob.data());
    uint16_t *lp = wp + w_elems;                                              // low parts: bf16(w - bf16(w)), same layout
    float *bp = reinterpret_cast<float *>(blob.data() + 2 * w_elems * 2);
    auto bf16_val = [](uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; std::memcpy(&f, &u, 4); return f; };
    // k index of fragment element j of lane half h in k-step ks: natural for layer 1 (B comes from memory), permuted for
    // the layers whose B operand is the previous accumulator tile
    auto k_nat = [](int ks, int h, int j) { return 16 * ks + 8 * h + j; };
    auto k_acc = [](int ks, int h, int j) { return 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3); };
    auto pack = [&](uint16_t *dst, const float *w, int n_out_feat, int n_in, int tiles, bool natural) {
        uint16_t *dlo = lp + (dst - wp);
        for (int mt = 0; mt < tiles; ++mt)
            for (int ks = 0; ks < kKS; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int o = 32 * mt + (lane & 31), h = lane >> 5;
                        const int k = natural ? k_nat(ks, h, j) : k_acc(ks, h, j);
                        const float v = (o < n_out_feat && k < n_in) ? w[(size_t)o * n_in + k] : 0.0f;   // torch Linear: [out][in]
                        const size_t q = ((size_t)(mt * kKS + ks) * 64 + lane) * 8 + j;
                        dst[q] = bf16_rne(v);
                        dlo[q] = bf16_rne(v - bf16_val(dst[q]));
                    }
    };
    // The padded hidden feature `one` carries the constant 1.0 through the chain: layer 1 produces it (zero weights, bias 1),
    // the later layers copy it (1.0 on the diagonal) and read their biases from its weight column.
    const int one = hidden;
    auto with_bias = [&](const float *w, const float *b, int n_out_feat, bool keep_one) {
        std::vector<float> a((size_t)kKP * kKP, 0.0f);                      // [out][in], in padded to kKP
        for (int o = 0; o < n_out_feat; ++o) {
            for (int k = 0; k < hidden; ++k) a[(size_t)o * kKP + k] = w[(size_t)o * hidden + k];
            a[(size_t)o * kKP + one] = b[o];
        }
        if (keep_one) a[(size_t)one * kKP + one] = 1.0f;
        return a;
    };
    pack(wp, w1, hidden, in_dim, kMT, true);
    { const std::vector<float> a = with_bias(w2, b2, hidden, true); pack(wp + n_hid, a.data(), kKP, kKP, kMT, false); }
    { const std::vector<float> a = with_bias(w3, b3, hidden, true); pack(wp + 2 * n_hid, a.data(), kKP, kKP, kMT, false); }
    { const std::vector<float> a = with_bias(w4, b4, act_dim, false); pack(wp + 3 * n_hid, a.data(), kKP, kKP, 1, false); }
    for (int k = 0; k < hidden; ++k) bp[k] = b1[k];
    bp[one] = 1.0f;

    swarm_policy *p = new (std::nothrow) swarm_policy;
    if (!p) return SWARM_POLICY_ERR_INVALID;
    p->device = device; p->in_dim = in_dim; p->hidden = hidden; p->act_dim = act_dim; p->d_blob = nullptr; p->smem_set = false;
    p->precision = 0;
    if (hipMalloc(&p->d_blob, bytes) != hipSuccess || hipMemcpy(p->d_blob, blob.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) {
        if (p->d_blob) (void)hipFree(p->d_blob);
        delete p;
        g_policy_error = "swarm_policy_create: device allocation / upload failed";
        return SWARM_POLICY_ERR_HIP;
    }
    const bf8 *wd = reinterpret_cast<const bf8 *>(p->d_blob);
    const float *bd = reinterpret_cast<const float *>(static_cast<unsigned char *>(p->d_blob) + 2 * w_elems * 2);
    p->p.w1 = wd; p->p.w2 = wd + n_hid / 8; p->p.w3 = wd + 2 * n_hid / 8; p->p.w4 = wd + 3 * n_hid / 8;
    const bf8 *ld = wd + w_elems / 8;
    p->p.l1 = ld; p->p.l2 = ld + n_hid / 8; p->p.l3 = ld + 2 * n_hid / 8; p->p.l4 = ld + 3 * n_hid / 8;
    p->p.b1 = bd; p->p.b2 = bd + kKP; p->p.b3 = bd + 2 * kKP; p->p.b4 = bd + 3 * kKP;
    p->p.in_dim = in_dim; p->p.act_dim = act_dim; p->p.rows = 0; p->p.noise_scale = 0.0f; p->p.noise_key = 0;
    *out = p;
    return SWARM_POLICY_OK;
}

void swarm_policy_destroy(swarm_policy_t *p)
{
    if (!p) return;
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(p->device);
    (void)hipFree(p->d_blob);
    (void)hipSetDevice(prev);
    delete p;
}

static int policy_forward(swarm_policy_t *p, const void *obs, bool in_bf16, int64_t rows, float *act, void *stream,
                          float noise_scale = 0.0f, uint64_t seed = 0, uint64_t step = 0)
{
    if (!p || !obs || !act || rows < 0) { g_policy_error = "swarm_policy_forward: bad argument"; return SWARM_POLICY_ERR_INVALID; }
    if (in_bf16 && (p->in_dim & 7)) { g_policy_error = "swarm_policy_forward_bf16: in_dim must be a multiple of 8"; return SWARM_POLICY_ERR_INVALID; }
    if (rows == 0) return SWARM_POLICY_OK;
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(p->device) != hipSuccess) { g_policy_error = "swarm_policy_forward: hipSetDevice failed"; return SWARM_POLICY_ERR_HIP; }
    MlpParams q = p->p;
    q.rows = rows;
    q.noise_scale = noise_scale > 0.0f ? noise_scale : 0.0f;
    q.noise_key = pmix64(pmix64(seed + 0x9E3779B97F4A7C15ull) ^ (0xD1B54A32D192ED03ull * (step + 1)));
    // one row tile per wave; the two-tile instantiation (SWARM_POLICY_TPW=2, measurement knob) is slower: 98 vs 87 us on
    // 262144 bf16 rows, 118 vs 106 us on fp32 rows -- one wave per SIMD costs more than the halved LDS reads give back
    int tpw = 1;
    if (const char *ev = std::getenv("SWARM_POLICY_TPW")) tpw = ev[0] == '2' ? 2 : 1;
    if (p->precision == 1) tpw = 1;
    const int n_waves = waves_of(p->precision == 1);
    const long long per_block = (long long)n_waves * 32 * tpw;
    const unsigned grid = (unsigned)((rows + per_block - 1) / per_block);
    if (!p->smem_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_policy_mlp<false, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmemBytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_policy_mlp<true, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmemBytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_policy_mlp<false, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmemBytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_policy_mlp<true, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmemBytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_policy_mlp<false, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmemBytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_policy_mlp<true, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmemBytes);
        p->smem_set = true;
    }
    const dim3 g(grid), b(64 * n_waves);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (p->precision == 1) {
        if (in_bf16) hipLaunchKernelGGL((k_policy_mlp<true, 1, true>), g, b, kSmemBytes, st, q, obs, act);
        else hipLaunchKernelGGL((k_policy_mlp<false, 1, true>), g, b, kSmemBytes, st, q, obs, act);
    } else if (tpw == 2) {
        if (in_bf16) hipLaunchKernelGGL((k_policy_mlp<true, 2, false>), g, b, kSmemBytes, st, q, obs, act);
        else hipLaunchKernelGGL((k_policy_mlp<false, 2, false>), g, b, kSmemBytes, st, q, obs, act);
    } else {
        if (in_bf16) hipLaunchKernelGGL((k_policy_mlp<true, 1, false>), g, b, kSmemBytes, st, q, obs, act);
        else hipLaunchKernelGGL((k_policy_mlp<false, 1, false>), g, b, kSmemBytes, st, q, obs, act);
    }
    const hipError_t e = hipGetLastError();
    (void)hipSetDevice(prev);
    if (e != hipSuccess) { g_policy_error = std::string("swarm_policy_forward: ") + hipGetErrorString(e); return SWARM_POLICY_ERR_HIP; }
    return SWARM_POLICY_OK;
}

int swarm_policy_set_precision(swarm_policy_t *p, int precision)
{
    if (!p || (precision != SWARM_POLICY_BF16 && precision != SWARM_POLICY_BF16X3)) { g_policy_error = "swarm_policy_set_precision: bad argument"; return SWARM_POLICY_ERR_INVALID; }
    p->precision = precision;
    return SWARM_POLICY_OK;
}

int swarm_policy_forward(swarm_policy_t *p, const float *obs, int64_t rows, float *act, void *stream)
{
    return policy_forward(p, obs, false, rows, act, stream);
}

int swarm_policy_forward_bf16(swarm_policy_t *p, const void *obs_bf16, int64_t rows, float *act, void *stream)
{
    return policy_forward(p, obs_bf16, true, rows, act, stream);
}

int swarm_policy_forward_explore(swarm_policy_t *p, const void *obs, int obs_is_bf16, int64_t rows, float *act,
                                 float noise_scale, uint64_t seed, uint64_t step, void *stream)
{
    return policy_forward(p, obs, obs_is_bf16 != 0, rows, act, stream, noise_scale, seed, step);
}

}  // extern "C"
