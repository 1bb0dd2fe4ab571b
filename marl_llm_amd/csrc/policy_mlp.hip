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
constexpr int kWaves = 4;

struct MlpParams {
    const bf8 *w1, *w2, *w3, *w4;          // packed fragments: [feature tile][k-step][lane] x 8 bf16
    const float *b1, *b2, *b3, *b4;        // padded biases (kKP, kKP, kKP, 32)
    int in_dim, act_dim;
    long long rows;
};

__device__ __forceinline__ int feat_of(int mt, int reg, int h) { return 32 * mt + (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// A workgroup = kWaves wavefronts x TPW row tiles of 32 rows each.  Per layer the packed weight fragments (72 KB for a
// hidden layer) are copied once into LDS and read from there by every wave (ds_read_b128, lane-linear = conflict-free);
// each fragment read feeds TPW MFMAs.  Without this the kernel is bound by streaming 233 KB of weights per 32 rows
// through L2 -> L1 (measured 248 us for 262144 rows vs 136 us with the staging); weights per row drop by kWaves * TPW.
template <int TPW>
__device__ __forceinline__ void stage_weights(bf8 *wl, const bf8 *__restrict__ w, int n_frag)
{
    __syncthreads();                                                        // everyone is done with the previous layer's weights
    for (int q = threadIdx.x; q < n_frag * 64; q += 64 * kWaves) wl[q] = w[q];
    __syncthreads();
}

template <int TPW>
__device__ __forceinline__ void activate_pack(const f16v (&acc)[TPW][kMT], bf8 (&pk)[TPW][kMT][2])
{
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int kt = 0; kt < kMT; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = acc[t][kt][8 * s + j];
                    pk[t][kt][s][j] = (__bf16)fmaxf(v, 0.01f * v);       // leaky ReLU, slope 0.01 (networks.py:40-42)
                }
}

// hidden layer l+1 from the packed activations of layer l; weights in LDS
template <int TPW>
__device__ __forceinline__ void hidden_layer(const bf8 *wl, const float *__restrict__ bias, bf8 (&pk)[TPW][kMT][2], int lane, int h)
{
    f16v acc[TPW][kMT];
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const float bv = bias[feat_of(mt, reg, h)];
#pragma unroll
            for (int t = 0; t < TPW; ++t) acc[t][mt][reg] = bv;
        }
#pragma unroll
    for (int kt = 0; kt < kMT; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int mt = 0; mt < kMT; ++mt) {
                const bf8 a = wl[(mt * kKS + kt * 2 + s) * 64 + lane];
#pragma unroll
                for (int t = 0; t < TPW; ++t) acc[t][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pk[t][kt][s], acc[t][mt], 0, 0, 0);
            }
    activate_pack<TPW>(acc, pk);
}

template <int TPW>
__global__ void __launch_bounds__(64 * kWaves)
k_policy_mlp(const MlpParams P, const float *__restrict__ obs, float *__restrict__ act)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    bf8 *wl = reinterpret_cast<bf8 *>(smem_raw);                            // [feature tile][k-step][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long long row0 = ((long long)blockIdx.x * kWaves + wave) * (32 * TPW);      // first row of this wave
    const bool wave_on = row0 < P.rows;                                     // idle waves still take part in the barriers

    bf8 pk[TPW][kMT][2];
    stage_weights<TPW>(wl, P.w1, kMT * kKS);
    if (wave_on) {   // layer 1: B fragments straight from the observation rows (natural k order)
        f16v acc[TPW][kMT];
#pragma unroll
        for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const float bv = P.b1[feat_of(mt, reg, h)];
#pragma unroll
                for (int t = 0; t < TPW; ++t) acc[t][mt][reg] = bv;
            }
        const float *x[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const long long row = row0 + 32 * t + r;
            x[t] = obs + (size_t)(row < P.rows ? row : P.rows - 1) * P.in_dim;
        }
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) {
            const int k0 = 16 * ks + 8 * h;
            bf8 b[TPW];
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
                if (k0 < P.in_dim) lo = *reinterpret_cast<const float4 *>(x[t] + k0);          // in_dim is a multiple of 4
                if (k0 + 4 < P.in_dim) hi = *reinterpret_cast<const float4 *>(x[t] + k0 + 4);
                b[t][0] = (__bf16)lo.x; b[t][1] = (__bf16)lo.y; b[t][2] = (__bf16)lo.z; b[t][3] = (__bf16)lo.w;
                b[t][4] = (__bf16)hi.x; b[t][5] = (__bf16)hi.y; b[t][6] = (__bf16)hi.z; b[t][7] = (__bf16)hi.w;
            }
#pragma unroll
            for (int mt = 0; mt < kMT; ++mt) {
                const bf8 a = wl[(mt * kKS + ks) * 64 + lane];
#pragma unroll
                for (int t = 0; t < TPW; ++t) acc[t][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[t], acc[t][mt], 0, 0, 0);
            }
        }
        activate_pack<TPW>(acc, pk);
    }
    stage_weights<TPW>(wl, P.w2, kMT * kKS);
    if (wave_on) hidden_layer<TPW>(wl, P.b2, pk, lane, h);
    stage_weights<TPW>(wl, P.w3, kMT * kKS);
    if (wave_on) hidden_layer<TPW>(wl, P.b3, pk, lane, h);
    stage_weights<TPW>(wl, P.w4, kKS);
    if (!wave_on) return;
    // output layer: one feature tile; action k is accumulator register k of the lower lane half (row = reg for reg < 4, h = 0)
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        f16v o;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) o[reg] = P.b4[feat_of(0, reg, h) & 31];
#pragma unroll
        for (int kt = 0; kt < kMT; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[(kt * 2 + s) * 64 + lane], pk[t][kt][s], o, 0, 0, 0);
        const long long row = row0 + 32 * t + r;
        if (h == 0 && row < P.rows) {
            float *y = act + (size_t)row * P.act_dim;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < P.act_dim) y[k] = tanhf(o[k]);                      // networks.py:43 tanh output
        }
    }
}

constexpr int kSmemBytes = kMT * kKS * 64 * 16;                             // one hidden layer's fragments: 72 KB

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
    if (in_dim < 4 || in_dim > kKP || (in_dim & 3) || hidden < 1 || hidden > kKP || act_dim < 1 || act_dim > 4) {
        g_policy_error = "swarm_policy_create: supported shapes are in_dim <= 192 (multiple of 4), hidden <= 192, act_dim <= 4";
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
    const size_t bytes = w_elems * 2 + (3 * kKP + 32) * 4;
    std::vector<unsigned char> blob(bytes, 0);
    uint16_t *wp = reinterpret_cast<uint16_t *>(blob.data());
    float *bp = reinterpret_cast<float *>(blob.data() + w_elems * 2);
    // k index of fragment element j of lane half h in k-step ks: natural for layer 1 (B comes from memory), permuted for
    // the layers whose B operand is the previous accumulator tile
    auto k_nat = [](int ks, int h, int j) { return 16 * ks + 8 * h + j; };
    auto k_acc = [](int ks, int h, int j) { return 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3); };
    auto pack = [&](uint16_t *dst, const float *w, int n_out_feat, int n_in, int tiles, bool natural) {
        for (int mt = 0; mt < tiles; ++mt)
            for (int ks = 0; ks < kKS; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int o = 32 * mt + (lane & 31), h = lane >> 5;
                        const int k = natural ? k_nat(ks, h, j) : k_acc(ks, h, j);
                        const float v = (o < n_out_feat && k < n_in) ? w[(size_t)o * n_in + k] : 0.0f;   // torch Linear: [out][in]
                        dst[((size_t)(mt * kKS + ks) * 64 + lane) * 8 + j] = bf16_rne(v);
                    }
    };
    pack(wp, w1, hidden, in_dim, kMT, true);
    pack(wp + n_hid, w2, hidden, hidden, kMT, false);
    pack(wp + 2 * n_hid, w3, hidden, hidden, kMT, false);
    pack(wp + 3 * n_hid, w4, act_dim, hidden, 1, false);
    for (int k = 0; k < hidden; ++k) { bp[k] = b1[k]; bp[kKP + k] = b2[k]; bp[2 * kKP + k] = b3[k]; }
    for (int k = 0; k < act_dim; ++k) bp[3 * kKP + k] = b4[k];

    swarm_policy *p = new (std::nothrow) swarm_policy;
    if (!p) return SWARM_POLICY_ERR_INVALID;
    p->device = device; p->in_dim = in_dim; p->hidden = hidden; p->act_dim = act_dim; p->d_blob = nullptr; p->smem_set = false;
    if (hipMalloc(&p->d_blob, bytes) != hipSuccess || hipMemcpy(p->d_blob, blob.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) {
        if (p->d_blob) (void)hipFree(p->d_blob);
        delete p;
        g_policy_error = "swarm_policy_create: device allocation / upload failed";
        return SWARM_POLICY_ERR_HIP;
    }
    const bf8 *wd = reinterpret_cast<const bf8 *>(p->d_blob);
    const float *bd = reinterpret_cast<const float *>(static_cast<unsigned char *>(p->d_blob) + w_elems * 2);
    p->p.w1 = wd; p->p.w2 = wd + n_hid / 8; p->p.w3 = wd + 2 * n_hid / 8; p->p.w4 = wd + 3 * n_hid / 8;
    p->p.b1 = bd; p->p.b2 = bd + kKP; p->p.b3 = bd + 2 * kKP; p->p.b4 = bd + 3 * kKP;
    p->p.in_dim = in_dim; p->p.act_dim = act_dim; p->p.rows = 0;
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

int swarm_policy_forward(swarm_policy_t *p, const float *obs, int64_t rows, float *act, void *stream)
{
    if (!p || !obs || !act || rows < 0) { g_policy_error = "swarm_policy_forward: bad argument"; return SWARM_POLICY_ERR_INVALID; }
    if (rows == 0) return SWARM_POLICY_OK;
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(p->device) != hipSuccess) { g_policy_error = "swarm_policy_forward: hipSetDevice failed"; return SWARM_POLICY_ERR_HIP; }
    MlpParams q = p->p;
    q.rows = rows;
    // one 32-row tile per wave: 204 VGPRs and 72 KB of LDS -> two workgroups (8 waves) per CU.  Two tiles per wave (each
    // fragment read feeding two MFMAs) needs 400 VGPRs, i.e. one wave per SIMD, and measured slower (159 vs 136 us at
    // 262144 rows): the template parameter is kept for that experiment only.
    const long long per_block = (long long)kWaves * 32;
    const unsigned grid = (unsigned)((rows + per_block - 1) / per_block);
    if (!p->smem_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_policy_mlp<1>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmemBytes);
        p->smem_set = true;
    }
    hipLaunchKernelGGL(k_policy_mlp<1>, dim3(grid), dim3(64 * kWaves), kSmemBytes, static_cast<hipStream_t>(stream), q, obs, act);
    const hipError_t e = hipGetLastError();
    (void)hipSetDevice(prev);
    if (e != hipSuccess) { g_policy_error = std::string("swarm_policy_forward: ") + hipGetErrorString(e); return SWARM_POLICY_ERR_HIP; }
    return SWARM_POLICY_OK;
}

}  // extern "C"
