// swarm_env.hip -- MI355X (gfx950 / CDNA4) batched AssemblySwarm environment step + its C ABI.
//
// One fused kernel advances E independent environments by one AssemblySwarmEnv.step():
//   contact / wall forces -> prior policy -> semi-implicit Euler -> neighbour search ->
//   target-cell scan -> occupied-cell filter -> capped sensed list -> reward -> observation rows.
// Reference semantics being reproduced (never copied):
//   ENV = /root/reference/cus_gym/gym/envs/customized_envs/assembly.py
//   CPP = /root/reference/cus_gym/gym/envs/customized_envs/envs_cplus/src/AssemblyEnv.cpp
//
// Mapping: lane = agent.  NPAD (agents per env rounded up to a power of two, 8..256) is a template
// parameter; a 64-wide wavefront holds 64/NPAD whole environments when NPAD < 64, and an environment
// of 128/256 agents is a workgroup of 2/4 wavefronts.  Per-env data that every agent re-reads (target
// cells, agent positions/velocities) is staged once in LDS and read with wave-uniform addresses
// (LDS broadcast).  Per-cell "which agents are within r_avoid/2" masks come straight from wavefront
// ballots.  The observation block of an environment is streamed out with consecutive lanes writing
// consecutive addresses (the rows of one env are contiguous in HBM).
//
// Numerics: everything that decides an index, a flag, the state or the reward is IEEE double in the
// reference's operation order; this file MUST be compiled with -ffp-contract=off (the reference is
// built for baseline x86-64, no FMA).  Threshold tests of the form sqrt(d2) < t are evaluated as
// d2 < cut(t) with cut(t) = the smallest double whose correctly rounded sqrt is >= t, computed on the
// host; sqrt is monotonic, so the two tests are equivalent bit for bit and the scans need no sqrt.
//
// There is no CPU path in this file.

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "swarm_env.h"

namespace {

constexpr int kTopoMax = 6;
typedef unsigned long long u64;

struct KP {
    int n_env, n_a, ng_max, ngw, topo, g_max, occ_max, obs_dim;
    int with_self, periodic, boundary, with_prior, export_idx;
    int cxy_stride;            // double2 elements per env in LDS
    int g_stride;              // int16 elements per agent row in LDS
    int off_cxy, off_sp, off_cmask, off_sbits, off_obits, off_sidx, off_snei, off_sncf, smem_bytes;
    double c_sen, c_near, c_occ, c_avoid, c_ball;     // squared-distance cut-offs
    double d_sen, r_avoid, size_a, size2, k_ball, k_wall, c_wall, vel_max, dt;
    double bx0, by1, bx2, by3, w_half, h_half;
    double *p, *dp;
    int *nei, *near_cell, *in_flag;
    const double *cells;
    const int *n_g;
    const double *c_in;
    int *exp_sensed, *exp_occ;
};

template <typename T> struct Pair;
template <> struct Pair<float>  { typedef float2 type; };
template <> struct Pair<double> { typedef double2 type; };

__device__ __forceinline__ void wrap_rel(double &x, double &y, double wh, double hh)
{   // CPP:700-715
    if (x < -wh) x += 2 * wh; else if (x > wh) x -= 2 * wh;
    if (y < -hh) y += 2 * hh; else if (y > hh) y -= 2 * hh;
}

__device__ __forceinline__ double clamp_ref(double v, double lo, double hi)
{   // CPP:11-14  std::max(lo, std::min(v, hi))
    double m = (hi < v) ? hi : v;
    return (lo < m) ? m : lo;
}

template <int NPAD, typename OT, bool DO_STEP>
__global__ void __launch_bounds__((NPAD < 64 ? 64 : NPAD))
k_env(const KP P, const void *__restrict__ action, const int act_f64, OT *__restrict__ obs,
      float *__restrict__ reward, uint8_t *__restrict__ done, OT *__restrict__ a_prior)
{
    constexpr int T = NPAD < 64 ? 64 : NPAD;
    constexpr int EPB = NPAD < 64 ? 64 / NPAD : 1;
    constexpr int NW = T / 64;
    typedef typename Pair<OT>::type OT2;

    extern __shared__ __align__(16) unsigned char smem[];
    double2 *cxy = reinterpret_cast<double2 *>(smem + P.off_cxy);
    double *sp = reinterpret_cast<double *>(smem + P.off_sp);          // [4][T]: px, py, vx, vy
    u64 *cmask = reinterpret_cast<u64 *>(smem + P.off_cmask);          // [cell][NW]
    unsigned *sbits = reinterpret_cast<unsigned *>(smem + P.off_sbits); // [word][T]
    unsigned *obits = reinterpret_cast<unsigned *>(smem + P.off_obits); // [word][T] (export only)
    short *sidx = reinterpret_cast<short *>(smem + P.off_sidx);        // [T][g_stride]
    short *snei = reinterpret_cast<short *>(smem + P.off_snei);        // [T][kTopoMax]
    int *sncf = reinterpret_cast<int *>(smem + P.off_sncf);            // [T]: nearest cell | in_flag<<30

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int el = NPAD < 64 ? tid / NPAD : 0;
    const int i = NPAD < 64 ? tid % NPAD : tid;
    const int e = blockIdx.x * EPB + el;
    const int n_a = P.n_a;
    const bool act = (e < P.n_env) && (i < n_a);
    const int es = e < P.n_env ? e : P.n_env - 1;
    const int ng = P.n_g[es];
    int ngb = ng;
    if (EPB > 1) {
        for (int k = 0; k < EPB; ++k) {
            int ek = blockIdx.x * EPB + k;
            int v = P.n_g[ek < P.n_env ? ek : P.n_env - 1];
            ngb = v > ngb ? v : ngb;
        }
    }
    const int ngwb = (ngb + 31) >> 5;
    double2 *cxy_e = cxy + (size_t)el * P.cxy_stride;

    // ---- stage this env's target cells (ENV: grid_center (2, n_g)) in LDS as (x, y) pairs
    {
        const double *gx = P.cells + (size_t)es * 2 * P.ng_max;
        const double *gy = gx + P.ng_max;
        for (int c = i; c < ngwb * 32; c += NPAD) {
            double2 g;
            g.x = c < ng ? gx[c] : 0.0;
            g.y = c < ng ? gy[c] : 0.0;
            cxy_e[c] = g;
        }
    }
    // ---- state
    double px = 0, py = 0, vx = 0, vy = 0;
    const size_t sbase = (size_t)es * 2 * n_a;
    if (act) {
        px = P.p[sbase + i]; py = P.p[sbase + n_a + i];
        vx = P.dp[sbase + i]; vy = P.dp[sbase + n_a + i];
    }
    sp[tid] = px; sp[T + tid] = py; sp[2 * T + tid] = vx; sp[3 * T + tid] = vy;
    __syncthreads();

    if (DO_STEP) {
        // ---- ball-to-ball contact spring: ENV:442-457 (_get_dist_b2b) + CPP:735-815 (_sf_b2b_all).
        // Entry (i,k) = collide * d_edge * k_ball * (-(delta/d_center)), delta = p_k - p_i (wrapped when
        // periodic), d_center unwrapped for every pair the reference evaluates (its numpy wrap only touches
        // agent 0's row, which the i>j loop never reads).  Summed over k in index order.
        double sfx = 0.0, sfy = 0.0;
        for (int k = 0; k < n_a; ++k) {
            const int tk = el * NPAD + k;
            const double dx = sp[tk] - px, dy = sp[T + tk] - py;
            const double d2 = dx * dx + dy * dy;
            if (act && k != i && d2 < P.c_ball) {
                const double dc = sqrt(d2);
                const double de = fabs(dc - P.size2);
                double wx = dx, wy = dy;
                if (P.periodic) wrap_rel(wx, wy, P.w_half, P.h_half);
                const double ux = wx / dc, uy = wy / dc;
                sfx += 1.0 * de * P.k_ball * (-ux);
                sfy += 1.0 * de * P.k_ball * (-uy);
            }
        }
        double ax = 0.0, ay = 0.0;
        if (act) {
            const size_t ab = ((size_t)e * n_a + i) * 2;
            if (act_f64) { ax = ((const double *)action)[ab]; ay = ((const double *)action)[ab + 1]; }
            else { ax = (double)((const float *)action)[ab]; ay = (double)((const float *)action)[ab + 1]; }
        }
        double Fx = 1 * ax + sfx, Fy = 1 * ay + sfy;                       // ENV:638,640
        if (P.boundary) {                                                   // CPP:817-855, ENV:515-518
            const double d0 = px - P.size_a - P.bx0;
            const double d1 = P.by1 - (py + P.size_a);
            const double d2 = P.bx2 - (px + P.size_a);
            const double d3 = py - P.size_a - P.by3;
            const double a0 = d0 < 0 ? fabs(d0) : 0.0, a1 = d1 < 0 ? fabs(d1) : 0.0;
            const double a2 = d2 < 0 ? fabs(d2) : 0.0, a3 = d3 < 0 ? fabs(d3) : 0.0;
            const double sx = (a0 - a2) * P.k_wall, sy = (a3 - a1) * P.k_wall;
            const double v0 = d0 < 0 ? vx : 0.0, v2 = d2 < 0 ? vx : 0.0;
            const double v3 = d3 < 0 ? vy : 0.0, v1 = d1 < 0 ? vy : 0.0;
            const double gx = (-v0 - v2) * P.c_wall, gy = (-v3 - v1) * P.c_wall;
            Fx = Fx + sx + gx;
            Fy = Fy + sy + gy;
        }
        // ---- prior policy on the PRE-integration state with the previous neighbour list:
        // CPP:1061-1196 via ENV:605-624.  The nearest cell / in-shape flag of the pre-integration
        // position are the ones the previous observation pass cached.
        if (P.with_prior && act && a_prior != nullptr) {
            const int ncell = P.near_cell[(size_t)e * n_a + i];
            const int inf = P.in_flag[(size_t)e * n_a + i];
            double tx, ty;
            if (inf) { tx = px - px; ty = py - py; }
            else { const double2 g = cxy_e[ncell]; tx = g.x - px; ty = g.y - py; }
            double qx = 0.0, qy = 0.0;
            const double dt_ = sqrt(tx * tx + ty * ty);
            if (dt_ > 0) { qx += 2.0 * tx / dt_; qy += 2.0 * ty / dt_; }
            double avx = 0.0, avy = 0.0; int cnt = 0;
            for (int k = 0; k < P.topo; ++k) {
                const int j = P.nei[((size_t)e * n_a + i) * P.topo + k];
                if (j < 0) continue;
                const int tj = el * NPAD + j;
                const double x = px - sp[tj], y = py - sp[T + tj];
                const double d = sqrt(x * x + y * y);
                if (d > 0 && d < P.r_avoid) {
                    const double ux = x / d, uy = y / d;
                    const double factor = 3.0 * (P.r_avoid / d - 1.0);
                    qx += factor * ux; qy += factor * uy;
                }
                avx += sp[2 * T + tj]; avy += sp[3 * T + tj]; ++cnt;
            }
            if (cnt > 0) {
                avx /= cnt; avy /= cnt;
                qx += 2.0 * (avx - vx); qy += 2.0 * (avy - vy);
            }
            OT2 o; o.x = (OT)clamp_ref(qx, -1.0, 1.0); o.y = (OT)clamp_ref(qy, -1.0, 1.0);
            reinterpret_cast<OT2 *>(a_prior)[(size_t)e * n_a + i] = o;
        }
        // ---- integration, ENV:643-652
        double nvx = vx + (Fx / 1.0) * P.dt, nvy = vy + (Fy / 1.0) * P.dt;
        nvx = nvx < -P.vel_max ? -P.vel_max : (nvx > P.vel_max ? P.vel_max : nvx);
        nvy = nvy < -P.vel_max ? -P.vel_max : (nvy > P.vel_max ? P.vel_max : nvy);
        double npx = px + nvx * P.dt, npy = py + nvy * P.dt;
        if (P.periodic) {                                                    // ENV:773-776
            if (npx < P.bx0) npx += 2 * P.w_half;
            if (npx > P.bx2) npx -= 2 * P.w_half;
            if (npy < P.by3) npy += 2 * P.h_half;
            if (npy > P.by1) npy -= 2 * P.h_half;
        }
        __syncthreads();                 // every lane is done with the old positions in LDS
        px = npx; py = npy; vx = nvx; vy = nvy;
        sp[tid] = px; sp[T + tid] = py; sp[2 * T + tid] = vx; sp[3 * T + tid] = vy;
        if (act) {
            P.p[sbase + i] = px; P.p[sbase + n_a + i] = py;
            P.dp[sbase + i] = vx; P.dp[sbase + n_a + i] = vy;
        }
        __syncthreads();
    }

    // ---- neighbour search, CPP:77-100 + _get_focused CPP:628-698: the topo nearest agents with
    // norm < d_sen (self removed), ascending.  Also the "nearby" agent mask of the occupied-cell filter
    // (CPP:152-164: un-wrapped distance < d_sen + r_avoid/2, self included).
    double nd[kTopoMax]; int nj[kTopoMax];
#pragma unroll
    for (int k = 0; k < kTopoMax; ++k) { nd[k] = INFINITY; nj[k] = -1; }
    u64 nearby[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) nearby[w] = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const int jn = NPAD < 64 ? NPAD : 64;
        for (int jj = 0; jj < jn; ++jj) {
            const int j = w * 64 + jj;
            if (j >= n_a) break;
            const int tj = el * NPAD + j;
            double rx = sp[tj] - px, ry = sp[T + tj] - py;
            const double d2u = rx * rx + ry * ry;
            if (act && d2u < P.c_near) nearby[w] |= 1ull << (tj & 63);
            double d2 = d2u;
            if (P.periodic) { wrap_rel(rx, ry, P.w_half, P.h_half); d2 = rx * rx + ry * ry; }
            if (act && j != i && d2 < P.c_sen) {
                double cd = d2; int cj = j;
#pragma unroll
                for (int k = 0; k < kTopoMax; ++k) {
                    const bool s = cd < nd[k];
                    const double td = nd[k]; const int tjj = nj[k];
                    nd[k] = s ? cd : td; nj[k] = s ? cj : tjj;
                    cd = s ? td : cd;    cj = s ? tjj : cj;
                }
            }
        }
    }
    bool collision = false;                                   // CPP:459-491 (on the NEW neighbour list)
#pragma unroll
    for (int k = 0; k < kTopoMax; ++k) {
        const bool used = k < P.topo && nj[k] >= 0;
        snei[tid * kTopoMax + k] = (short)(used ? nj[k] : -1);
        if (used && nd[k] < P.c_avoid) collision = true;
        if (act && k < P.topo) P.nei[((size_t)e * n_a + i) * P.topo + k] = used ? nj[k] : -1;
    }

    // ---- target-cell scan, _get_target_grid_state CPP:858-908: first-minimum nearest cell, sensed-cell
    // bits (d < d_sen), and per cell the ballot of agents with d <= r_avoid/2 (CPP:183-186 inverted).
    double best = INFINITY; int bc = 0;
    {
        u64 mym = 0;
        for (int w = 0; w < ngwb; ++w) {
            unsigned word = 0;
            for (int b = 0; b < 32; ++b) {
                const int c = w * 32 + b;
                const double2 g = cxy_e[c];
                const double rx = g.x - px, ry = g.y - py;
                const double d2 = rx * rx + ry * ry;
                const bool valid = act && c < ng;
                if (valid && d2 < best) { best = d2; bc = c; }
                if (valid && d2 < P.c_sen) word |= 1u << b;
                const u64 m = __ballot(valid && d2 < P.c_occ);
                if (lane == (c & 63)) mym = m;
            }
            sbits[w * T + tid] = word;
            if ((w & 1) || w == ngwb - 1) cmask[(size_t)((w >> 1) * 64 + lane) * NW + wave] = mym;
        }
    }
    const bool in_shape = act && ng > 0 && best < P.c_in[es];          // CPP:889
    sncf[tid] = bc | (in_shape ? (1 << 30) : 0);
    if (act) {
        P.near_cell[(size_t)e * n_a + i] = bc;
        P.in_flag[(size_t)e * n_a + i] = in_shape ? 1 : 0;
    }
    __syncthreads();

    // ---- occupied-cell filter, CPP:144-216: a sensed cell is occupied iff some nearby agent is within
    // r_avoid/2 of it; only agents inside the shape filter (CPP:150).
    int n_kept = 0, n_occ = 0;
    for (int w = 0; w < ngwb; ++w) {
        const unsigned word = sbits[w * T + tid];
        unsigned kw = word;
        if (in_shape) {
            unsigned it = word;
            while (it) {
                const int b = __ffs(it) - 1;
                it &= it - 1;
                const int c = w * 32 + b;
                bool occ = false;
#pragma unroll
                for (int q = 0; q < NW; ++q) occ = occ || ((cmask[(size_t)c * NW + q] & nearby[q]) != 0);
                if (occ) kw &= ~(1u << b);
            }
            sbits[w * T + tid] = kw;
        }
        if (P.export_idx) obits[w * T + tid] = word & ~kw;
        n_kept += __popc(kw);
        n_occ += __popc(word & ~kw);
    }

    // ---- capped sensed list (CPP:236-271) into LDS, and the exploration reward over it (CPP:494-551)
    {
        const int G = P.g_max;
        const bool sub = n_kept > G;
        const double step = sub ? (double)(n_kept - 1) / (G - 1) : 0.0;
        int s = 0, k = 0, target = 0;
        double num0 = 0.0, num1 = 0.0, den = 0.0;
        short *row = sidx + (size_t)tid * P.g_stride;
        for (int w = 0; w < ngwb; ++w) {
            unsigned it = sbits[w * T + tid];
            while (it) {
                const int b = __ffs(it) - 1;
                it &= it - 1;
                const int c = w * 32 + b;
                const bool sel = sub ? (k == target) : true;
                if (sel && s < G) {
                    row[s] = (short)c;
                    if (in_shape) {
                        const double2 g = cxy_e[c];
                        const double x = g.x - px, y = g.y - py;
                        const double z = sqrt(x * x + y * y);
                        double psi;                                   // _rho_cos_dec(z, 0, d_sen), CPP:1012-1020
                        if (z < 0.0 * P.d_sen) psi = 1.0;
                        else if (z < P.d_sen) psi = (1.0 / 2.0) * (1.0 + cos(M_PI * (z / P.d_sen - 0.0) / (1.0 - 0.0)));
                        else psi = 0.0;
                        num0 += psi * x; num1 += psi * y; den += psi;
                    }
                    ++s;
                    if (sub) target = (int)round(s * step);
                }
                ++k;
            }
        }
        const int n_sel = s;
        for (; s < G; ++s) row[s] = -1;
        bool uniform = false;
        if (in_shape && n_sel > 0) {
            if (den == 0) den = 1E-8;
            const double v0 = 1.0 * num0 / den, v1 = 1.0 * num1 / den;
            uniform = sqrt(v0 * v0 + v1 * v1) < 0.05;
        }
        if (act) {
            if (reward != nullptr) reward[(size_t)e * n_a + i] = (in_shape && !collision && uniform) ? 1.0f : 0.0f;   // CPP:554-556
            if (done != nullptr) done[(size_t)e * n_a + i] = 0;                                                         // ENV:480-482
        }
        if (P.export_idx && act) {
            int *es_ = P.exp_sensed + ((size_t)e * n_a + i) * G;
            for (int q = 0; q < G; ++q) es_[q] = row[q];
            // occupied list with its own cap, CPP:217-233
            const int O = P.occ_max;
            int *eo = P.exp_occ + ((size_t)e * n_a + i) * O;
            const bool osub = n_occ > O;
            const double ostep = osub ? (double)(n_occ - 1) / (O - 1) : 0.0;
            int os = 0, ok = 0, otarget = 0;
            for (int w = 0; w < ngwb; ++w) {
                unsigned it = obits[w * T + tid];
                while (it) {
                    const int b = __ffs(it) - 1;
                    it &= it - 1;
                    const bool sel = osub ? (ok == otarget) : true;
                    if (sel && os < O) { eo[os++] = w * 32 + b; if (osub) otarget = (int)round(os * ostep); }
                    ++ok;
                }
            }
            for (; os < O; ++os) eo[os] = -1;
        }
    }
    __syncthreads();

    // ---- observation rows, CPP:102-137,274-306: streamed out as (value, value) pairs, consecutive lanes
    // -> consecutive addresses; the rows of this block's environments are contiguous in HBM.
    if (obs != nullptr) {
        const int PPR = P.obs_dim >> 1;                  // pairs per row
        const int base_pairs = 2 * (P.with_self + P.topo);
        const int envs_here = (P.n_env - blockIdx.x * EPB) < EPB ? (P.n_env - blockIdx.x * EPB) : EPB;
        const int total = envs_here * n_a * PPR;
        OT2 *out = reinterpret_cast<OT2 *>(obs) + (size_t)blockIdx.x * EPB * n_a * PPR;
        const int dr = T / PPR, dq = T % PPR;
        int r = tid / PPR, q = tid % PPR;
        for (int L = tid; L < total; L += T) {
            const int elr = EPB > 1 ? r / n_a : 0;
            const int ir = r - elr * n_a;
            const int tr = elr * NPAD + ir;
            const double qx = sp[tr], qy = sp[T + tr], ux = sp[2 * T + tr], uy = sp[3 * T + tr];
            const int ncf = sncf[tr];
            const double2 *cx = cxy + (size_t)elr * P.cxy_stride;
            double a = 0.0, b = 0.0;
            if (q < base_pairs) {
                const int blk = q >> 1, half = q & 1;
                if (P.with_self && blk == 0) {                          // CPP:103-113
                    a = half ? ux : qx; b = half ? uy : qy;
                } else {
                    const int j = snei[tr * kTopoMax + (blk - P.with_self)];
                    if (j >= 0) {
                        const int tj = elr * NPAD + j;
                        if (half) { a = sp[2 * T + tj] - ux; b = sp[3 * T + tj] - uy; }     // CPP:80-81
                        else {
                            a = sp[tj] - qx; b = sp[T + tj] - qy;                           // CPP:79
                            if (P.periodic) wrap_rel(a, b, P.w_half, P.h_half);
                        }
                    }
                }
            } else if (q == base_pairs) {                               // CPP:136
                if (ncf >> 30) { a = qx - qx; b = qy - qy; }
                else { const double2 g = cx[ncf & 0xFFFF]; a = g.x - qx; b = g.y - qy; }
            } else if (q == base_pairs + 1) {                           // CPP:137
                if (ncf >> 30) { a = ux - ux; b = uy - uy; }
                else { a = 0.0 - ux; b = 0.0 - uy; }
            } else {                                                    // CPP:274-291
                const int c = sidx[(size_t)tr * P.g_stride + (q - base_pairs - 2)];
                if (c >= 0) { const double2 g = cx[c]; a = g.x - qx; b = g.y - qy; }
            }
            OT2 o; o.x = (OT)a; o.y = (OT)b;
            out[L] = o;
            q += dq; r += dr;
            if (q >= PPR) { q -= PPR; ++r; }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------

thread_local std::string g_create_error;

// smallest double x with sqrt(x) >= t (IEEE sqrt is correctly rounded and monotonic), so that
// sqrt(d2) < t  <=>  d2 < x  for every d2 >= 0.
double cut_lt(double t)
{
    if (!(t > 0)) return 0.0;
    double x = t * t;
    while (x > 0 && std::sqrt(x) >= t) x = std::nextafter(x, 0.0);
    while (std::sqrt(x) < t) x = std::nextafter(x, INFINITY);
    return x;
}
// sqrt(d2) <= t  <=>  d2 < cut_le(t)
double cut_le(double t) { return cut_lt(std::nextafter(t, INFINITY)); }

int npad_for(int n)
{
    int v = 8;
    while (v < n) v <<= 1;
    return v;
}

}  // namespace

struct swarm_env {
    swarm_config_t cfg;
    KP kp;
    int device;
    int npad;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    bool have_cells, have_state, observed;
    std::vector<char> cells_set;
    std::string err;
    // device buffers
    double *d_p, *d_dp, *d_cells, *d_cin;
    int *d_nei, *d_near, *d_inflag, *d_ng, *d_exp_sensed, *d_exp_occ;
};

namespace {

int fail(swarm_env *h, int code, const std::string &msg)
{
    if (h) h->err = msg; else g_create_error = msg;
    return code;
}

#define HIP_TRY(h, call)                                                                       \
    do {                                                                                       \
        hipError_t e__ = (call);                                                               \
        if (e__ != hipSuccess)                                                                 \
            return fail(h, SWARM_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

struct DeviceGuard {
    int prev;
    bool ok;
    explicit DeviceGuard(int dev) : prev(-1), ok(false)
    {
        if (hipGetDevice(&prev) != hipSuccess) return;
        ok = (prev == dev) || hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

void layout(KP &k, int npad)
{
    const int T = npad < 64 ? 64 : npad;
    const int EPB = npad < 64 ? 64 / npad : 1;
    const int NW = T / 64;
    k.ngw = (k.ng_max + 31) / 32;
    k.cxy_stride = k.ngw * 32 + 1;            // +1 pair: envs of one wave start on different LDS banks
    int half = (k.g_max + 1) / 2;
    if ((half & 1) == 0) ++half;              // odd dword stride: lane-per-row int16 writes spread over banks
    k.g_stride = 2 * half;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 15) & ~size_t(15); return (int)o; };
    k.off_cxy = take((size_t)EPB * k.cxy_stride * 16);
    k.off_sp = take((size_t)4 * T * 8);
    k.off_cmask = take((size_t)((k.ngw + 1) / 2) * 64 * NW * 8);
    k.off_sbits = take((size_t)k.ngw * T * 4);
    k.off_obits = take((size_t)k.ngw * T * 4);
    k.off_sidx = take((size_t)T * k.g_stride * 2);
    k.off_snei = take((size_t)T * kTopoMax * 2);
    k.off_sncf = take((size_t)T * 4);
    k.smem_bytes = (int)off;
}

template <int NPAD, typename OT, bool DO_STEP>
int launch_t(swarm_env *h, const void *action, int act_f64, void *obs, float *reward, uint8_t *done, void *a_prior)
{
    constexpr int T = NPAD < 64 ? 64 : NPAD;
    constexpr int EPB = NPAD < 64 ? 64 / NPAD : 1;
    auto kern = k_env<NPAD, OT, DO_STEP>;
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   h->kp.smem_bytes));
    const int grid = (h->cfg.n_env + EPB - 1) / EPB;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(T), h->kp.smem_bytes, h->stream, h->kp, action, act_f64,
                       static_cast<OT *>(obs), reward, done, static_cast<OT *>(a_prior));
    HIP_TRY(h, hipGetLastError());
    return SWARM_OK;
}

template <int NPAD>
int launch_n(swarm_env *h, bool do_step, const void *action, int act_f64, void *obs, float *reward, uint8_t *done,
             void *a_prior)
{
    const bool f64 = h->cfg.obs_dtype == SWARM_F64;
    if (do_step) {
        return f64 ? launch_t<NPAD, double, true>(h, action, act_f64, obs, reward, done, a_prior)
                   : launch_t<NPAD, float, true>(h, action, act_f64, obs, reward, done, a_prior);
    }
    return f64 ? launch_t<NPAD, double, false>(h, action, act_f64, obs, reward, done, a_prior)
               : launch_t<NPAD, float, false>(h, action, act_f64, obs, reward, done, a_prior);
}

int launch(swarm_env *h, bool do_step, const void *action, int act_f64, void *obs, float *reward, uint8_t *done,
           void *a_prior)
{
    switch (h->npad) {
    case 8: return launch_n<8>(h, do_step, action, act_f64, obs, reward, done, a_prior);
    case 16: return launch_n<16>(h, do_step, action, act_f64, obs, reward, done, a_prior);
    case 32: return launch_n<32>(h, do_step, action, act_f64, obs, reward, done, a_prior);
    case 64: return launch_n<64>(h, do_step, action, act_f64, obs, reward, done, a_prior);
    case 128: return launch_n<128>(h, do_step, action, act_f64, obs, reward, done, a_prior);
    case 256: return launch_n<256>(h, do_step, action, act_f64, obs, reward, done, a_prior);
    }
    return fail(h, SWARM_ERR_INVALID, "unsupported agent count");
}

}  // namespace

extern "C" {

int swarm_abi_version(void) { return SWARM_ABI_VERSION; }

void swarm_default_config(swarm_config_t *c)
{
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->n_env = 1; c->n_agents = 30; c->n_cells_max = 576;
    c->topo_nei_max = 6; c->num_obs_grid_max = 80; c->num_occupied_grid_max = 200;
    c->is_boundary = 1; c->with_self_state = 1; c->with_prior = 1;
    c->obs_dtype = SWARM_F32; c->device = -1;
    c->d_sen = 0.4; c->r_avoid = 0.15; c->size_a = 0.035;
    c->k_ball = 30; c->k_wall = 100; c->c_wall = 5; c->vel_max = 0.8; c->dt = 0.1;
    c->boundary[0] = -2.4; c->boundary[1] = 2.4; c->boundary[2] = 2.4; c->boundary[3] = -2.4;
}

const char *swarm_last_error(const swarm_env_t *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int swarm_create(const swarm_config_t *cfg, swarm_env_t **out)
{
    if (!cfg || !out) return fail(nullptr, SWARM_ERR_INVALID, "swarm_create: null argument");
    *out = nullptr;
    if (cfg->n_env < 1) return fail(nullptr, SWARM_ERR_INVALID, "n_env must be >= 1");
    if (cfg->n_agents < 1 || cfg->n_agents > 256) return fail(nullptr, SWARM_ERR_INVALID, "n_agents must be in [1, 256]");
    if (cfg->n_cells_max < 1 || cfg->n_cells_max > 32767) return fail(nullptr, SWARM_ERR_INVALID, "n_cells_max must be in [1, 32767]");
    if (cfg->topo_nei_max < 1 || cfg->topo_nei_max > kTopoMax) return fail(nullptr, SWARM_ERR_INVALID, "topo_nei_max must be in [1, 6]");
    if (cfg->num_obs_grid_max < 2 || cfg->num_obs_grid_max > 4096) return fail(nullptr, SWARM_ERR_INVALID, "num_obs_grid_max must be in [2, 4096]");
    if (cfg->num_occupied_grid_max < 2) return fail(nullptr, SWARM_ERR_INVALID, "num_occupied_grid_max must be >= 2");
    if (cfg->obs_dtype != SWARM_F32 && cfg->obs_dtype != SWARM_F64) return fail(nullptr, SWARM_ERR_INVALID, "obs_dtype must be SWARM_F32 or SWARM_F64");
    if (!(cfg->d_sen > 0) || !(cfg->r_avoid > 0) || !(cfg->dt > 0)) return fail(nullptr, SWARM_ERR_INVALID, "d_sen, r_avoid, dt must be positive");

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1)
        return fail(nullptr, SWARM_ERR_HIP, std::string("no HIP device available (") + hipGetErrorString(e) + "); this library has no CPU path");
    int dev = cfg->device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= ndev) return fail(nullptr, SWARM_ERR_INVALID, "device ordinal out of range");

    swarm_env *h = new (std::nothrow) swarm_env();
    if (!h) return fail(nullptr, SWARM_ERR_INVALID, "out of host memory");
    h->cfg = *cfg; h->device = dev; h->stream = nullptr; h->ev0 = h->ev1 = nullptr;
    h->have_cells = h->have_state = h->observed = false;
    h->d_p = h->d_dp = h->d_cells = h->d_cin = nullptr;
    h->d_nei = h->d_near = h->d_inflag = h->d_ng = h->d_exp_sensed = h->d_exp_occ = nullptr;
    h->cells_set.assign((size_t)cfg->n_env, 0);
    h->npad = npad_for(cfg->n_agents);

    KP &k = h->kp;
    std::memset(&k, 0, sizeof(k));
    k.n_env = cfg->n_env; k.n_a = cfg->n_agents; k.ng_max = cfg->n_cells_max;
    k.topo = cfg->topo_nei_max; k.g_max = cfg->num_obs_grid_max; k.occ_max = cfg->num_occupied_grid_max;
    k.with_self = cfg->with_self_state ? 1 : 0;
    k.obs_dim = 2 * 2 * (k.topo + 1 + k.with_self) + 2 * k.g_max;               // ENV:801
    k.boundary = cfg->is_boundary ? 1 : 0; k.periodic = cfg->is_boundary ? 0 : 1;  // ENV:99-103
    k.with_prior = cfg->with_prior ? 1 : 0;
    k.d_sen = cfg->d_sen; k.r_avoid = cfg->r_avoid; k.size_a = cfg->size_a;
    k.size2 = cfg->size_a + cfg->size_a;                                         // ENV:785-786
    k.k_ball = cfg->k_ball; k.k_wall = cfg->k_wall; k.c_wall = cfg->c_wall; k.vel_max = cfg->vel_max; k.dt = cfg->dt;
    k.bx0 = cfg->boundary[0]; k.by1 = cfg->boundary[1]; k.bx2 = cfg->boundary[2]; k.by3 = cfg->boundary[3];
    k.w_half = (k.bx2 - k.bx0) / 2.0; k.h_half = (k.by1 - k.by3) / 2.0;         // CPP:70-71
    k.c_sen = cut_lt(k.d_sen);                            // norm < d_sen              CPP:658,902
    k.c_near = cut_lt(k.d_sen + k.r_avoid / 2.0);         // norm < d_sen + r_avoid/2  CPP:161
    k.c_occ = cut_le(k.r_avoid / 2.0);                    // !(norm > r_avoid/2)       CPP:185
    k.c_avoid = cut_lt(k.r_avoid);                        // r_avoid > norm            CPP:482
    k.c_ball = cut_lt(k.size2);                           // d_center - sizes < 0      ENV:450-451
    layout(k, h->npad);

    DeviceGuard g(dev);
    if (!g.ok) { delete h; return fail(nullptr, SWARM_ERR_HIP, "hipSetDevice failed"); }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { delete h; return fail(nullptr, SWARM_ERR_HIP, "hipGetDeviceProperties failed"); }
    if ((size_t)k.smem_bytes > (size_t)prop.sharedMemPerBlock && (size_t)k.smem_bytes > 160 * 1024) {
        delete h;
        return fail(nullptr, SWARM_ERR_INVALID, "configuration needs more LDS per workgroup than the device has (reduce n_cells_max / num_obs_grid_max)");
    }
    const size_t E = (size_t)cfg->n_env, N = (size_t)cfg->n_agents;
    hipError_t a = hipSuccess;
    auto alloc = [&](void **p, size_t bytes) { if (a == hipSuccess) a = hipMalloc(p, bytes); };
    alloc((void **)&h->d_p, E * 2 * N * 8); alloc((void **)&h->d_dp, E * 2 * N * 8);
    alloc((void **)&h->d_cells, E * 2 * (size_t)k.ng_max * 8); alloc((void **)&h->d_cin, E * 8);
    alloc((void **)&h->d_ng, E * 4);
    alloc((void **)&h->d_nei, E * N * (size_t)k.topo * 4); alloc((void **)&h->d_near, E * N * 4);
    alloc((void **)&h->d_inflag, E * N * 4);
    if (a == hipSuccess) a = hipMemset(h->d_ng, 0, E * 4);
    if (a == hipSuccess) a = hipMemset(h->d_nei, 0xFF, E * N * (size_t)k.topo * 4);
    if (a == hipSuccess) a = hipMemset(h->d_near, 0, E * N * 4);
    if (a == hipSuccess) a = hipMemset(h->d_inflag, 0, E * N * 4);
    if (a == hipSuccess) a = hipMemset(h->d_cells, 0, E * 2 * (size_t)k.ng_max * 8);
    if (a == hipSuccess) a = hipEventCreate(&h->ev0);
    if (a == hipSuccess) a = hipEventCreate(&h->ev1);
    if (a != hipSuccess) {
        std::string m = std::string("device allocation failed: ") + hipGetErrorString(a);
        swarm_destroy(h);
        return fail(nullptr, SWARM_ERR_HIP, m);
    }
    k.p = h->d_p; k.dp = h->d_dp; k.nei = h->d_nei; k.near_cell = h->d_near; k.in_flag = h->d_inflag;
    k.cells = h->d_cells; k.n_g = h->d_ng; k.c_in = h->d_cin;
    *out = h;
    return SWARM_OK;
}

int swarm_destroy(swarm_env_t *h)
{
    if (!h) return SWARM_OK;
    {
        DeviceGuard g(h->device);
        (void)hipStreamSynchronize(h->stream);
        (void)hipFree(h->d_p); (void)hipFree(h->d_dp); (void)hipFree(h->d_cells); (void)hipFree(h->d_cin);
        (void)hipFree(h->d_ng); (void)hipFree(h->d_nei); (void)hipFree(h->d_near); (void)hipFree(h->d_inflag);
        (void)hipFree(h->d_exp_sensed); (void)hipFree(h->d_exp_occ);
        if (h->ev0) (void)hipEventDestroy(h->ev0);
        if (h->ev1) (void)hipEventDestroy(h->ev1);
    }
    delete h;
    return SWARM_OK;
}

int swarm_set_stream(swarm_env_t *h, void *s)
{
    if (!h) return SWARM_ERR_INVALID;
    h->stream = static_cast<hipStream_t>(s);
    return SWARM_OK;
}

int swarm_synchronize(swarm_env_t *h)
{
    if (!h) return SWARM_ERR_INVALID;
    DeviceGuard g(h->device);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SWARM_OK;
}

int swarm_obs_dim(const swarm_env_t *h) { return h ? h->kp.obs_dim : -1; }

int swarm_set_cells(swarm_env_t *h, int env_begin, int count, const double *cells, const int32_t *n_g, const double *l_cell)
{
    if (!h) return SWARM_ERR_INVALID;
    if (!cells || !n_g || !l_cell) return fail(h, SWARM_ERR_INVALID, "swarm_set_cells: null argument");
    if (env_begin < 0 || count < 1 || env_begin + count > h->cfg.n_env) return fail(h, SWARM_ERR_INVALID, "swarm_set_cells: env range out of bounds");
    std::vector<double> cin((size_t)count);
    for (int k = 0; k < count; ++k) {
        if (n_g[k] < 1 || n_g[k] > h->cfg.n_cells_max) return fail(h, SWARM_ERR_INVALID, "swarm_set_cells: n_g must be in [1, n_cells_max]");
        if (!(l_cell[k] > 0)) return fail(h, SWARM_ERR_INVALID, "swarm_set_cells: l_cell must be positive");
        cin[(size_t)k] = cut_lt(std::sqrt(2) * l_cell[k] / 2);            // CPP:889
    }
    DeviceGuard g(h->device);
    const size_t row = (size_t)2 * h->kp.ng_max;
    HIP_TRY(h, hipMemcpyAsync(h->d_cells + (size_t)env_begin * row, cells, (size_t)count * row * 8, hipMemcpyDefault, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_ng + env_begin, n_g, (size_t)count * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_cin + env_begin, cin.data(), (size_t)count * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));           // cin is a host temporary
    for (int k = 0; k < count; ++k) h->cells_set[(size_t)(env_begin + k)] = 1;
    h->have_cells = true;
    for (char c : h->cells_set) if (!c) { h->have_cells = false; break; }
    h->observed = false;
    return SWARM_OK;
}

int swarm_set_state(swarm_env_t *h, const double *p, const double *dp)
{
    if (!h) return SWARM_ERR_INVALID;
    if (!p || !dp) return fail(h, SWARM_ERR_INVALID, "swarm_set_state: null argument");
    DeviceGuard g(h->device);
    const size_t bytes = (size_t)h->cfg.n_env * 2 * h->cfg.n_agents * 8;
    HIP_TRY(h, hipMemcpyAsync(h->d_p, p, bytes, hipMemcpyDefault, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_dp, dp, bytes, hipMemcpyDefault, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->have_state = true;
    h->observed = false;
    return SWARM_OK;
}

int swarm_get_state(swarm_env_t *h, double *p, double *dp)
{
    if (!h) return SWARM_ERR_INVALID;
    DeviceGuard g(h->device);
    const size_t bytes = (size_t)h->cfg.n_env * 2 * h->cfg.n_agents * 8;
    if (p) HIP_TRY(h, hipMemcpyAsync(p, h->d_p, bytes, hipMemcpyDefault, h->stream));
    if (dp) HIP_TRY(h, hipMemcpyAsync(dp, h->d_dp, bytes, hipMemcpyDefault, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SWARM_OK;
}

int swarm_observe(swarm_env_t *h, void *obs)
{
    if (!h) return SWARM_ERR_INVALID;
    if (!h->have_cells) return fail(h, SWARM_ERR_STATE, "swarm_observe: target cells not set for every env (swarm_set_cells)");
    if (!h->have_state) return fail(h, SWARM_ERR_STATE, "swarm_observe: state not set (swarm_set_state)");
    DeviceGuard g(h->device);
    int rc = launch(h, false, nullptr, 0, obs, nullptr, nullptr, nullptr);
    if (rc == SWARM_OK) h->observed = true;
    return rc;
}

int swarm_step(swarm_env_t *h, const void *action, int action_dtype, void *obs, float *reward, uint8_t *done, void *a_prior)
{
    if (!h) return SWARM_ERR_INVALID;
    if (!action) return fail(h, SWARM_ERR_INVALID, "swarm_step: null action");
    if (action_dtype != SWARM_F32 && action_dtype != SWARM_F64) return fail(h, SWARM_ERR_INVALID, "swarm_step: bad action_dtype");
    if (!h->observed) return fail(h, SWARM_ERR_STATE, "swarm_step: call swarm_observe after setting cells/state (the reference's reset() ends with _get_obs())");
    DeviceGuard g(h->device);
    return launch(h, true, action, action_dtype == SWARM_F64, obs, reward, done, a_prior);
}

int swarm_get_indices(swarm_env_t *h, int32_t *neighbor_index, int32_t *in_flags, int32_t *sensed_index, int32_t *occupied_index)
{
    if (!h) return SWARM_ERR_INVALID;
    if (!h->observed) return fail(h, SWARM_ERR_STATE, "swarm_get_indices: nothing observed yet");
    DeviceGuard g(h->device);
    const size_t EN = (size_t)h->cfg.n_env * h->cfg.n_agents;
    if (sensed_index || occupied_index) {
        if (!h->d_exp_sensed) {
            HIP_TRY(h, hipMalloc((void **)&h->d_exp_sensed, EN * (size_t)h->kp.g_max * 4));
            HIP_TRY(h, hipMalloc((void **)&h->d_exp_occ, EN * (size_t)h->kp.occ_max * 4));
        }
        // re-run the observation pass on the current state with the export switched on; it recomputes the
        // same caches from the same state, so it is idempotent.
        h->kp.export_idx = 1; h->kp.exp_sensed = h->d_exp_sensed; h->kp.exp_occ = h->d_exp_occ;
        int rc = launch(h, false, nullptr, 0, nullptr, nullptr, nullptr, nullptr);
        h->kp.export_idx = 0;
        if (rc != SWARM_OK) return rc;
        if (sensed_index) HIP_TRY(h, hipMemcpyAsync(sensed_index, h->d_exp_sensed, EN * (size_t)h->kp.g_max * 4, hipMemcpyDefault, h->stream));
        if (occupied_index) HIP_TRY(h, hipMemcpyAsync(occupied_index, h->d_exp_occ, EN * (size_t)h->kp.occ_max * 4, hipMemcpyDefault, h->stream));
    }
    if (neighbor_index) HIP_TRY(h, hipMemcpyAsync(neighbor_index, h->d_nei, EN * (size_t)h->kp.topo * 4, hipMemcpyDefault, h->stream));
    if (in_flags) HIP_TRY(h, hipMemcpyAsync(in_flags, h->d_inflag, EN * 4, hipMemcpyDefault, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SWARM_OK;
}

double swarm_step_algorithmic_bytes(const swarm_env_t *h)
{
    if (!h) return 0.0;
    // Per agent-step: action 2*4 r, state p/dp 4*8 r + 4*8 w (fp64 here), obs D*sizeof w, reward 4 + done 1 +
    // prior 2*sizeof w; per env: target cells 2*n_g_max*8 r.  (with this build's dtypes.)
    const double so = h->cfg.obs_dtype == SWARM_F64 ? 8.0 : 4.0;
    const double per_agent = 8.0 + 64.0 + h->kp.obs_dim * so + 5.0 + 2.0 * so;
    return (double)h->cfg.n_env * (h->cfg.n_agents * per_agent + 2.0 * h->kp.ng_max * 8.0);
}

int swarm_timer_start(swarm_env_t *h)
{
    if (!h) return SWARM_ERR_INVALID;
    DeviceGuard g(h->device);
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    return SWARM_OK;
}

int swarm_timer_stop(swarm_env_t *h, float *ms)
{
    if (!h || !ms) return SWARM_ERR_INVALID;
    DeviceGuard g(h->device);
    HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    HIP_TRY(h, hipEventSynchronize(h->ev1));
    HIP_TRY(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
    return SWARM_OK;
}

}  // extern "C"
