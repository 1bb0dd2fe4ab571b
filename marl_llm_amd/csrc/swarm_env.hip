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
// of 128/256 agents takes 2/4 wavefronts.  A workgroup holds WPE copies ("splits") of its agent threads
// that share the env's LDS footprint: sequential per-agent work (forces, prior, ordered neighbour
// insertion, reward decision) runs on one split each, everything else is dealt over all of them.
// Agent positions / velocities live in LDS and are read with wave-uniform addresses (LDS broadcast).
// Target cells: when they form a lattice (the reference's tiled shapes always do) the sensed / covered /
// nearest-cell queries walk lattice ROWS with bit operations on row masks and need no coordinates; the
// exact fp64 coordinates are gathered from an interleaved copy in global memory only where a value or an
// exact tie-break needs them.  Arbitrary cell sets take an fp32 pre-filter scan with exact fp64 fallback.
// The observation block of an environment is streamed out with consecutive lanes writing consecutive
// addresses (the rows of one env are contiguous in HBM).  DESIGN.md section 3 has the phase list.
//
// Numerics: everything that decides an index, a flag, the state or the reward is IEEE double in the
// reference's operation order; this file MUST be compiled with -ffp-contract=off (the reference is
// built for baseline x86-64, no FMA).  Threshold tests of the form sqrt(d2) < t are evaluated as
// d2 < cut(t) with cut(t) = the smallest double whose correctly rounded sqrt is >= t, computed on the
// host; sqrt is monotonic, so the two tests are equivalent bit for bit and the scans need no sqrt.
//
// There is no CPU path in this file.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "swarm_env.h"

namespace {

constexpr int kTopoMax = 6;
constexpr int kNeiStride = 8;     // shorts per agent in the LDS neighbour list: 6 ids, [6] = collision flag
typedef unsigned long long u64;

// Per-environment description of the target cells as a subset of a (rotated) square lattice, when they are one
// (the reference tiles a silhouette image into square cells and rotates / shifts them: assembly_cfg.py:56-99,
// assembly.py:175-187).  Cell (column a, row b) sits at o + a*u + b*v and the cell index order is row-major.
struct LatEnv {
    double ox, oy;
    double uxi, uyi, vxi, vyi;    // (p - o) . (uxi, uyi) = column coordinate, (p - o) . (vxi, vyi) = row coordinate
    float R, Rc;                  // d_sen / l and (r_avoid / 2) / l in lattice steps
    int nrows, ncols;
    short rowstart[64];           // cell index of the first cell of each row
    unsigned long long rowmask[64];   // occupied columns of each row
};

struct KP {
    int n_env, n_a, ng_max, ngw, topo, g_max, occ_max, obs_dim;
    int with_self, periodic, boundary, with_prior, export_idx;
    int export_small;          // also write neighbor_index / nearest cell / in_flags to HBM (export launches only: the step itself keeps them in LDS)
    int cxy_stride;            // double2 elements per env in LDS
    int cxq_stride;            // floats per env in the fp32 pair layout
    int g_stride;              // int16 elements per agent row in LDS
    int off_cxy, off_sp, off_cmask, off_sbits, off_obits, off_sidx, off_snei, off_sncf, off_snear, off_pc;
    int smem_lat, smem_lat_export, smem_generic;   // dynamic LDS bytes by launch kind
    double c_sen, c_near, c_occ, c_avoid, c_ball;     // squared-distance cut-offs
    double c_close, c_close2;  // (1.9 r_avoid)^2 and (3 r_avoid)^2 capped at c_sen: pre-selection radii of the neighbour insertion (any values are exact; the second is used for N > 128)
    // fp32 pre-filter bands: d2_32 < *_lo  =>  exact test true;  d2_32 >= *_hi  =>  exact test false
    float csen_lo, csen_hi, cocc_lo, cocc_hi;
    float coord_lim;           // |coordinate| bound the bands were derived for
    float min_tol_a, min_tol_b;   // nearest-cell ambiguity tolerance: a*sqrt(d2) + b*d2
    float rew_ga, rew_gb;      // the reward is re-evaluated in fp64 when | |v| - 0.05 | <= rew_ga * n / den + rew_gb
    int force_exact;           // debug: take every exact fallback path
    int cap_int;               // G-1 odd: the cap's round(i*step) is an exact integer division by 2(G-1)
    unsigned cap_magic; int cap_shift;
    int dbg_phase, dbg_extra;  // diagnostics only (tools/ablate.py): run phase dbg_phase dbg_extra EXTRA times; the
                               // phases are idempotent, so results are unchanged and the extra cost is the phase's cost
    int off_cxyf, off_partc, off_lat, off_cov, off_flag;
    // lattice (row-space) launches only: per-agent frame, per (window row, agent) column masks / first cell index, per-agent
    // row counts, the agent permutation of the list phase, the fp32 reward verdicts, the occupied columns (export only)
    int off_hdr, off_srow, off_pcr, off_perm, off_rres, off_orow, off_partd;
    float rew_ga_lat, rew_gb_lat;   // guard band of the fp32 reward decision in lattice steps (see swarm_create)
    float rew_thr_k;           // 0.05 / d_sen: the reward's |v| threshold in lattice steps is rew_thr_k * (d_sen / l)
    int lattice;               // every env's cells are a lattice subset whose sensing window is <= 15 rows: row-space path
    int lat_rw, lat_cw;        // row half-windows (lattice steps) for d_sen and r_avoid/2
    int lat_nrs, lat_nrc;      // rows a radius can touch: floor(2 (rho_max + margin)) + 1, for d_sen and r_avoid/2
    int lat_n32;               // every env's lattice has <= 32 columns: 32-bit row masks
    double c_near_hi;          // c_near * (1 + 1e-9): pairs in [c_near, c_near_hi) flag the exact occupied-cell path
    const LatEnv *lat;
    double d_sen, r_avoid, size_a, size2, k_ball, k_wall, c_wall, vel_max, dt;
    double bx0, by1, bx2, by3, w_half, h_half;
    double *p, *dp;
    int *nei, *near_cell, *in_flag;
    double2 *sf_next;          // [E][N]: contact-spring force on agent i in the CURRENT state = the force term of the next step
    const double *cells;       // [E][2][ng_max] (the ABI's layout)
    const double2 *cells_xy;   // [E][ng_max] (x, y) interleaved copy: one 16-byte gather per cell
    const int *n_g;
    const double *c_in;
    int *exp_sensed, *exp_occ;
    void *prior_next;          // [E][N] pairs of the handle's obs dtype: the prior policy of the next step (written by every pass)
    double pk_att, pk_rep, pk_ali;   // gains of the prior policy: attraction, repulsion, alignment (CPP:1128-1132: 2, 3, 2)
    double pk_llm;             // repulsion gain of the Python twin that drives agent_strategy == 'llm' (ENV:895: 1.0)
    int llm;                   // also evaluate that twin and leave it in act_next as the NEXT step's action (ENV:525-529)
    double2 *act_next;         // [E][N]
    long long *stamps;         // diagnostic build only (-DSWARM_STAMPS): per-block phase clocks
};

typedef float f2v __attribute__((ext_vector_type(2)));

template <typename T> struct Pair;
template <> struct Pair<float>  { typedef float2 type; };
template <> struct Pair<double> { typedef double2 type; };
typedef __attribute__((ext_vector_type(2))) __bf16 bf2v;
template <> struct Pair<__bf16> { typedef bf2v type; };

// output conversion of an exact fp64 value: one rounding for f64 / f32; bf16 = the f32 value rounded again (RNE), i.e. what a
// consumer gets from `obs_f32.to(bfloat16)`
template <typename OT> __device__ __forceinline__ OT to_out(double v) { return (OT)v; }
// streaming store of one observation pair: the observation block (768 B per agent and step, written once, read by another
// kernel) would otherwise sweep the target cells, lattice rows and agent state of every environment out of the L2 between
// launches; the non-temporal hint keeps those resident for the next step's gathers
typedef double d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_nt(float2 *dst, float2 v) { f2v t = {v.x, v.y}; __builtin_nontemporal_store(t, reinterpret_cast<f2v *>(dst)); }
__device__ __forceinline__ void store_nt(double2 *dst, double2 v) { d2v t = {v.x, v.y}; __builtin_nontemporal_store(t, reinterpret_cast<d2v *>(dst)); }
__device__ __forceinline__ void store_nt(bf2v *dst, bf2v v) { __builtin_nontemporal_store(v, dst); }
template <> __device__ __forceinline__ __bf16 to_out<__bf16>(double v) { return (__bf16)(float)v; }

__device__ __forceinline__ void wrap_rel(double &x, double &y, double wh, double hh)
{   // CPP:700-715
    if (x < -wh) x += 2 * wh; else if (x > wh) x -= 2 * wh;
    if (y < -hh) y += 2 * hh; else if (y > hh) y -= 2 * hh;
}

// gfx950 hazard: a VALU instruction that reads an SGPR (lane mask / scalar source) written by a VALU instruction -- the
// v_cmp that produced the mask -- needs two wait states in between.  The compiler's hazard recogniser inserts them for its
// own instructions but cannot see inside inline asm, so both helpers below carry their own `s_nop 1` (it stalls only this
// wave for two cycles; other waves issue meanwhile).  Without it the asm reads a stale mask whenever the scheduler happens
// to place it right behind the compare.

// old[LANE] = value (wave-uniform value, compile-time lane): one v_writelane_b32 (the lane select must be an
// inline constant: a second SGPR operand would violate the constant-bus limit).
template <int LANE>
__device__ __forceinline__ int writelane_c(int value, int old)
{
    asm("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(value), "n"(LANE));
    return old;
}

// (acc << 1) | bit, the bit coming per lane from a 64-bit lane mask (a compare result): ONE v_addc_co_u32 with
// the mask as carry-in.
__device__ __forceinline__ unsigned shl1_or_mask(unsigned acc, unsigned long long mask)
{
    unsigned long long carry_out;
    asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(acc), "=s"(carry_out) : "v"(acc), "s"(mask));
    return acc;
}

// The pair pass's five threshold tests of one (i, j) pair in ONE asm block: all compares first, then all add-with-carry
// accumulations (acc = 2 acc + test).  The compare that feeds an accumulation is at least four instructions ahead of it, so
// the VALU-writes-SGPR -> VALU-reads-it hazard (two wait states) is covered by the block's own instructions: no s_nop at all
// (shl1_or_mask pays one per test).  d2u: un-wrapped squared distance (nearby / exception band / contact), d2: wrapped
// (candidates, close candidates).
__device__ __forceinline__ void pair_tests5(double d2u, double d2, double c_nb, double c_hi, double c_ht, double c_cd, double c_c1,
                                            unsigned &nb, unsigned &hi, unsigned &ht, unsigned &cd, unsigned &c1)
{
    unsigned long long m0, m1, m2, m3, m4;
    asm("v_cmp_lt_f64_e64 %5, %10, %12\n\t"
        "v_cmp_lt_f64_e64 %6, %10, %13\n\t"
        "v_cmp_lt_f64_e64 %7, %10, %14\n\t"
        "v_cmp_lt_f64_e64 %8, %11, %15\n\t"
        "v_cmp_lt_f64_e64 %9, %11, %16\n\t"
        "v_addc_co_u32_e64 %0, %5, %0, %0, %5\n\t"
        "v_addc_co_u32_e64 %1, %6, %1, %1, %6\n\t"
        "v_addc_co_u32_e64 %2, %7, %2, %2, %7\n\t"
        "v_addc_co_u32_e64 %3, %8, %3, %3, %8\n\t"
        "v_addc_co_u32_e64 %4, %9, %4, %4, %9"
        : "+v"(nb), "+v"(hi), "+v"(ht), "+v"(cd), "+v"(c1), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(m4)
        : "v"(d2u), "v"(d2), "s"(c_nb), "s"(c_hi), "s"(c_ht), "s"(c_cd), "s"(c_c1));
}

// The same with the sixth test of the N > 128 instantiations (the wider pre-selection ring) in the block: the test costs its
// compare and its accumulation, no wait state.
__device__ __forceinline__ void pair_tests6(double d2u, double d2, double c_nb, double c_hi, double c_ht, double c_cd, double c_c1, double c_c2,
                                            unsigned &nb, unsigned &hi, unsigned &ht, unsigned &cd, unsigned &c1, unsigned &c2)
{
    unsigned long long m0, m1, m2, m3, m4, m5;
    asm("v_cmp_lt_f64_e64 %6, %12, %14\n\t"
        "v_cmp_lt_f64_e64 %7, %12, %15\n\t"
        "v_cmp_lt_f64_e64 %8, %12, %16\n\t"
        "v_cmp_lt_f64_e64 %9, %13, %17\n\t"
        "v_cmp_lt_f64_e64 %10, %13, %18\n\t"
        "v_cmp_lt_f64_e64 %11, %13, %19\n\t"
        "v_addc_co_u32_e64 %0, %6, %0, %0, %6\n\t"
        "v_addc_co_u32_e64 %1, %7, %1, %1, %7\n\t"
        "v_addc_co_u32_e64 %2, %8, %2, %2, %8\n\t"
        "v_addc_co_u32_e64 %3, %9, %3, %3, %9\n\t"
        "v_addc_co_u32_e64 %4, %10, %4, %4, %10\n\t"
        "v_addc_co_u32_e64 %5, %11, %5, %5, %11"
        : "+v"(nb), "+v"(hi), "+v"(ht), "+v"(cd), "+v"(c1), "+v"(c2), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(m4), "=&s"(m5)
        : "v"(d2u), "v"(d2), "s"(c_nb), "s"(c_hi), "s"(c_ht), "s"(c_cd), "s"(c_c1), "s"(c_c2));
}

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N-1>{})
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ double clamp_ref(double v, double lo, double hi)
{   // CPP:11-14  std::max(lo, std::min(v, hi))
    double m = (hi < v) ? hi : v;
    return (lo < m) ? m : lo;
}

#define REPS(k) ((P.dbg_phase == (k)) ? P.dbg_extra + 1 : 1)
#define FENCE() asm volatile("" ::: "memory")
// diagnostics only (tools/ablate.py --cumulative): leave the kernel after segment k (debug phase 15, extra = k); outputs
// are then incomplete, so only timing / counter runs use it
#define EXIT_AT(k) do { if (P.dbg_phase == 15 && P.dbg_extra == (k)) return; } while (0)

#ifdef SWARM_STAMPS
// written straight to global memory by lane 0 of every wave (no registers held across the kernel)
#define STAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (P.stamps != nullptr && (threadIdx.x & 63) == 0) P.stamps[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 24 + (k)] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

// Workgroup geometry.  AG "agent threads" hold the agents of the workgroup's environment(s) (lane = agent);
// the workgroup has WPE = 4 copies ("splits") of them (256 agents: 1024 threads).  Splits "A" and "B" own the
// per-agent sequential work (forces / integration / reward decision; prior / ordered neighbour insertion); rows,
// words, slots and rank ranges of the other phases are dealt over all splits.  One environment's LDS footprint is
// thereby shared by four times more wavefronts, which is what buys the occupancy that hides the LDS / fp64 latencies.
// HALF (N < 64, lattice launches, small grids): only half of the 64 agent threads hold agents -- half as many environments
// per workgroup, twice as many workgroups -- and the list phase gives every agent EIGHT lanes instead of four.  A batch that
// fills a fraction of the chip (32 agents x 1024 envs: 512 workgroups on 256 CUs) is bound by the length of one workgroup's
// serial chain, not by throughput: shorter chains on more workgroups.
template <int NPAD, bool HALF = false> struct Geo {
    static_assert(!HALF || NPAD < 64, "the half-occupied geometry is for N < 64");
    static constexpr int AG = NPAD < 64 ? 64 : NPAD;
    static constexpr int EPB = NPAD < 64 ? (64 / NPAD) / (HALF ? 2 : 1) : 1;
    static constexpr int ACTW = NPAD < 64 ? EPB * NPAD : 64;     // agent threads of a 64-group that hold agents
    static constexpr int LPA = HALF ? 8 : 4;                     // lanes per agent in the list phase
    static constexpr int AGW = 64 / LPA;                         // agents per wave there (= ACTW / 4)
    static constexpr int NW = AG / 64;
    static constexpr int WPE = 4;
    static constexpr int T = AG * WPE;
#ifndef SWARM_WPS
#define SWARM_WPS 7
#endif
#ifndef SWARM_WPS_SMALL
#define SWARM_WPS_SMALL 6
#endif
    // waves per SIMD the register allocation aims for.  N = 64, lattice kernels: SEVEN 4-wave workgroups per CU (<= 72 VGPRs --
    // the allocator needs 65 -- and 22,976 B of LDS each: 18 of the CU's 128 allocation granules of 1280 B; the kernel loses
    // 7 % from six to five workgroups per CU and gains 4 % from six to seven).  N < 64 (several environments per wavefront:
    // their lattice tables make it 19 - 21 granules): six (76 - 78 VGPRs, no scratch; 32 x 8192: 103.6 -> 97.0 us against
    // five).  N = 128 (512 threads, ~47 KB of LDS): three workgroups per CU instead of two (80 VGPRs, 24 B of scratch;
    // 128 x 4096: 218 -> 185 us).  The generic-scan kernels: six / five.  N = 256: one 16-wave workgroup per CU.
#ifndef SWARM_WPS_128
#define SWARM_WPS_128 6
#endif
    static constexpr int WPS_LAT = NPAD == 64 ? SWARM_WPS : (NPAD < 64 ? SWARM_WPS_SMALL : (NPAD == 128 ? SWARM_WPS_128 : 1));
    static constexpr int WPS_GEN = NPAD == 64 ? 6 : (NPAD < 64 ? 5 : 1);
};

constexpr double kSentinel = 1.0e200;     // coordinates of padding cells: d2 overflows to +inf

// cos(pi * t) for t in [0, 1], absolute error ~2e-16 (Taylor in x = pi*min(t, 1-t) <= pi/2 up to x^22).
// Only the reward's psi weights use it (CPP:1012-1020 calls libm cos(M_PI * z / r)); it is not bit-identical
// to glibc's cos and does not need to be: it feeds a sum that is compared with a threshold.
__device__ __forceinline__ double cospi01(double t)
{
    const bool flip = t > 0.5;
    const double r = flip ? 1.0 - t : t;
    const double x = M_PI * r;
    const double u = x * x;
    double c = -8.8967913924505741e-22;                 // -1/22!
    c = fma(c, u, 4.1103176233121648e-19);              // +1/20!
    c = fma(c, u, -1.5619206968586225e-16);
    c = fma(c, u, 4.7794773323873853e-14);
    c = fma(c, u, -1.1470745597729725e-11);
    c = fma(c, u, 2.0876756987868100e-09);
    c = fma(c, u, -2.7557319223985888e-07);
    c = fma(c, u, 2.4801587301587302e-05);
    c = fma(c, u, -1.3888888888888889e-03);
    c = fma(c, u, 4.1666666666666664e-02);
    c = fma(c, u, -0.5);
    c = fma(c, u, 1.0);
    return flip ? -c : c;
}

// fp32 fast path of the reward weight: psi(u) = 1/2 (1 + cos(pi sqrt(u))), u = (z/d_sen)^2 in [0, 1], as a
// degree-6 minimax polynomial in u (cos(pi sqrt(u)) is entire in u): no sqrt, no range reduction.
// |abs err| < 3e-7 in fp32 Horner form (fit error 5.5e-9).
__device__ __forceinline__ float psi_u_f32(float u)
{
    float c = 7.969553699e-04f;
    c = fmaf(c, u, -1.267949212e-02f);
    c = fmaf(c, u, 1.175149009e-01f);
    c = fmaf(c, u, -6.675792336e-01f);
    c = fmaf(c, u, 2.029347420e+00f);
    c = fmaf(c, u, -2.467400551e+00f);
    c = fmaf(c, u, 1.0f);
    return c;
}

// LAT: every env's target cells are a lattice subset (host-detected, KP::lattice) -- a compile-time switch, so that each
// instantiation carries ONE cell path and stays inside the 64 KB instruction cache.
// The same weight as a degree-5 polynomial (Chebyshev-node fit; |abs err| < 6.5e-7 in fp32 Horner form): one fused multiply-add
// less per kept cell in the lattice path's list walk; the reward's guard band there allows 1.2e-6 for it (set_lattice_mode).
__device__ __forceinline__ float psi5_u_f32(float u)
{
    float c = -1.028585434e-02f;
    c = fmaf(c, u, 1.148182452e-01f);
    c = fmaf(c, u, -6.661784649e-01f);
    c = fmaf(c, u, 2.029018402e+00f);
    c = fmaf(c, u, -2.467372179e+00f);
    c = fmaf(c, u, 9.999995828e-01f);
    return c;
}

template <int NPAD, typename OT, bool DO_STEP, bool LAT, bool HALF = false>
__global__ void __launch_bounds__((Geo<NPAD, HALF>::T), (LAT ? Geo<NPAD, HALF>::WPS_LAT : Geo<NPAD, HALF>::WPS_GEN))
k_env(const KP P, const void *__restrict__ action, const int act_f64, OT *__restrict__ obs,
      float *__restrict__ reward, uint8_t *__restrict__ done, OT *__restrict__ a_prior)
{
    typedef Geo<NPAD, HALF> G_;
    constexpr int AG = G_::AG, EPB = G_::EPB, NW = G_::NW, WPE = G_::WPE, T = G_::T;
    constexpr int ACTW = G_::ACTW, LPA = G_::LPA, AGW = G_::AGW;
    static_assert(!HALF || LAT, "the half-occupied geometry exists for the lattice path only");
    typedef typename Pair<OT>::type OT2;

    extern __shared__ __align__(16) unsigned char smem[];
    float *cxq = reinterpret_cast<float *>(smem + P.off_cxyf);          // fp32 cells, per PAIR {xa, xb, ya, yb} (pre-filter)
    double *sp = reinterpret_cast<double *>(smem + P.off_sp);            // [4][AG]: px, py, vx, vy
    u64 *cmask = reinterpret_cast<u64 *>(smem + P.off_cmask);            // [cell][NW]
    float *rsum = reinterpret_cast<float *>(smem + P.off_cmask);         // [WPE][3][AG]  (aliases cmask, later phase)
    unsigned *sbits = reinterpret_cast<unsigned *>(smem + P.off_sbits);  // [word][AG]
    unsigned *obits = reinterpret_cast<unsigned *>(smem + P.off_obits);  // [word][AG] (export launches only)
    short *sidx = reinterpret_cast<short *>(smem + P.off_sidx);          // [AG][g_stride]
    typedef std::conditional_t<LAT, short, int> pc_t;                    // (lattice launches: 16-bit -- the kernel's LDS budget is seven workgroups per CU)
    pc_t *part_c = reinterpret_cast<pc_t *>(smem + P.off_partc);         // [WPE][AG] per-split nearest-cell candidates (cell index < 2^15)
    u64 *pm = reinterpret_cast<u64 *>(smem + P.off_sidx);                // [WPE][2 or 3][NW][AG] partial pair masks (aliases sidx, earlier phase)
    unsigned *owords = reinterpret_cast<unsigned *>(smem + P.off_cmask); // [word][AG] occupied bits (NW == 1; aliases cmask)
    short *snei = reinterpret_cast<short *>(smem + P.off_snei);          // [AG][kTopoMax]
    int *sncf = reinterpret_cast<int *>(smem + P.off_sncf);              // [AG]: nearest cell | in_flag<<30
    u64 *snear = reinterpret_cast<u64 *>(smem + P.off_snear);            // [NW][AG] nearby-agent masks
    u64 *lrm = reinterpret_cast<u64 *>(smem + P.off_lat);                // [EPB][64] lattice row masks
    short *lrs = reinterpret_cast<short *>(smem + P.off_lat + (size_t)EPB * 64 * 8);   // [EPB][64] row starts
    unsigned *cov = reinterpret_cast<unsigned *>(smem + P.off_cov);      // [EPB][ngw+1] cells within r_avoid/2 of ANY agent
    int *sflag = reinterpret_cast<int *>(smem + P.off_flag);             // per-lane exception flags: generic launches [AG] ints; lattice launches one BYTE per agent thread (flag_* below)
    unsigned char *pc = smem + P.off_pc;                                 // [word][AG] kept-bit counts
    // row-space representation of the lattice path (LAT): window row t of agent thread `at` = lattice row b0 + t, its
    // columns are stored relative to the agent's first column ca0 (<= 17 columns are ever in range: 32-bit words)
    constexpr int NRC = 16;                                              // window rows stored per agent (lat_nrs <= 15)
    float4 *hdr = reinterpret_cast<float4 *>(smem + P.off_hdr);          // [AG] {apr = a - ca0, bpr = b - b0, b0, ca0} (last two: ints)
    unsigned *srow = reinterpret_cast<unsigned *>(smem + P.off_srow);    // [NRC][AG] sensed, then kept columns of window row t (17 bits) | cell index of the row's column ca0 << 17
    unsigned char *pcr = smem + P.off_pcr;                               // [AG][NRC] kept cells per window row
    u64 *covrow = reinterpret_cast<u64 *>(smem + P.off_cov);             // [EPB][64] columns within r_avoid/2 of ANY agent, per lattice row
    unsigned char *perm = smem + P.off_perm;                             // [T/64][64] agent threads in ascending list length (per wave)
    unsigned *orow = reinterpret_cast<unsigned *>(smem + P.off_orow);    // [NRC][AG] occupied columns (export launches only)

    const int tid = threadIdx.x, lane = tid & 63;
    STAMP(0);
    const int at = tid % AG;                 // agent thread
    // split: wave-uniform (AG is a multiple of 64), kept in an SGPR.  The roles rotate with the workgroup index: the
    // k-th wave of every workgroup lands on the same SIMD, and splits A / B carry extra sequential work (forces, prior,
    // ordered insertion, reward combine) -- rotating spreads that over the four SIMDs of a CU.
    const int sx = __builtin_amdgcn_readfirstlane((tid / AG + (int)blockIdx.x) % WPE);
    const int aw = at >> 6;                  // which 64-agent group of the environment
    auto flag_get = [&]() -> bool { return LAT ? ((reinterpret_cast<const unsigned *>(sflag)[at >> 2] >> ((at & 3) * 8)) & 1u) != 0 : (sflag[at] & 1) != 0; };
    const bool thr_on = NPAD >= 64 || at < ACTW;         // (half-occupied geometry: agent threads ACTW..63 hold nothing)
    const int el = (NPAD < 64 && thr_on) ? at / NPAD : 0;
    const int i = NPAD < 64 ? at % NPAD : at;
    const int e = blockIdx.x * EPB + el;
    const int n_a = P.n_a;
    const bool act = thr_on && (e < P.n_env) && (i < n_a);
    const int es = e < P.n_env ? e : P.n_env - 1;
    const int ng = P.n_g[es];
    int ngb = ng;
    if (EPB > 1) {
        for (int k = 0; k < EPB; ++k) {
            const int ek = blockIdx.x * EPB + k;
            const int v = P.n_g[ek < P.n_env ? ek : P.n_env - 1];
            ngb = v > ngb ? v : ngb;
        }
    }
    const int W = (ngb + 31) >> 5;                       // target-cell words of this workgroup
    // Split roles: split 0 ("A") = forces / integration / reward combine; split SB ("B") = prior policy and
    // neighbour search, which run concurrently with A's work; every split scans target-cell words.  Words
    // [0, w0) belong to split B (it gets fewer), the rest are dealt round-robin to the other splits.
    constexpr int SB = WPE > 1 ? 1 : 0;
    int w0 = W;
    if (WPE > 1) { w0 = (W - 2 * (WPE - 1)) / WPE; w0 = w0 < 0 ? 0 : w0; }
    auto mine = [&](int w) -> bool {
        if (WPE == 1) return true;
        if (w < w0) return sx == SB;
        const int o = (w - w0) % (WPE - 1);            // o-th of the non-B splits
        return sx == (o < SB ? o : o + 1);
    };
    // exact (fp64) cell coordinates are read from global memory where they are needed (prior target, nearest-cell
    // merge, observation values, the rare exact fallbacks); the env's 8.7 KB of cells stay L1/L2 resident.
    const double2 *gce = P.cells_xy + (size_t)es * P.ng_max;
    auto cell64 = [&](int c) -> double2 { double2 g = gce[c < ng ? c : 0]; if (!(c < ng)) { g.x = kSentinel; g.y = kSentinel; } return g; };
    const float *cq_e = cxq + (size_t)el * P.cxq_stride;

    // ---- issue every global load of the step up front (their latency overlaps the cell staging)
    const size_t sbase = (size_t)es * 2 * n_a;
    double px = __builtin_nan(""), py = __builtin_nan(""), vx = 0.0, vy = 0.0;   // inactive lanes: NaN positions,
    double ax = 0.0, ay = 0.0;                                                    // every comparison is false
    double sf0x = 0.0, sf0y = 0.0;
    OT2 pri_now; pri_now.x = to_out<OT>(0.0); pri_now.y = to_out<OT>(0.0);
    const bool copy_prior = DO_STEP && P.with_prior && a_prior != nullptr;
    if (sx == 0 && act) {
        // the contact-spring force of THIS step (ENV:442-457 + CPP:735-815) is a function of the pre-integration positions
        // only: the previous pass evaluated it on exactly these positions, off its critical path, and left it in HBM
        if (DO_STEP) { const double2 sfl = P.sf_next[(size_t)e * n_a + i]; sf0x = sfl.x; sf0y = sfl.y; }
        px = P.p[sbase + i]; py = P.p[sbase + n_a + i];
        vx = P.dp[sbase + i]; vy = P.dp[sbase + n_a + i];
        if (DO_STEP) {
            // act_f64 bit 0: doubles; bit 1: the reference's (2, n_a) component-major layout with the envs side by side on
            // the agent axis (the numpy API, ENV:487), else agent-major pairs [E][N][2]
            if (act_f64 & 2) {
                const size_t a0 = (size_t)e * n_a + i, a1 = a0 + (size_t)P.n_env * n_a;
                if (act_f64 & 1) { ax = ((const double *)action)[a0]; ay = ((const double *)action)[a1]; }
                else { ax = (double)((const float *)action)[a0]; ay = (double)((const float *)action)[a1]; }
            }
            else if (act_f64 & 1) { const size_t ab = ((size_t)e * n_a + i) * 2; ax = ((const double *)action)[ab]; ay = ((const double *)action)[ab + 1]; }
            else { const float2 af = reinterpret_cast<const float2 *>(action)[(size_t)e * n_a + i]; ax = (double)af.x; ay = (double)af.y; }
            // the prior policy of THIS step (CPP:1061-1196 via ENV:605-624) is a function of the pre-integration state and
            // of the neighbour list / nearest cell of the previous observation: the previous launch evaluated it at its end,
            // where all of that sat in registers and LDS, and left it in HBM -- here it is only handed to the caller
            if (copy_prior) pri_now = reinterpret_cast<const OT2 *>(P.prior_next)[(size_t)e * n_a + i];
        }
    }
    // ---- generic (non-lattice) mode: stage an fp32 copy of the target cells (ENV: grid_center (2, n_g)) in LDS, laid
    // out per pair of cells {xa, xb, ya, yb} for packed arithmetic, padded with a sentinel (fp32: +inf).  The lattice
    // walk needs no cell coordinates except on its rare exact paths, which read the fp64 cells from global memory.
    constexpr bool use_lat = LAT;
    if constexpr (!use_lat)
    for (int rep = 0, reps = REPS(9); rep < reps; ++rep)
    for (int k = 0; k < EPB; ++k) {
        FENCE();
        const int ek0 = blockIdx.x * EPB + k;
        const int ek = ek0 < P.n_env ? ek0 : P.n_env - 1;
        const int ngk = P.n_g[ek];
        const double *gx = P.cells + (size_t)ek * 2 * P.ng_max;
        const double *gy = gx + P.ng_max;
        for (int c = tid; c < W * 32; c += T) {
            double2 g;
            g.x = c < ngk ? gx[c] : kSentinel;
            g.y = c < ngk ? gy[c] : kSentinel;
            float *q = cxq + (size_t)k * P.cxq_stride + (c >> 1) * 4 + (c & 1);
            q[0] = (float)g.x; q[2] = (float)g.y;
        }
    }
    if (sx == 0) { if constexpr (LAT) { if ((at & 3) == 0) sflag[at >> 2] = 0; } else sflag[at] = 0; }
    if constexpr (use_lat) {
        for (int q = tid; q < EPB * 64; q += T) {
            const int ek0 = blockIdx.x * EPB + (q >> 6);
            const LatEnv &Lq = P.lat[ek0 < P.n_env ? ek0 : P.n_env - 1];
            lrm[q] = Lq.rowmask[q & 63]; lrs[q] = Lq.rowstart[q & 63];
            covrow[q] = 0;                                                   // covered columns are OR-ed in
        }
        if (sx == WPE - 1) reinterpret_cast<uint4 *>(pcr)[at] = uint4{0u, 0u, 0u, 0u};   // rows past lat_nrs stay empty
    }
    // ---- forces + integration (split A): wall spring / damper + the contact spring the previous pass left in sf_next,
    // semi-implicit Euler; the other splits meanwhile initialise LDS, and ONE barrier publishes everything.
    double npx = px, npy = py, nvx = vx, nvy = vy;
    auto forces_integrate = [&]() {
        for (int rep = 0, reps = REPS(1); rep < reps; ++rep) {
            FENCE();
            const double sfx = sf0x, sfy = sf0y;                            // contact spring: evaluated by the previous pass (below)
            STAMP(8);
            double Fx = 1 * ax + sfx, Fy = 1 * ay + sfy;                       // ENV:638,640
            if (P.boundary) {                                                   // CPP:817-855, ENV:515-518
                const double d0 = px - P.size_a - P.bx0;
                const double d1 = P.by1 - (py + P.size_a);
                const double d2 = P.bx2 - (px + P.size_a);
                const double d3 = py - P.size_a - P.by3;
                const double a0 = d0 < 0 ? fabs(d0) : 0.0, a1 = d1 < 0 ? fabs(d1) : 0.0;
                const double a2 = d2 < 0 ? fabs(d2) : 0.0, a3 = d3 < 0 ? fabs(d3) : 0.0;
                const double sx_ = (a0 - a2) * P.k_wall, sy_ = (a3 - a1) * P.k_wall;
                const double v0 = d0 < 0 ? vx : 0.0, v2 = d2 < 0 ? vx : 0.0;
                const double v3 = d3 < 0 ? vy : 0.0, v1 = d1 < 0 ? vy : 0.0;
                const double gx = (-v0 - v2) * P.c_wall, gy = (-v3 - v1) * P.c_wall;
                Fx = Fx + sx_ + gx;
                Fy = Fy + sy_ + gy;
            }
            STAMP(10);
            // ---- integration, ENV:643-652
            nvx = vx + (Fx / 1.0) * P.dt; nvy = vy + (Fy / 1.0) * P.dt;
            nvx = nvx < -P.vel_max ? -P.vel_max : (nvx > P.vel_max ? P.vel_max : nvx);
            nvy = nvy < -P.vel_max ? -P.vel_max : (nvy > P.vel_max ? P.vel_max : nvy);
            npx = px + nvx * P.dt; npy = py + nvy * P.dt;
            if (P.periodic) {                                                    // ENV:773-776
                if (npx < P.bx0) npx += 2 * P.w_half;
                if (npx > P.bx2) npx -= 2 * P.w_half;
                if (npy < P.by3) npy += 2 * P.h_half;
                if (npy > P.by1) npy -= 2 * P.h_half;
            }
        }
    };
    auto publish_new_state = [&]() {
        sp[at] = npx; sp[AG + at] = npy; sp[2 * AG + at] = nvx; sp[3 * AG + at] = nvy;
        if (DO_STEP && act) {                              // (the observation-only pass leaves the state as it is)
            P.p[sbase + i] = npx; P.p[sbase + n_a + i] = npy;
            P.dp[sbase + i] = nvx; P.dp[sbase + n_a + i] = nvy;
            if (copy_prior) store_nt(&reinterpret_cast<OT2 *>(a_prior)[(size_t)e * n_a + i], pri_now);
        }
    };
    // The contact spring came from the previous pass (sf_next), so the integration needs nobody else's position: split A goes
    // from its loads straight to the new state and publishes it -- one barrier, for every N (round 2 parked the old
    // positions in LDS for the contact loop: a wave barrier at N <= 64, two more workgroup barriers above).
    if (sx == 0) {
        STAMP(11);                                                        // (diagnostic builds: the state / action loads have landed)
        // the whole workgroup waits for these ~100 instructions: issue them ahead of the CU's other waves (-1.5 %; the
        // same for the wave with the longest lists in the list phase, or for every wave past it, gained nothing)
        __builtin_amdgcn_s_setprio(3);
        if (DO_STEP) forces_integrate();
        publish_new_state();
        __builtin_amdgcn_s_setprio(0);
        STAMP(12);
    }
    __syncthreads();
    STAMP(1);
    EXIT_AT(0);
    px = sp[at]; py = sp[AG + at];            // (the velocities are re-read from LDS where the prior policy needs them:
                                              // held in registers across the whole kernel they were spilled to scratch)
    STAMP(2);
    EXIT_AT(1);

    // ---- pairwise masks + neighbour search, CPP:77-100 + _get_focused CPP:628-698: the topo nearest agents with
    // norm < d_sen (self removed), ascending; and the "nearby" agent mask of the occupied-cell filter
    // (CPP:152-164: un-wrapped distance < d_sen + r_avoid/2, self included).  For N <= 64 the branch-free mask
    // pass is dealt over all splits (a quarter of the agents j each) and OR-combined through LDS; the ordered
    // insertion of the (few) candidates runs on split B.
    constexpr int JN = NPAD < 64 ? NPAD : 64;                   // lanes >= n_a hold NaN positions: never candidates
    const double *spx = sp + el * NPAD, *spy = sp + AG + el * NPAD;
    // every split evaluates 1/WPE of the agents j of each 64-agent group (branch-free, unrolled); the partial masks
    // are OR-combined through LDS so that every lane ends up with its complete "nearby" masks (split B also with the
    // candidate masks)
    u64 nearbyN[NW], candN[NW], cand1N[NW], cand2N[NW], hitN[NW] = {};
    // which split evaluates the contact spring of the next step: B, beside the others' walk (at N = 256 B's insertion over
    // four 64-agent groups looks like the long pole in the stamps, but giving the spring to split A's walkers measured +1.5 %)
    constexpr int CS = SB;
    {
        constexpr int JQ = JN / WPE;                  // agents j per split and 64-agent group
        static_assert(JQ * WPE == JN && JQ <= 32, "pair pass: JN must split evenly into <= 32 agents per split");
        constexpr int PMK = 4;                        // masks per agent: nearby, candidates, close candidates, contacts
        // One compare + one add-with-carry per test: the compare's lane mask is the carry-in of acc = 2 acc + carry, so
        // after the JQ agents of this split bit (JQ-1-q) of acc is the answer for agent q; reversed and shifted into
        // place (agent j = sx*JQ + q) at the end.
        unsigned a_nb[1], a_cd[1], a_c1[1], a_ht[1];       // N <= 64: this split's partial masks (N > 64 ORs each group into LDS at once)
        bool exc = false;
        auto place = [&](unsigned acc) -> u64 { return (u64)(__brev(acc) >> (32 - JQ)) << (sx * JQ); };
        for (int rep = 0, reps = REPS(2); rep < reps; ++rep) {
            FENCE();
            exc = false;
            // (N > 64: the loop over the 64-agent groups stays rolled -- unrolled, the pair pass alone was 20 KB of a kernel
            // that has to fit a 64 KB instruction cache)
#pragma unroll 1
            for (int w = 0; w < NW; ++w) {
                unsigned nb = 0, cd = 0, c1 = 0, ht = 0, c2 = 0, a_hi = 0;
                const double *qx = spx + w * 64 + sx * JQ, *qy = spy + w * 64 + sx * JQ;
                // (two agents j per 16-byte LDS read at an immediate offset -- as 8-byte reads the compiler pairs x with y in one
                // stride-64 read whose address costs a vector add per pair; the periodic variant is a second copy of the loop,
                // not a branch and a register copy per pair: 601 -> 574 us at 256 x 4096)
                auto pairs = [&](auto per_c) {
                    const bool PER = per_c;                              // a compile-time constant in the two copies of the lattice kernels
                    static_assert(JQ % 2 == 0, "pair pass: two agents per LDS read");
#pragma unroll
                    for (int q = 0; q < JQ; q += 2) {
                        const double2 x2 = *reinterpret_cast<const double2 *>(qx + q), y2 = *reinterpret_cast<const double2 *>(qy + q);
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            double rx = (h ? x2.y : x2.x) - px, ry = (h ? y2.y : y2.x) - py;
                            const double d2u = rx * rx + ry * ry;
                            double d2 = d2u;
                            if (PER) { wrap_rel(rx, ry, P.w_half, P.h_half); d2 = rx * rx + ry * ry; }
                            // nearby (CPP:161), its exception band, contact pairs of the NEXT step (ENV:442-457) on the un-wrapped
                            // distance; candidates (CPP:658) and close candidates on the wrapped one
                            if constexpr (NW > 2)         // with the wider pre-selection ring (N > 128: pays there)
                                pair_tests6(d2u, d2, P.c_near, P.c_near_hi, P.c_ball, P.c_sen, P.c_close, P.c_close2, nb, a_hi, ht, cd, c1, c2);
                            else
                                pair_tests5(d2u, d2, P.c_near, P.c_near_hi, P.c_ball, P.c_sen, P.c_close, nb, a_hi, ht, cd, c1);
                        }
                    }
                };
                if constexpr (LAT) { if (P.periodic) pairs(std::true_type{}); else pairs(std::false_type{}); }
                else pairs(P.periodic != 0);                 // (the generic-scan kernels are beyond the instruction cache as it is: one copy)
                exc = exc || (a_hi != nb);             // some agent is not "nearby" by a hair (see the occupied-cell filter)
                if constexpr (NPAD < 64) { a_nb[0] = nb; a_cd[0] = cd; a_c1[0] = c1; a_ht[0] = ht; }
                else if (rep == reps - 1) {
                    // N >= 64: a split's share of a 64-agent group is JQ = 16 agents -- exactly one 16-bit quarter of the
                    // group's mask word.  Each split stores its quarter (plain 2-byte store, no atomics, nothing to zero);
                    // behind the barrier a reader gets the whole 64-bit word with one load.
                    unsigned short *pq = reinterpret_cast<unsigned short *>(pm) + sx;
                    auto quarter = [](unsigned acc) -> unsigned short { return (unsigned short)(__brev(acc) >> 16); };
                    pq[((size_t)(0 * NW + w) * AG + at) * 4] = quarter(nb);
                    pq[((size_t)(1 * NW + w) * AG + at) * 4] = quarter(cd);
                    pq[((size_t)(2 * NW + w) * AG + at) * 4] = quarter(c1);
                    pq[((size_t)(3 * NW + w) * AG + at) * 4] = quarter(ht);
                    if constexpr (NW > 2) pq[((size_t)(4 * NW + w) * AG + at) * 4] = quarter(c2);
                }
            }
        }
        STAMP(14);
        if (use_lat && exc) atomicOr(reinterpret_cast<unsigned *>(sflag) + (at >> 2), 1u << ((at & 3) * 8));   // resolve the occupied-cell filter of this agent exactly
        if constexpr (NPAD < 64) {
            // N < 64 (several environments per wavefront, JQ < 16): every split stores its partial masks, the readers OR the
            // WPE copies
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                // bit position of agent j in the wave-wide masks = its lane (el*NPAD + j)
                const u64 nbm = place(a_nb[w]);
                pm[((sx * PMK + 0) * NW + w) * AG + at] = NPAD < 64 ? (nbm << (el * NPAD)) : nbm;
                pm[((sx * PMK + 1) * NW + w) * AG + at] = place(a_cd[w]);
                pm[((sx * PMK + 2) * NW + w) * AG + at] = place(a_c1[w]);
                pm[((sx * PMK + 3) * NW + w) * AG + at] = place(a_ht[w]);
            }
            __syncthreads();
            nearbyN[0] = 0; candN[0] = 0; cand1N[0] = 0; cand2N[0] = 0;
#pragma unroll
            for (int q = 0; q < WPE; ++q) nearbyN[0] |= pm[(q * PMK + 0) * AG + at];
            if (sx == SB) {
#pragma unroll
                for (int q = 0; q < WPE; ++q) { candN[0] |= pm[(q * PMK + 1) * AG + at]; cand1N[0] |= pm[(q * PMK + 2) * AG + at]; }
            }
            if (sx == SB) {
#pragma unroll
                for (int q = 0; q < WPE; ++q) hitN[0] |= pm[(q * PMK + 3) * AG + at];
                hitN[0] &= ~(1ull << i);                                       // k != i
            }
        } else {
            __syncthreads();
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                nearbyN[w] = pm[(0 * NW + w) * AG + at]; candN[w] = 0; cand1N[w] = 0; cand2N[w] = 0;
                if (sx == SB) { candN[w] = pm[(1 * NW + w) * AG + at]; cand1N[w] = pm[(2 * NW + w) * AG + at]; cand2N[w] = pm[(4 * NW + w) * AG + at]; }
            }
            if (sx == CS) {
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    hitN[w] = pm[(3 * NW + w) * AG + at];
                    if (w == (i >> 6)) hitN[w] &= ~(1ull << (i & 63));         // k != i
                }
            }
        }
    }
    const u64 nearby1 = nearbyN[0];
    STAMP(15);
    EXIT_AT(2);
    if (NW > 1 && sx == SB) __builtin_amdgcn_s_setprio(2);     // N > 64: B's insertion is a chain of dependent operations over four groups
                                                                // of candidates -- issue it first, the walkers' independent work fills the gaps
    if (sx == SB) for (int rep = 0, reps = REPS(10); rep < reps; ++rep) {
        FENCE();
        bool collision = false;
        double nd[kTopoMax]; int nj[kTopoMax];
#pragma unroll
        for (int k = 0; k < kTopoMax; ++k) { nd[k] = INFINITY; nj[k] = -1; }
        u64 nearby[NW], cand[NW];
        {
            // candidates = agents within d_sen, self removed (CPP:672-676).  When at least `topo` of them are within
            // the smaller radius sqrt(c_close), the topo nearest all come from those (every other one is strictly
            // farther): insert only them -- same list, far fewer loop trips.
            u64 c1[NW]; int n1 = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) { nearby[w] = nearbyN[w]; cand[w] = candN[w]; c1[w] = cand1N[w]; }
            if (i < 64 * NW) { const u64 self = ~(1ull << (i & 63)); cand[NPAD <= 64 ? 0 : (i >> 6)] &= self; c1[NPAD <= 64 ? 0 : (i >> 6)] &= self; }
#pragma unroll
            for (int w = 0; w < NW; ++w) n1 += __popcll(c1[w]);
            const bool few = n1 >= P.topo;
            if constexpr (NW > 2) {             // second, wider ring before falling back to everything within d_sen
                u64 c2[NW]; int n2 = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) c2[w] = cand2N[w];
                if (i < 64 * NW) c2[i >> 6] &= ~(1ull << (i & 63));
#pragma unroll
                for (int w = 0; w < NW; ++w) n2 += __popcll(c2[w]);
                const bool some = n2 >= P.topo;
#pragma unroll
                for (int w = 0; w < NW; ++w) cand[w] = some ? c2[w] : cand[w];
            }
#pragma unroll
            for (int w = 0; w < NW; ++w) cand[w] = few ? c1[w] : cand[w];
        }
        // pass B: the topo nearest candidates in ascending (distance, index) order.  Each candidate's exact fp64 squared
        // distance is turned into a sort key whose lowest 8 mantissa bits carry its index (non-negative doubles order like
        // their bit patterns), and the key ripples through a sorted 7-entry register list with one v_min_f64 + one
        // v_max_f64 per entry -- a third of the instructions of a compare-and-swap chain on (distance, index) pairs.  The
        // keys differ from the squared distances by < 256 ulp, so the order is the reference's (by norm = sqrt, monotonic)
        // unless two of the 7 smallest squared distances agree in all but those bits (a true tie and every pair whose
        // square roots could coincide included): such a lane -- the 7th entry guards the boundary of the list -- redoes its
        // insertion with the exact compare-and-swap chain on the norms below.
        double dmin = INFINITY;
        {
            double nk[kTopoMax + 1];
#pragma unroll
            for (int k = 0; k <= kTopoMax; ++k) nk[k] = INFINITY;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                u64 h = cand[w];
                while (h) {
                    const int jj = __ffsll((unsigned long long)h) - 1;
                    h &= h - 1;
                    double rx = spx[w * 64 + jj] - px, ry = spy[w * 64 + jj] - py;
                    if (P.periodic) wrap_rel(rx, ry, P.w_half, P.h_half);
                    const double cd = rx * rx + ry * ry;
                    dmin = fmin(dmin, cd);
                    double key = __builtin_bit_cast(double, (__builtin_bit_cast(u64, cd) & ~0xFFull) | (u64)(w * 64 + jj));
#pragma unroll
                    for (int k = 0; k <= kTopoMax; ++k) {
                        const double lo = fmin(nk[k], key);
                        key = fmax(nk[k], key);
                        nk[k] = lo;
                    }
                }
            }
            bool amb = false;
#pragma unroll
            for (int k = 0; k < kTopoMax; ++k) {
                const u64 kb = __builtin_bit_cast(u64, nk[k]), kn = __builtin_bit_cast(u64, nk[k + 1]);
                const bool fin = nk[k] < INFINITY;
                nj[k] = fin ? (int)(kb & 0xFFull) : -1;
                amb = amb || (fin && nk[k + 1] < INFINITY && ((kb ^ kn) >> 8) == 0);
            }
            if (__any(amb)) {                     // rare: the exact (distance, index) insertion, ascending j => ties keep the lower index first
                int xj[kTopoMax];
#pragma unroll
                for (int k = 0; k < kTopoMax; ++k) { nd[k] = INFINITY; xj[k] = -1; }
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    u64 h = amb ? cand[w] : 0;
                    while (h) {
                        const int jj = __ffsll((unsigned long long)h) - 1;
                        h &= h - 1;
                        double rx = spx[w * 64 + jj] - px, ry = spy[w * 64 + jj] - py;
                        if (P.periodic) wrap_rel(rx, ry, P.w_half, P.h_half);
                        // the reference sorts by the NORM (CPP:636-641): two squared distances a few ulp apart can round to the
                        // same sqrt and are then a tie (lower index first), so this path compares the norms themselves
                        double cd = sqrt(rx * rx + ry * ry); int cj = w * 64 + jj;
                        bool ins = false;                 // once the candidate is placed the tail only shifts: an entry
#pragma unroll                                            // carried down must pass equal distances (it was ahead of them)
                        for (int k = 0; k < kTopoMax; ++k) {
                            const bool sw = ins || cd < nd[k];
                            ins = sw;
                            const double td = nd[k]; const int tjj = xj[k];
                            nd[k] = sw ? cd : td; xj[k] = sw ? cj : tjj;
                            cd = sw ? td : cd;    cj = sw ? tjj : cj;
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < kTopoMax; ++k) nj[k] = amb ? xj[k] : nj[k];
            }
        }
        STAMP(13);
        if constexpr (NW > 1) {
#pragma unroll
            for (int w = 0; w < NW; ++w) snear[w * AG + at] = nearby[w];
        }
#pragma unroll
        for (int k = 0; k < kTopoMax; ++k) {                      // CPP:459-491 collision test on the NEW list
            const bool used = k < P.topo && nj[k] >= 0;
            snei[at * kNeiStride + k] = (short)(used ? nj[k] : -1);
            if (P.export_small && act && k < P.topo) P.nei[((size_t)e * n_a + i) * P.topo + k] = used ? nj[k] : -1;
        }
        // collision <=> some listed neighbour is closer than r_avoid (CPP:472-487) <=> the NEAREST candidate is (the list is
        // sorted, and the nearest candidate is always listed)
        collision = dmin < P.c_avoid;
        snei[at * kNeiStride + kTopoMax] = (short)(collision ? 1 : 0);
    }
    if (sx == CS) {
        // ---- ball-to-ball contact spring of the NEXT step: ENV:442-457 (_get_dist_b2b) + CPP:735-815 (_sf_b2b_all), on the
        // positions just published.  Entry (i,k) = collide * d_edge * k_ball * (-(delta/d_center)), delta = p_k - p_i (wrapped
        // when periodic), d_center un-wrapped for every pair the reference evaluates (its numpy wrap only touches agent 0's
        // row, which the i>j loop never reads); summed over k in index order (CPP:799-807).  The colliding pairs (centre
        // distance < 2 size_a) are the contact masks of the pair pass.  (Which split: CS above.)
        double sfx = 0.0, sfy = 0.0;
        for (int rep = 0, reps = REPS(1); rep < reps; ++rep) {
            FENCE();
            sfx = 0.0; sfy = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                u64 h = hitN[w];
                while (h) {
                    const int kk = __ffsll((unsigned long long)h) - 1;
                    h &= h - 1;
                    const double dx = spx[w * 64 + kk] - px, dy = spy[w * 64 + kk] - py;
                    const double dc = sqrt(dx * dx + dy * dy);
                    const double de = fabs(dc - P.size2);
                    double wx = dx, wy = dy;
                    if (P.periodic) wrap_rel(wx, wy, P.w_half, P.h_half);
                    const double ux = wx / dc, uy = wy / dc;
                    sfx += 1.0 * de * P.k_ball * (-ux);
                    sfy += 1.0 * de * P.k_ball * (-uy);
                }
            }
        }
        if (act) P.sf_next[(size_t)e * n_a + i] = make_double2(sfx, sfy);
        if (NW > 1) __builtin_amdgcn_s_setprio(0);
    }
    STAMP(3);
    EXIT_AT(3);

    // ---- target-cell scan over this split's words, _get_target_grid_state CPP:858-908: first-minimum
    // nearest cell, sensed-cell bits (d < d_sen), and per cell the ballot of agents with d <= r_avoid/2
    // (CPP:183-186 inverted).
    // fp32 pre-filter: a decision is taken in fp32 only when the fp32 squared distance is outside a guard band
    // around its threshold (band = rigorous bound of the fp32 evaluation error for |coordinates| <= coord_lim);
    // otherwise the word / the argmin is re-evaluated in fp64 exactly as the reference does.
    const float pxf = (float)px, pyf = (float)py;
    const bool lane_far = act && !(fabs(px) <= (double)P.coord_lim && fabs(py) <= (double)P.coord_lim);
    const bool wave_exact = (P.force_exact != 0) || (__any(lane_far) != 0);
    float best32 = INFINITY, second32 = INFINITY; int bc = 0;
    const f2v pxx = {pxf, pxf}, pyy = {pyf, pyf};
    if constexpr (use_lat) {
        // ---- lattice path.  Row b of the lattice holds the cells of columns rowmask[b]; the columns within
        // lattice distance rho of the agent form an interval.  Columns inside the radius shrunk by the margin lat_m
        // are in range for certain, columns outside the radius grown by lat_m are not; the (rare) columns in between
        // are decided by the reference's exact fp64 test on the stored coordinates.  lat_m = 5x the model's error
        // bound: per coordinate 2^-24 |coordinate| (fp32 cast of the fp64 lattice coordinate) + 1e-6 (lattice fit
        // tolerance of detect_lattice) + ~1e-6 (fp32 radius / sqrt roundings), i.e. < 2e-5 steps for |coordinate| <= 128.
        // The sets stay in ROW space: window row t of an agent is lattice row b0 + t, its in-range columns are kept as a
        // 32-bit word relative to the agent's first possible column ca0 (the window is <= 17 columns wide).
        const LatEnv &L = P.lat[es];
        const double apd = (px - L.ox) * L.uxi + (py - L.oy) * L.uyi;
        const double bpd = (px - L.ox) * L.vxi + (py - L.oy) * L.vyi;
        const float apf = (float)apd, bpf = (float)bpd;
        const int nrows = L.nrows, ncols = L.ncols;
        const float lat_m = fmaxf(1e-4f, 8e-7f * fmaxf(fabsf(apf), fabsf(bpf)));
        // first window row / first window column: no row below b0s and no column below ca0 can be in range (the interval
        // of a row starts at ceil(a - h) with h <= R + margin)
        const int b0s = (int)ceilf(bpf - (L.R + 2.0f * lat_m));
        int ca0 = (int)ceilf(apf - (L.R + 2.0f * lat_m));
        ca0 = ca0 < 0 ? 0 : (ca0 > 63 ? 63 : ca0);
        if (sx == 0) {
            float4 hq;
            hq.x = (float)(apd - (double)ca0); hq.y = (float)(bpd - (double)b0s);
            hq.z = __int_as_float(b0s); hq.w = __int_as_float(ca0);
            hdr[at] = hq;
        }
        const u64 *rm = lrm + el * 64;
        const short *rs = lrs + el * 64;
        // Rows are dealt over the splits other than B: B runs the ordered neighbour insertion meanwhile (about one
        // split's share of the walk), so no split is the straggler at the barrier that follows.
        constexpr int WS = WPE > 1 ? WPE - 1 : 1;
        const int wr = WPE > 1 ? (sx < SB ? sx : sx - 1) : 0;
        // The walk is instantiated for 32-bit row masks (every env's lattice has <= 32 columns: the reference's shapes
        // do) and for 64-bit ones.
        auto walk = [&](auto tag) {
        typedef decltype(tag) MT;
        constexpr int MB = (int)sizeof(MT) * 8;
        auto rowmask = [&](int b) -> MT { return MB == 32 ? (MT)reinterpret_cast<const unsigned *>(rm)[2 * b] : (MT)rm[b]; };
        auto popc = [](MT v) -> int { return MB == 32 ? __popc((unsigned)v) : __popcll((unsigned long long)v); };
        auto ffs0 = [](MT v) -> int { return (MB == 32 ? __ffs((unsigned)v) : __ffsll((unsigned long long)v)) - 1; };
        // columns lo..hi inclusive, 0 <= lo, hi <= MB-1; empty when hi < lo
        auto range = [](int lo, int hi) -> MT { return hi >= lo ? (MT)((~(MT)0 >> (MB - 1 - hi)) & (~(MT)0 << lo)) : (MT)0; };
        auto row_sel = [&](int b, float rho, double cut) -> MT {
            // columns of row b within lattice distance rho of (apf, bpf); exact test d2 < cut on the boundary columns
            const bool rowok = act && b >= 0 && b < nrows;
            const float dy = (float)b - bpf, dy2 = dy * dy;
            const float ro = rho + lat_m, ri = rho - lat_m, ri2 = ri > 0.0f ? ri * ri : 0.0f;
            const float ho2 = ro * ro - dy2;
            const bool any_o = rowok && ho2 > 0.0f;
            // raw v_sqrt_f32 (1 ulp): its error is far inside the margin
            const float ho = __builtin_amdgcn_sqrtf(fmaxf(ho2, 0.0f));
            int ao0 = (int)ceilf(apf - ho), ao1 = (int)floorf(apf + ho);     // columns that may be in range
            ao0 = ao0 < 0 ? 0 : ao0; ao1 = ao1 > ncols - 1 ? ncols - 1 : ao1;
            const int bq = b < 0 ? 0 : (b > 63 ? 63 : b);
            const MT rowm = any_o ? rowmask(bq) : (MT)0;
            const int rst = rs[bq];
            // The margin is far below one lattice step (the in-range band of a row is < 0.06 columns wide), so only the two
            // END columns of the interval can sit in the uncertain band; every column between them is in range for certain.
            // An end column is certain when its own model distance is inside the shrunk radius.
            const float e0 = (float)ao0 - apf, e1 = (float)ao1 - apf;
            const bool c0 = fmaf(e0, e0, dy2) < ri2, c1 = fmaf(e1, e1, dy2) < ri2;
            MT acc = range(ao0, ao1);
            MT bnd = (MT)((c0 ? (MT)0 : (MT)1 << (ao0 & (MB - 1))) | (c1 ? (MT)0 : (MT)1 << (ao1 & (MB - 1)))) & acc & rowm;   // boundary columns
            acc &= ~bnd;
            while (__any(bnd != 0)) {
                if (bnd != 0) {
                    const int a = ffs0(bnd);
                    bnd &= bnd - 1;
                    const int c = rst + popc(rowm & (MT)(((MT)1 << a) - 1));
                    const double2 g = cell64(c);                             // the reference's test on the stored cell
                    const double ex = g.x - px, ey = g.y - py;
                    if (ex * ex + ey * ey < cut) acc |= (MT)1 << a;
                }
            }
            return rowm & acc;
        };
        for (int rep = 0, reps = REPS(3); rep < reps; ++rep) {
            FENCE();
            // rows b with |b - bpf| < rho + margin: the first is ceil(bpf - rho - margin), at most floor(2 (rho + margin)) + 1
            // of them (lat_nrs / lat_nrc: that count for the largest rho of any env)
            for (int t = wr; t < P.lat_nrs; t += WS)
                srow[t * AG + at] = (unsigned)(row_sel(b0s + t, L.R, P.c_sen) >> ca0);
            // the same walk with radius r_avoid/2: the env's "covered by any agent" columns, per lattice row
            const int b0c = (int)ceilf(bpf - (L.Rc + 2.0f * lat_m));
            for (int t = wr; t < P.lat_nrc; t += WS) {
                const MT selc = row_sel(b0c + t, L.Rc, P.c_occ);
                const int bq = b0c + t < 0 ? 0 : (b0c + t > 63 ? 63 : b0c + t);         // selc == 0 outside the lattice
                if (selc != 0) {
                    if constexpr (MB == 32) atomicOr(reinterpret_cast<unsigned *>(&covrow[el * 64 + bq]), (unsigned)selc);
                    else atomicOr(reinterpret_cast<unsigned long long *>(&covrow[el * 64 + bq]), (unsigned long long)selc);
                }
            }
            // nearest cell (CPP:858-908) from the lattice too: in row b the nearest cell is the set column closest to
            // the agent's column coordinate, on either side of it -- two candidates per row, rows dealt over the splits.
            // best / runner-up are tracked in lattice units; a runner-up within the model's error of the best sends the
            // lane to the exact scan below (cells further out on the same side of a row are >= 1 step^2 worse than that
            // side's candidate, so they are never a runner-up within tolerance).
            best32 = INFINITY; second32 = INFINITY; bc = 0;
            // split the row at the agent's column: `dn` = set columns left of it, `up` = right of it; the nearest
            // of each side is that side's only possible best or runner-up
            const int ar0 = (int)floorf(apf) + 1;
            const int a_r = ar0 < 0 ? 0 : (ar0 > MB - 1 ? MB - 1 : ar0);
            const MT lowm = (MT)(((MT)1 << a_r) - 1);                        // columns < a_r
            const int nr_hi = EPB == 1 ? nrows : 64;
            // candidates of row b: model distance^2 (lattice units) and cell index of the nearest set column on each side
            auto row_cands = [&](int b, float &d2d, float &d2u, int &c_dn) {
                const MT rowm = (EPB == 1 || b < nrows) ? rowmask(b) : (MT)0;
                const float dy = (float)b - bpf, dy2 = dy * dy;
                const MT up = rowm >> a_r, dn = rowm & lowm;
                const int a_up = a_r + ffs0(up);
                const int a_dn = MB - 1 - (MB == 32 ? __clz((int)dn) : __clzll((long long)dn));
                const float dxu = (float)a_up - apf, dxd = (float)a_dn - apf;
                d2u = (act && up != 0) ? fmaf(dxu, dxu, dy2) : INFINITY;
                d2d = (act && dn != 0) ? fmaf(dxd, dxd, dy2) : INFINITY;
                c_dn = rs[b] + popc(dn) - 1;                                 // the `up` candidate is cell c_dn + 1
            };
            for (int b = wr; b < nr_hi; b += WS) {
                float d2d, d2u; int c_dn;
                row_cands(b, d2d, d2u, c_dn);
                // lower cell index first: on an exact tie the strict compare keeps it (and the tie is re-done exactly)
                second32 = __builtin_amdgcn_fmed3f(best32, second32, d2d);
                bool lt = d2d < best32;
                best32 = lt ? d2d : best32; bc = lt ? c_dn : bc;
                second32 = __builtin_amdgcn_fmed3f(best32, second32, d2u);
                lt = d2u < best32;
                best32 = lt ? d2u : best32; bc = lt ? c_dn + 1 : bc;
            }
            // runner-up within the model's error of the best (|coordinate error| <= 2^-23 max(|a|, |b|) from the fp32
            // cast + 1e-6 lattice fit tolerance, in steps): decide among this split's candidates with the reference's
            // fp64 distances, first minimum in ascending cell order (rows ascending, left candidate before right).
            const float lat_dlt = 1.2e-7f * fmaxf(fabsf(apf), fabsf(bpf)) + 2e-6f;
            const float tol = 6.0f * sqrtf(second32) * lat_dlt + 1e-9f;
            const bool unc_min = act && (wave_exact || (second32 < INFINITY && (second32 - best32) <= tol));
            if (__any(unc_min)) {
                if (unc_min) {
                    const float thr = wave_exact ? INFINITY : best32 + tol;
                    double bestd = INFINITY; int bcd = bc;
                    for (int b = wr; b < nr_hi; b += WS) {
                        float d2d, d2u; int c_dn;
                        row_cands(b, d2d, d2u, c_dn);
                        const bool td = d2d <= thr, tu = d2u <= thr && d2u < INFINITY;
                        if ((td && d2d < INFINITY) || tu) {
                            const double2 g0 = cell64((td && d2d < INFINITY) ? c_dn : 0), g1 = cell64(tu ? c_dn + 1 : 0);
                            const double e0x = g0.x - px, e0y = g0.y - py, e1x = g1.x - px, e1y = g1.y - py;
                            const double q0 = e0x * e0x + e0y * e0y, q1 = e1x * e1x + e1y * e1y;
                            if (td && d2d < INFINITY && q0 < bestd) { bestd = q0; bcd = c_dn; }
                            if (tu && q1 < bestd) { bestd = q1; bcd = c_dn + 1; }
                        }
                    }
                    bc = bcd;
                }
            }
        }
        };
        if (sx != SB || WPE == 1) { if (P.lat_n32) walk(0u); else walk((u64)0); }
    } else
    for (int rep = 0, reps = REPS(3); rep < reps; ++rep) {
    FENCE();
    best32 = INFINITY; second32 = INFINITY; bc = 0;
    for (int w = 0; w < W; ++w) {
        if (!mine(w)) continue;
        unsigned word = 0, rword = 0;
        int mlo = 0, mhi = 0;                 // lanes 0..31: ballot (lo, hi halves) of cell b = lane
        int bl = 0; const float best_in = best32;
        unsigned orw = 0;                     // NW == 1: reversed occupied-bit accumulator
        const unsigned nlo = (unsigned)nearby1, nhi = (unsigned)(nearby1 >> 32);
        u64 unc = 0;
        const float4 *cq = reinterpret_cast<const float4 *>(cq_e + w * 64);
        static_for<16>([&](auto prc) {
            constexpr int pr = decltype(prc)::value;
            const float4 q = cq[pr];                              // {xa, xb, ya, yb}
            const f2v gx = {q.x, q.y}, gy = {q.z, q.w};
            const f2v rx = gx - pxx, ry = gy - pyy;
            const f2v d2v = __builtin_elementwise_fma(rx, rx, ry * ry);   // packed: two cells per instruction
            static_for<2>([&](auto hc) {
                constexpr int hh = decltype(hc)::value;
                constexpr int b = 2 * pr + hh;
                const float d2 = hh ? d2v.y : d2v.x;
                second32 = __builtin_amdgcn_fmed3f(best32, second32, d2);
                const bool lt = d2 < best32;                     // strict: ties keep the earlier cell
                best32 = lt ? d2 : best32;
                bl = lt ? b : bl;                                // word-local index: an inline constant
                const u64 m_slo = __ballot(d2 < P.csen_lo), m_shi = __ballot(d2 < P.csen_hi);
                const u64 m_olo = __ballot(d2 < P.cocc_lo), m_ohi = __ballot(d2 < P.cocc_hi);
                rword = shl1_or_mask(rword, m_slo);
                if constexpr (NW == 1) {
                    // occupied for lane i  <=>  some NEARBY agent is within r_avoid/2 of this cell (CPP:161,185)
                    const unsigned t = ((unsigned)m_olo & nlo) | ((unsigned)(m_olo >> 32) & nhi);
                    orw = shl1_or_mask(orw, __ballot(t != 0));
                } else {
                    mlo = writelane_c<b>((int)(unsigned)m_olo, mlo);
                    mhi = writelane_c<b>((int)(unsigned)(m_olo >> 32), mhi);
                }
                unc |= (m_slo ^ m_shi) | (m_olo ^ m_ohi);
            });
        });
        word = __brev(rword);
        unsigned oword = __brev(orw);
        if (best32 < best_in) bc = w * 32 + bl;
        u64 mym = ((u64)(unsigned)mhi << 32) | (unsigned)mlo;
        if (unc != 0 || wave_exact) {                     // rare: redo this word exactly
            word = 0; oword = 0;
#pragma unroll 4
            for (int b = 0; b < 32; ++b) {
                const double2 g = cell64(w * 32 + b);
                const double rx = g.x - px, ry = g.y - py;
                const double d2 = rx * rx + ry * ry;
                if (d2 < P.c_sen) word |= 1u << b;
                const u64 m = __ballot(d2 < P.c_occ);
                if (lane == b) mym = m;
                if (NW == 1 && ((((unsigned)m & nlo) | ((unsigned)(m >> 32) & nhi)) != 0)) oword |= 1u << b;
            }
        }
        sbits[w * AG + at] = word;
        if constexpr (NW == 1) owords[w * AG + at] = oword;
        else if (lane < 32) cmask[(size_t)(w * 32 + lane) * NW + aw] = mym;
    }
    }
    if (!use_lat) {   // nearest cell of this split: unambiguous in fp32 unless the runner-up is within tolerance
        const float tol = P.min_tol_a * sqrtf(second32) + P.min_tol_b * second32 + 1e-9f;
        const bool unc_min = act && (wave_exact || (second32 < INFINITY && (second32 - best32) <= tol));
        if (__any(unc_min)) {
            const float thr = unc_min ? (wave_exact ? INFINITY : best32 + tol) : -1.0f;
            double bestd = INFINITY; int bcd = bc;
            for (int w = 0; w < W; ++w) {
                if (!mine(w)) continue;
                for (int b = 0; b < 32; ++b) {
                    const int cc = w * 32 + b;
                    const float rx = cq_e[(cc >> 1) * 4 + (cc & 1)] - pxf, ry = cq_e[(cc >> 1) * 4 + 2 + (cc & 1)] - pyf;
                    if (fmaf(rx, rx, ry * ry) <= thr || (wave_exact && unc_min)) {
                        const double2 g = cell64(w * 32 + b);
                        const double ex = g.x - px, ey = g.y - py;
                        const double d2 = ex * ex + ey * ey;
                        if (d2 < bestd) { bestd = d2; bcd = w * 32 + b; }     // ascending c: first minimum
                    }
                }
            }
            if (unc_min) bc = bcd;
        }
    }
    part_c[sx * AG + at] = (pc_t)bc;
    double *part_d = reinterpret_cast<double *>(smem + P.off_partd);     // [WPE - 1][AG] (lattice launches; the list phase's `perm` reuses it)
    if constexpr (LAT) {
        // each walking split evaluates the exact distance of ITS candidate here, before the barrier (the gather overlaps the
        // other waves' walk); the merge behind the barrier then compares values that sit in LDS instead of every split
        // gathering every candidate
        if (sx != SB || WPE == 1) {
            const double2 g = cell64(bc);
            const double ex = g.x - px, ey = g.y - py;
            part_d[(WPE > 1 && sx > SB ? sx - 1 : sx) * AG + at] = ex * ex + ey * ey;      // [walking split][AG]
        }
    }
    STAMP(16);
    EXIT_AT(4);
    __syncthreads();
    STAMP(17);
    // merge the splits' candidates exactly: (d2 in fp64, cell index) lexicographic minimum = first minimum
    double best = INFINITY; bc = 0;
    for (int rep = 0, reps = REPS(12); rep < reps; ++rep) {
    FENCE();
    best = INFINITY; bc = 0;
#pragma unroll
    for (int s = 0; s < WPE; ++s) {
        if (LAT && WPE > 1 && s == SB) continue;                     // split B does not walk
        const int c = part_c[s * AG + at];
        double d;
        if constexpr (LAT) d = part_d[(WPE > 1 && s > SB ? s - 1 : s) * AG + at];
        else { const double2 g = cell64(c); const double ex = g.x - px, ey = g.y - py; d = ex * ex + ey * ey; }
        if (d < best || (d == best && c < bc)) { best = d; bc = c; }
    }
    }
    const bool in_shape = act && best < P.c_in[es];                    // CPP:889
    if (sx == 0) {
        sncf[at] = bc | (in_shape ? (1 << 30) : 0);
        if (P.export_small && act) {
            P.near_cell[(size_t)e * n_a + i] = bc;
            P.in_flag[(size_t)e * n_a + i] = in_shape ? 1 : 0;
        }
    }
    STAMP(4);
    EXIT_AT(5);

    // ---- observation rows, CPP:102-137,274-306: the rows of this workgroup's environments are contiguous in HBM
    const int PPR = P.obs_dim >> 1;                  // pairs per row
    const int HP = 2 * (P.with_self + P.topo) + 2;   // head pairs: agent block + target pos/vel
    const int envs_here = (P.n_env - blockIdx.x * EPB) < EPB ? (P.n_env - blockIdx.x * EPB) : EPB;
    const int rows = envs_here * n_a;
    OT2 *out = reinterpret_cast<OT2 *>(obs) + (size_t)blockIdx.x * EPB * n_a * PPR;

    // ---- prior policy, calculateActionPrior / robotPolicy CPP:1061-1196: a function of the positions, velocities,
    // neighbour list and nearest cell of THIS observation -- exactly what the reference feeds it at the start of the
    // next step (ENV:605-624: pre-integration state, previous neighbor_index).  Split B evaluates it here, where all of
    // that sits in LDS, and leaves it in HBM for the next launch; the other splits write its share of the head pairs.
    auto prior_policy = [&]() {
    if (sx == SB && P.with_prior) {
        double qx = 0.0, qy = 0.0;
        double lx = 0.0, ly = 0.0;                       // agent_strategy 'llm': the Python twin (ENV:892-940), other repulsion gain
        {
            const int ncf = sncf[at];
            // target: own position when in shape (CPP:889-897) => zero attraction; else the nearest cell
            double tx = px - px, ty = py - py;
            if (!(ncf >> 30)) { const double2 g = cell64(ncf & 0xFFFF); tx = g.x - px; ty = g.y - py; }
            const double dt_ = sqrt(tx * tx + ty * ty);
            if (dt_ > 0) { qx += P.pk_att * tx / dt_; qy += P.pk_att * ty / dt_; }
            lx = qx; ly = qy;
            double avx = 0.0, avy = 0.0; int cnt = 0;
#pragma unroll
            for (int k = 0; k < kTopoMax; ++k) {
                const int j = snei[at * kNeiStride + k];
                const bool used = j >= 0;
                const int tj = el * NPAD + (used ? j : 0);
                const double x = px - sp[tj], y = py - sp[AG + tj];
                const double d2n = x * x + y * y;
                if (used && d2n > 0 && d2n < P.c_avoid) {            // 0 < d < r_avoid (CPP:1150-1160), d = sqrt(d2n): sqrt is monotonic
                    const double d = sqrt(d2n);
                    const double ux = x / d, uy = y / d;
                    const double factor = P.pk_rep * (P.r_avoid / d - 1.0);
                    qx += factor * ux; qy += factor * uy;
                    if (P.llm) { const double fl = P.pk_llm * (P.r_avoid / d - 1.0); lx += fl * ux; ly += fl * uy; }
                }
                if (used) { avx += sp[2 * AG + tj]; avy += sp[3 * AG + tj]; ++cnt; }
            }
            if (cnt > 0) {
                avx /= cnt; avy /= cnt;
                const double sxv = P.pk_ali * (avx - sp[2 * AG + at]), syv = P.pk_ali * (avy - sp[3 * AG + at]);
                qx += sxv; qy += syv; lx += sxv; ly += syv;
            }
        }
        if (act) {
            OT2 o; o.x = to_out<OT>(clamp_ref(qx, -1.0, 1.0)); o.y = to_out<OT>(clamp_ref(qy, -1.0, 1.0));
            reinterpret_cast<OT2 *>(P.prior_next)[(size_t)e * n_a + i] = o;
            if (P.llm) { double2 u; u.x = clamp_ref(lx, -1.0, 1.0); u.y = clamp_ref(ly, -1.0, 1.0); P.act_next[(size_t)e * n_a + i] = u; }
        }
    }
    };

    auto head_blocks = [&](bool early) {
        for (int rep = 0, reps = REPS(7); rep < reps; ++rep) {
            FENCE();
            // The head of a row is NB = self + topo + 1 BLOCKS of four values {x, y, vx, vy}: [own state], the neighbours,
            // the target.  Lane = (row, block): one 16-byte store per lane, eight lanes cover a row's 128-byte head.  Every
            // value is (minuend - subtrahend): own block own - 0; neighbour k: neighbour - own (zeros without a neighbour,
            // CPP:79-81; the relative position wrapped when periodic, CPP:79); target: in shape own - own, else cell - own for
            // the position and 0 - own for the velocity (CPP:136-137).
            typedef OT OT4 __attribute__((ext_vector_type(4)));
            const int NB = P.with_self + P.topo + 1;
            // Who writes which blocks.  early == false (generic path): every split but B (busy with the prior policy) takes
            // items htid, htid + HT, ...  early == true (lattice path): the pass runs BEFORE the barrier that ends the list
            // phase -- it needs nothing from it -- on the two splits with the shortest lists there (A: five chunks of 64 items
            // out of eight, C: three), so that the split with the longest lists is not waited for twice.
            const int total = rows * NB;
            const bool b_out = WPE > 1 && P.with_prior != 0;
            const int HT = b_out ? T - AG : T;
            const int ps = tid / AG, ps_b = (SB + WPE - (int)(blockIdx.x % WPE)) % WPE;     // physical split index; split B's
            const int htid = b_out ? (ps < ps_b ? ps : ps - 1) * AG + at : tid;
            const int n_it = early ? (total + 63) >> 6 : (total + HT - 1) / HT;
            for (int it_ = 0; it_ < n_it; ++it_) {
                int item;
                if (early) {
                    const int c8 = it_ & 7, owner = c8 < 5 ? 0 : 2;
                    const int k = owner == 0 ? 5 * (it_ >> 3) + c8 : 3 * (it_ >> 3) + c8 - 5;      // the owner's k-th chunk
                    if (sx != owner || (k % NW) != aw) continue;
                    item = it_ * 64 + lane;
                } else {
                    if (b_out && sx == SB) break;
                    item = htid + it_ * HT;
                }
                if (item < total) {
                    const int r = NB == 8 ? item >> 3 : item / NB, blk = item - r * NB;
                    const int elr = EPB > 1 ? r / n_a : 0;
                    const int tr = elr * NPAD + (r - elr * n_a);
                    const double ox = sp[tr], oy = sp[AG + tr], ou = sp[2 * AG + tr], ov = sp[3 * AG + tr];
                    // straight-line code (every lane reads a neighbour slot, the nearest-cell word and that cell; the block kind
                    // only selects): a branch per kind would put each dependent LDS read / gather behind its own wait
                    const bool is_tgt = blk == NB - 1, is_nei = !is_tgt && !(P.with_self && blk == 0);
                    const int nslot = blk - P.with_self;
                    const int j = snei[tr * kNeiStride + (is_nei ? nslot : 0)];
                    const int ncf = sncf[tr];
                    const double2 g = P.cells_xy[(size_t)(blockIdx.x * EPB + elr) * P.ng_max + (ncf & 0xFFFF)];
                    const int tj = elr * NPAD + (j < 0 ? 0 : j);
                    const double nx = sp[tj], ny = sp[AG + tj], nu = sp[2 * AG + tj], nv = sp[3 * AG + tj];
                    const bool has = is_nei && j >= 0, out_t = is_tgt && !(ncf >> 30);
                    // minuend: neighbour | cell (target, outside the shape: velocity 0) | own; subtrahend: own | 0
                    const bool zero_m = is_nei && !has;
                    double mx = has ? nx : (out_t ? g.x : ox), my = has ? ny : (out_t ? g.y : oy);
                    double mu = has ? nu : (out_t ? 0.0 : ou), mv = has ? nv : (out_t ? 0.0 : ov);
                    if (zero_m) { mx = 0.0; my = 0.0; mu = 0.0; mv = 0.0; }
                    const bool sub_own = has || is_tgt;
                    const double qx = sub_own ? ox : 0.0, qy = sub_own ? oy : 0.0, qu = sub_own ? ou : 0.0, qv = sub_own ? ov : 0.0;
                    double a = mx - qx, b = my - qy;
                    if (P.periodic && is_nei) wrap_rel(a, b, P.w_half, P.h_half);
                    OT2 *dst = out + (size_t)r * PPR + 2 * blk;
                    if ((PPR & 1) == 0) {                                // rows are a whole number of blocks: 4-value stores stay aligned
                        const OT4 o = {to_out<OT>(a), to_out<OT>(b), to_out<OT>(mu - qu), to_out<OT>(mv - qv)};
                        __builtin_nontemporal_store(o, reinterpret_cast<OT4 *>(dst));
                    } else {                                             // odd list length (non-reference configs)
                        OT2 o0, o1;
                        o0.x = to_out<OT>(a); o0.y = to_out<OT>(b); o1.x = to_out<OT>(mu - qu); o1.y = to_out<OT>(mv - qv);
                        store_nt(dst, o0); store_nt(dst + 1, o1);
                    }
                }
            }
        }
    };

    // Exact exploration-reward verdict of ONE agent thread (CPP:529-549), wave-cooperative and called by the whole wave
    // with wave-uniform arguments: the 64 lanes evaluate one list slot each in fp64 -- psi needs a sqrt and a 12-term
    // cosine -- and the three sums then run over the lanes' values sequentially in slot order (v_readlane broadcasts), which
    // is the reference's order of additions.  Rare (the fp32 verdict decides outside its guard band); a lane looping alone
    // over its list made its workgroup a straggler.  La: agent thread, nL: its list length, eL: its environment.
    auto exact_uniform = [&](int La, int nL, int eL) -> bool {
        const double inv_dsen = 1.0 / P.d_sen;
        const double2 *gcl = P.cells_xy + (size_t)eL * P.ng_max;
        const double pxl = sp[La], pyl = sp[AG + La];
        const short *rowl = sidx + (size_t)La * P.g_stride;
        auto bcast = [](double v, int t) -> double {
            const u64 b = __builtin_bit_cast(u64, v);
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, t);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), t);
            return __builtin_bit_cast(double, ((u64)hi << 32) | lo);
        };
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        for (int base = 0; base < nL; base += 64) {
            const int q = base + lane;
            double t0 = 0.0, t1 = 0.0, t2 = 0.0;
            if (q < nL) {
                const int cc = rowl[q];
                const double2 gq = gcl[cc];
                const double x = gq.x - pxl, y = gq.y - pyl;
                const double z = sqrt(x * x + y * y);
                // _rho_cos_dec(z, 0, d_sen), CPP:1012-1020; z < d_sen holds for every sensed cell
                const double psi = z < P.d_sen ? 0.5 * (1.0 + cospi01(z * inv_dsen)) : 0.0;
                t0 = psi * x; t1 = psi * y; t2 = psi;
            }
            const int m = nL - base < 64 ? nL - base : 64;
            for (int t = 0; t < m; ++t) { a0 += bcast(t0, t); a1 += bcast(t1, t); a2 += bcast(t2, t); }
        }
        if (a2 == 0) a2 = 1E-8;
        const double v0 = 1.0 * a0 / a2, v1 = 1.0 * a1 / a2;
        return sqrt(v0 * v0 + v1 * v1) < 0.05;
    };

    // The G sensed-cell pairs of the observation rows (CPP:274-291): value = stored cell - own position in f64, rounded once.
    // own == false (generic path, behind the workgroup barrier that completes every list): the waves deal the rows.
    // own == true (lattice path): every wave writes the rows of the sixteen agents whose lists IT has just emitted (pw: its
    // agent permutation) -- all four lanes of an agent sit in one wave, so no workgroup barrier separates list and rows.
    auto sensed_rows = [&](bool own, const unsigned char *pw) {
        for (int rep = 0, reps = REPS(8); rep < reps; ++rep) {
            FENCE();
            const int Gp = P.g_max;
            const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = T >> 6;
            // row slot k of this wave -> agent thread tr, environment in the workgroup elr, output row r, "row exists"
            auto locate = [&](int k, int &tr, int &elr, int &r) -> bool {
                if (own) {
                    const int ta = (at & ~63) + pw[(64 - ACTW) + sx * AGW + k];
                    const int ii = NPAD < 64 ? (ta & 63) % NPAD : ta;
                    elr = NPAD < 64 ? (ta & 63) / NPAD : 0;
                    tr = ta; r = elr * n_a + ii;
                    return (blockIdx.x * EPB + elr) < P.n_env && ii < n_a;
                }
                r = k;
                const bool ok = r < rows;
                const int rq = ok ? r : rows - 1;
                elr = EPB > 1 ? rq / n_a : 0;
                tr = elr * NPAD + (rq - elr * n_a);
                return ok;
            };
            if (Gp == 80 && sizeof(OT) <= 4) {
                // the reference's list length: TWO slots per lane, so every store instruction writes 64 x 16 B (f32) of
                // consecutive addresses -- half the store instructions of the pair-per-lane form.  A row has 40 two-slot
                // chunks; 8 rows = 320 chunks = 5 full passes.
                typedef OT OT4 __attribute__((ext_vector_type(4)));
                const int ngrp = own ? AGW / 8 : (rows + nwv * 8 - 1) / (nwv * 8);
                for (int g8 = 0; g8 < ngrp; ++g8) {
#pragma unroll
                    for (int ps = 0; ps < 5; ++ps) {
                        const int ch = ps * 64 + lane;                   // 0..319
                        const int rl = (ch * 205) >> 13;                 // ch / 40 (exact for ch < 320)
                        const int m = ch - rl * 40;
                        // straight-line: both gathers of every pass are issued unconditionally (an empty slot reads cell 0
                        // and stores zeros), so the five passes' loads are in flight together instead of one wait each
                        int tr, elr, r;
                        const bool ok = locate(own ? g8 * 8 + rl : (wv + g8 * nwv) * 8 + rl, tr, elr, r);
                        const double2 *gr = P.cells_xy + (size_t)(blockIdx.x * EPB + elr) * P.ng_max;
                        const double qx = sp[tr], qy = sp[AG + tr];
                        const int cc = *reinterpret_cast<const int *>(sidx + (size_t)tr * P.g_stride + 2 * m);
                        const int c0 = (int)(short)(cc & 0xFFFF), c1 = cc >> 16;
                        const double2 g0 = gr[c0 < 0 ? 0 : c0], g1 = gr[c1 < 0 ? 0 : c1];
                        const OT z = to_out<OT>(0.0);
                        const OT a0 = to_out<OT>(g0.x - qx), b0 = to_out<OT>(g0.y - qy);
                        const OT a1 = to_out<OT>(g1.x - qx), b1 = to_out<OT>(g1.y - qy);
                        OT4 o = {c0 >= 0 ? a0 : z, c0 >= 0 ? b0 : z, c1 >= 0 ? a1 : z, c1 >= 0 ? b1 : z};
                        if (ok) __builtin_nontemporal_store(o, reinterpret_cast<OT4 *>(out + (size_t)r * PPR + HP + 2 * m));
                    }
                }
            } else {
                // any other list length / f64 rows: wave per row, lane = slot
                const int nrow = own ? AGW : (rows - wv + nwv - 1) / nwv;
                for (int k = 0; k < nrow; ++k) {
                    int tr, elr, r;
                    const bool ok = locate(own ? k : wv + k * nwv, tr, elr, r);
                    const double2 *gr = P.cells_xy + (size_t)(blockIdx.x * EPB + elr) * P.ng_max;
                    const double qx = sp[tr], qy = sp[AG + tr];
                    const short *srw = sidx + (size_t)tr * P.g_stride;
                    for (int q = lane; q < Gp; q += 64) {
                        const int c = srw[q];
                        const double2 g = gr[c < 0 ? 0 : c];
                        OT2 o; o.x = c >= 0 ? to_out<OT>(g.x - qx) : to_out<OT>(0.0); o.y = c >= 0 ? to_out<OT>(g.y - qy) : to_out<OT>(0.0);
                        if (ok) store_nt(&out[(size_t)r * PPR + HP + q], o);
                    }
                }
            }
        }
    };

    const int G = P.g_max;
    int n_sel = 0;                                   // length of this agent thread's capped list
    bool uniform = false, unsure = false;            // split A: fp32 verdict of the exploration reward / "redo it exactly"
    if constexpr (LAT) {
    // ================================================================================================================
    // Lattice path, row space.  After the walk every (window row t, agent) holds its sensed columns as a 32-bit word
    // relative to the agent's column ca0, and covrow[b] holds the env's covered columns of lattice row b.
    // (K) kept = in_shape ? sensed & ~covered : sensed (CPP:150,209-216), the row's first cell index and its kept count,
    //     rows dealt over the splits;  (S) the agent threads of each 64-group are ranked by list length;  (E) list
    //     emission + the fp32 reward sums in ONE walk over the kept bits, four lanes per agent (rank ranges), sixteen agents
    //     of SIMILAR list length per wave -- a wave loops as long as its longest range, so grouping by length is what turns
    //     the per-wave trip count from max(n) / 4 into (roughly) the quartile's n / 4.
    // ================================================================================================================
    const float4 hq = hdr[at];
    const int hb0 = __float_as_int(hq.z), hca0 = __float_as_int(hq.w);
    const LatEnv &L = P.lat[es];
    const u64 *rm = lrm + el * 64;
    const short *rs = lrs + el * 64;
    const bool flg = flag_get();                     // a pair within 1e-9 of the "nearby" threshold: exact occupied test
    for (int rep = 0, reps = REPS(4); rep < reps; ++rep) {
        FENCE();
        for (int t = sx; t < P.lat_nrs; t += WPE) {
            const int b = hb0 + t;
            const int bq = b < 0 ? 0 : (b > 63 ? 63 : b);
            const unsigned sel = srow[t * AG + at] & 0x1FFFFu;           // empty outside the lattice / for inactive lanes (the mask: REPS reruns)
            const u64 rowm = rm[bq];
            const int base = rs[bq] + __popcll(rowm & ((1ull << hca0) - 1ull));   // cell index of the first set column >= ca0
            unsigned kept = sel;
            if (in_shape) {
                // occupied <=> within r_avoid/2 of ANY agent: the covering agent of a SENSED cell is "nearby" (CPP:161) by
                // the triangle inequality, except when its distance sits within rounding of the nearby threshold -- those
                // lanes were flagged by the pair pass and are resolved with the reference's nearby x cell test.
                const unsigned cv = (unsigned)(covrow[el * 64 + bq] >> hca0);
                kept = sel & ~cv;
                if (flg) {
                    const unsigned relm = (unsigned)(rowm >> hca0);
                    unsigned it = sel & cv;
                    kept = sel;
                    while (it) {
                        const int c = __ffs(it) - 1; it &= it - 1;
                        const double2 g = cell64(base + __popc(relm & ((1u << c) - 1u)));
                        bool occ = false;
                        for (int q = 0; q < NW && !occ; ++q) {
                            u64 nbm = NW == 1 ? (nearby1 >> (NPAD < 64 ? el * NPAD : 0)) : snear[q * AG + at];
                            while (nbm && !occ) {
                                const int j = q * 64 + __ffsll((unsigned long long)nbm) - 1; nbm &= nbm - 1;
                                const double ex = g.x - spx[j], ey = g.y - spy[j];
                                occ = ex * ex + ey * ey < P.c_occ;
                            }
                        }
                        if (occ) kept &= ~(1u << c);
                    }
                }
            }
            if (rep == reps - 1) srow[t * AG + at] = kept | ((unsigned)base << 17);     // <= 17 window columns, cell index < 2^15
            pcr[at * NRC + t] = (unsigned char)__popc(kept);
            if (P.export_idx) orow[t * AG + at] = sel & ~kept;
        }
    }
    // unused list slots read -1 (CPP:46-50): the whole list region is filled here, the emission overwrites its slots
    {
        uint4 *fill = reinterpret_cast<uint4 *>(sidx);
        const int n16 = (AG * P.g_stride * 2) >> 4;
        for (int q = tid; q < n16; q += T) fill[q] = uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    }
    __syncthreads();
    STAMP(5);
    EXIT_AT(6);

    // ---- (S) rank of every agent thread by (list length, index) inside its 64-group: each split compares against a quarter
    // of the group (the other agents' keys come from the wave's own lanes), the partial counts are summed through LDS
    pc_t *prk = part_c;                                  // [WPE][AG] (the nearest-cell candidates are consumed)
    int n_kept;
    {
        const uint4 cq = reinterpret_cast<const uint4 *>(pcr)[at];
        n_kept = (int)__builtin_amdgcn_sad_u8(cq.x, 0u, __builtin_amdgcn_sad_u8(cq.y, 0u, __builtin_amdgcn_sad_u8(cq.z, 0u, __builtin_amdgcn_sad_u8(cq.w, 0u, 0u))));
        const int key = (lane < ACTW) ? n_kept * 64 + lane : lane - 64;      // empty agent threads first, in lane order
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) cnt += (__builtin_amdgcn_readlane(key, sx * 16 + q) < key) ? 1 : 0;
        prk[sx * AG + at] = (pc_t)cnt;
    }
    n_sel = n_kept > G ? G : n_kept;
    __syncthreads();
    STAMP(18);
    EXIT_AT(7);
    int ea;                                              // the agent thread this lane works for in (E)
    {
        int rank = 0;
#pragma unroll
        for (int s = 0; s < WPE; ++s) rank += prk[s * AG + at];
        unsigned char *pw = perm + (tid >> 6) * 64;      // this wave's own copy: no workgroup barrier needed
        pw[rank] = (unsigned char)lane;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // LPA lanes per agent, AGW agents per wave: split 0 takes the shortest lists, split 3 the longest
        ea = (at & ~63) + pw[(64 - ACTW) + sx * AGW + lane / LPA];
    }
    // No workgroup barrier follows: the workgroup ends with its slowest wave.  Issue priority by the work that is left -- split D
    // took the longest lists, C the next, B has the prior policy to run (D 3 / C 2 / B 1 / A 0).  N = 256, one workgroup per CU
    // and nothing else to fill a tail: 566 -> 550 us at 256 x 4096 (against 2/0/1: 558, 3/1/2: 555, 2/0/2: 561).  With seven
    // workgroups per CU at N = 64 it still pays: 83.9 -> 83.3 us; 32 x 1024 (one workgroup generation): 25.0 -> 24.35 us.
    { if (sx == 3) __builtin_amdgcn_s_setprio(3); else if (sx == 2) __builtin_amdgcn_s_setprio(2); else if (sx == SB) __builtin_amdgcn_s_setprio(1); }

    // ---- (E) capped sensed list (CPP:236-271) + exploration-reward sums (CPP:494-551), fp32 fast path.  The kept list of
    // an agent is cut into four contiguous RANK ranges, one per lane of its quad; each lane walks its range bit by bit
    // through the window rows.  Per kept cell: its list slot, its cell index (row base + set columns of the row mask
    // below it) and its reward weight psi(u), u = (distance / d_sen)^2 from the LATTICE coordinates -- |v| is invariant
    // under the lattice's rotation, so the sums run in lattice steps and need neither the stored cells nor a list
    // read-back.  The fp32 result only DECIDES outside a guard band; inside it the sums are redone in fp64 below.
    float qn0 = 0.0f, qn1 = 0.0f, qdn = 0.0f, qrl = 1.0f; int qnk = 0;      // the quad's reward sums / list length / d_sen in steps
    for (int rep = 0, reps = REPS(11); rep < reps; ++rep) {
        FENCE();
        const int sub = lane & (LPA - 1);
        const float4 ha = hdr[ea];
        const float apr = ha.x, bpr = ha.y;
        const int ab0 = __float_as_int(ha.z), aca0 = __float_as_int(ha.w);
        const int ael = NPAD < 64 ? (ea & 63) / NPAD : 0;
        const LatEnv &La = P.lat[(blockIdx.x * EPB + ael) < P.n_env ? (blockIdx.x * EPB + ael) : P.n_env - 1];
        const float Rl = La.R;                                   // d_sen in lattice steps
        // the walk works in units of d_sen (coordinates scaled by 1 / R): u = da^2 + db^2 needs no further factor, and the
        // |v| threshold becomes the constant 0.05 / d_sen.  (1 / R to 1 ulp: inside the guard band's allowance for psi)
        const float inv_r = __builtin_amdgcn_rcpf(Rl);
        const float apr_s = apr * inv_r;
        const u64 *rma = lrm + ael * 64;
        const uint4 cq = reinterpret_cast<const uint4 *>(pcr)[ea];
        const int s0 = (int)__builtin_amdgcn_sad_u8(cq.x, 0u, 0u), s1 = (int)__builtin_amdgcn_sad_u8(cq.y, 0u, 0u);
        const int s2 = (int)__builtin_amdgcn_sad_u8(cq.z, 0u, 0u), s3 = (int)__builtin_amdgcn_sad_u8(cq.w, 0u, 0u);
        const int nk = s0 + s1 + s2 + s3;
        const int per = (nk + LPA - 1) / LPA;
        const int k0 = sub * per < nk ? sub * per : nk;
        const int k1 = k0 + per < nk ? k0 + per : nk;
        // the window row holding rank k0: the LAST row whose first rank is <= k0 (empty rows in front of it are skipped)
        int t, before;
        {
            const int c1 = s0, c2 = s0 + s1, c3 = c2 + s2;
            const int d = (k0 >= c1 ? 1 : 0) + (k0 >= c2 ? 1 : 0) + (k0 >= c3 ? 1 : 0);
            before = d == 0 ? 0 : (d == 1 ? c1 : (d == 2 ? c2 : c3));
            const unsigned wd = d == 0 ? cq.x : (d == 1 ? cq.y : (d == 2 ? cq.z : cq.w));
            const int e1 = before + (int)(wd & 255u), e2 = e1 + (int)((wd >> 8) & 255u), e3 = e2 + (int)((wd >> 16) & 255u);
            const int j = (k0 >= e1 ? 1 : 0) + (k0 >= e2 ? 1 : 0) + (k0 >= e3 ? 1 : 0);
            before = j == 0 ? before : (j == 1 ? e1 : (j == 2 ? e2 : e3));
            t = 4 * d + j;
        }
        const unsigned *srp = srow + t * AG + ea;                    // kept columns (17 bits) | first cell index of the row << 17
        const unsigned wd0 = *srp;
        unsigned kept = k0 < k1 ? (wd0 & 0x1FFFFu) : 0u;
        {   // drop the (k0 - before) lowest set bits: position of that set bit by a binary search on popcounts
            int pb = 0, left = k0 - before;
#pragma unroll
            for (int sh = 16; sh >= 1; sh >>= 1) {
                const int c = __popc((kept >> pb) & ((1u << sh) - 1u));
                const bool go = left >= c;
                left -= go ? c : 0; pb += go ? sh : 0;
            }
            kept &= ~((1u << pb) - 1u);                              // pb <= 31: that set bit exists (k0 < k1)
        }
        int base = (int)(wd0 >> 17);
        // a lane with k0 < k1 only ever visits rows between two non-empty window rows -- rows of the lattice, 0 <= b < 64 --
        // so the walk needs no range check on the row; the first read of a lane without work is clamped
        const int bt0 = ab0 + t;
        const u64 *rmp = rma + (bt0 < 0 ? 0 : (bt0 > 63 ? 63 : bt0));
        unsigned relm = (unsigned)(*rmp >> aca0);
        float dbf = ((float)t - bpr) * inv_r, db2 = dbf * dbf;
        float n0 = 0.0f, n1 = 0.0f, dn = 0.0f;
        short *row = sidx + (size_t)ea * P.g_stride;
        const unsigned nm1 = (unsigned)(nk - 1);
        // rank(q) = round(q (n-1)/(G-1)) of list slot q (CPP:241-245), see the generic path for the integer form
        auto rank_of = [&](int q) -> int {
            if (P.cap_int) {
                const unsigned x = 2u * (unsigned)q * nm1 + (unsigned)(G - 1);
                const unsigned hq_ = __umulhi(x, P.cap_magic);
                return (int)((((x - hq_) >> 1) + hq_) >> P.cap_shift);
            }
            return (int)round(q * ((double)nm1 / (G - 1)));
        };
        auto walk_range = [&](auto capped) {
            constexpr bool CAP = decltype(capped)::value;
            int k = k0;
            // capped agent (n > G, rare): `slot` = number of selected ranks below k, `nxt` = the next selected rank
            int slot = k0, nxt = 0;
            if constexpr (CAP) {
                if (nk > G) {
                    // rank(q) < k  <=>  2 q (n-1) < (2k-1)(G-1) (integer form of the rounding, G-1 odd): a ceiling division
                    if (P.cap_int) slot = k0 > 0 ? (int)((unsigned)((2 * k0 - 1) * (G - 1) + 2 * (int)nm1 - 1) / (2u * nm1)) : 0;
                    else { slot = 0; while (slot < G && rank_of(slot) < k0) ++slot; }
                    slot = slot < G ? slot : G;
                    nxt = slot < G ? rank_of(slot) : 0x7FFFFFFF;
                }
            }
            while (__any(k < k1)) {
                if (k < k1) {
                    if (kept == 0) {                                 // next window row (more kept bits exist: k < k1 <= n)
                        srp += AG; ++rmp;
                        const unsigned wd = *srp;
                        kept = wd & 0x1FFFFu; base = (int)(wd >> 17);
                        relm = (unsigned)(*rmp >> aca0);
                        dbf += inv_r; db2 = dbf * dbf;
                    }
                    if (kept != 0) {
                        const int c = __ffs(kept) - 1;
                        kept &= kept - 1;
                        const int cell = base + __popc(relm & ((1u << c) - 1u));
                        bool take = true;
                        if constexpr (CAP) {
                            if (nk > G) {
                                take = k == nxt;
                                if (take) nxt = slot + 1 < G ? rank_of(slot + 1) : 0x7FFFFFFF;
                            }
                        }
                        if constexpr (CAP) { if (take) row[slot] = (short)cell; slot += take ? 1 : 0; }
                        else row[k] = (short)cell;                   // under the cap the slot of a cell is its rank
                        ++k;
                        const float da = fmaf((float)c, inv_r, -apr_s);
                        const float u = fmaf(da, da, db2);
                        float psi = psi5_u_f32(u);
                        if constexpr (CAP) psi = take ? psi : 0.0f;
                        n0 = fmaf(psi, da, n0); n1 = fmaf(psi, dbf, n1); dn += psi;
                    }
                }
            }
        };
        if (__any(nk > G)) walk_range(std::true_type{}); else walk_range(std::false_type{});
        // the quad's partial sums -> every lane of the quad (two DPP exchanges)
        auto quad_sum = [](float v) -> float {
            v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
            v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
            return v;
        };
        auto lanes_sum = [&](float v) -> float {                     // ... and the second quad's sum into an 8-lane group's first lane
            v = quad_sum(v);
            if constexpr (LPA == 8) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x104, 0xF, 0xF, true));   // row_shl:4
            return v;
        };
        qn0 = lanes_sum(n0); qn1 = lanes_sum(n1); qdn = lanes_sum(dn); qnk = nk; qrl = Rl;
    }
    STAMP(19);
    EXIT_AT(9);
    // ---- (R) reward of the quad's agent (CPP:554-556), decided by the quad's first lane right here: everything it needs was
    // produced by this wave (the list, the sums) or long before (in-shape flag, collision flag).  The fp32 verdict only
    // DECIDES outside its guard band; an unsure agent is redone exactly by the whole wave.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");           // this wave's list rows: written above, read below
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        const int nsl = qnk > G ? G : qnk;
        const bool in_a = (sncf[ea] >> 30) != 0;
        const int ael = NPAD < 64 ? (ea & 63) / NPAD : 0, ai = NPAD < 64 ? (ea & 63) % NPAD : ea;
        const int ae = blockIdx.x * EPB + ael;
        const bool lead = (lane & (LPA - 1)) == 0 && ae < P.n_env && ai < n_a;
        const bool has = lead && in_a && nsl > 0;
        const float thr = P.rew_thr_k;                               // 0.05 in units of d_sen (the walk's scale)
        const float idn = __builtin_amdgcn_rcpf(qdn);                // (1 ulp + two roundings: far inside the 4e-6 the band allows)
        const float v0f = qn0 * idn, v1f = qn1 * idn;
        const float vf = sqrtf(fmaf(v0f, v0f, v1f * v1f));
        bool uni = has && vf < thr;
        const float inv_rl = __builtin_amdgcn_rcpf(qrl);             // the band is kept in lattice steps on the host: scale it
        const bool uns = has && (P.force_exact || !(qdn > 1e-6f) || !(fabsf(vf - thr) > (P.rew_ga_lat * (float)nsl * idn + P.rew_gb_lat) * inv_rl * 1.0001f));
        u64 um = __ballot(uns);
        while (um != 0) {
            const int L = __ffsll((unsigned long long)um) - 1;
            um &= um - 1;
            const bool res = exact_uniform(__builtin_amdgcn_readlane(ea, L), __builtin_amdgcn_readlane(nsl, L), __builtin_amdgcn_readlane(ae, L));
            if (lane == L) uni = res;
        }
        if (lead) {
            const size_t oi = (size_t)ae * n_a + ai;
            if (reward != nullptr) __builtin_nontemporal_store((in_a && !(snei[ea * kNeiStride + kTopoMax] != 0) && uni) ? 1.0f : 0.0f, &reward[oi]);   // CPP:554-556
            if (done != nullptr) __builtin_nontemporal_store((uint8_t)0, &done[oi]);                                                                     // ENV:480-482
        }
    }
    STAMP(6);
    EXIT_AT(10);
    // Work that needs nothing from the list phase: the prior policy on split B, the observation heads on the two splits whose
    // lists were shortest (A and C); the split with the longest lists (D) goes straight to its rows.
    prior_policy();
    if (obs != nullptr) {
        head_blocks(NW == 1);                        // (N > 64: dealt over all splits but B; the early A / C deal measured +15 % at N = 256)
        STAMP(9);
        EXIT_AT(11);
        sensed_rows(true, perm + (tid >> 6) * 64);
    }
    } else {
    // ---- occupied-cell filter, CPP:144-216: a sensed cell is occupied iff some nearby agent is within
    // r_avoid/2 of it; only agents inside the shape filter (CPP:150).  kept = in_shape ? sensed & ~occupied : sensed.
    // Common case (N <= 64, no index export, no agent flagged by the pair pass): the occupied bits of every word are
    // already in LDS -- the env's covered-cell set (lattice walk) or the scan's per-agent words -- so nothing is
    // rewritten and no barrier is needed: the readers below AND the two words on the fly.  Otherwise (N > 64: per-cell
    // ballots; generic cell sets: the occupied words share LDS with the rank-select bits; export: the occupied bits
    // themselves are wanted; a flagged agent: exact per-cell test) the words of the sensed set are filtered in place, dealt
    // over the splits, behind a barrier.
    const bool flagged = use_lat && __any((sflag[at] & 1) != 0) != 0;
    const bool slow_filter = NW > 1 || !use_lat || P.export_idx != 0 || flagged;     // workgroup-uniform for N <= 64: every wave holds the same agents
    if (slow_filter) {
        {
            u64 nearby[NW];
    #pragma unroll
            for (int q = 0; q < NW; ++q) nearby[q] = (NW > 1 && in_shape) ? snear[q * AG + at] : 0;
            for (int rep = 0, reps = REPS(4); rep < reps; ++rep)
            for (int w = 0; w < W; ++w) {
                if (!mine(w)) continue;
                FENCE();
                const unsigned word = sbits[w * AG + at];
                unsigned kw = word;
                if (use_lat) {
                    // occupied <=> within r_avoid/2 of ANY agent: the covering agent of a SENSED cell is "nearby"
                    // (CPP:161) by the triangle inequality, except when its distance sits within rounding of the
                    // nearby threshold -- those lanes were flagged by the pair pass and are resolved exactly.
                    const unsigned cw_ = cov[el * (P.ngw + 1) + w];
                    if (in_shape) {
                        kw = word & ~cw_;
                        if ((sflag[at] & 1) != 0) {
                            unsigned it = word & cw_;
                            kw = word;
                            while (it) {
                                const int b = __ffs(it) - 1; it &= it - 1;
                                const double2 g = cell64(w * 32 + b);
                                bool occ = false;
                                for (int q = 0; q < NW && !occ; ++q) {
                                    u64 nbm = NW == 1 ? (nearby1 >> (NPAD < 64 ? el * NPAD : 0)) : snear[q * AG + at];
                                    while (nbm && !occ) {
                                        const int j = q * 64 + __ffsll((unsigned long long)nbm) - 1; nbm &= nbm - 1;
                                        const double ex = g.x - spx[j], ey = g.y - spy[j];
                                        occ = ex * ex + ey * ey < P.c_occ;
                                    }
                                }
                                if (occ) kw &= ~(1u << b);
                            }
                        }
                    }
                    if (rep == reps - 1) sbits[w * AG + at] = kw;
                } else if constexpr (NW == 1) {
                    if (in_shape) kw = word & ~owords[w * AG + at];       // occupied bits came out of the scan
                    if (rep == reps - 1) sbits[w * AG + at] = kw;
                } else if (in_shape) {
                    unsigned it = word;
                    while (it) {
                        int bb[4]; bool occ[4];
    #pragma unroll
                        for (int u = 0; u < 4; ++u) { bb[u] = it ? __ffs(it) - 1 : -1; it &= it - 1; }   // it - 1 of 0 is harmless: it stays 0
    #pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int c = w * 32 + (bb[u] < 0 ? 0 : bb[u]);
                            occ[u] = false;
    #pragma unroll
                            for (int q = 0; q < NW; ++q) occ[u] = occ[u] || ((cmask[(size_t)c * NW + q] & nearby[q]) != 0);
                        }
    #pragma unroll
                        for (int u = 0; u < 4; ++u) if (bb[u] >= 0 && occ[u]) kw &= ~(1u << bb[u]);
                    }
                    if (rep == reps - 1) sbits[w * AG + at] = kw;
                }
                asm volatile("" :: "v"(kw));
                if (P.export_idx) obits[w * AG + at] = word & ~kw;
                pc[w * AG + at] = (unsigned char)__popc(kw);
            }
        }

    }
    auto kept_word = [&](int w) -> unsigned {
        const unsigned word = sbits[w * AG + at];
        if (slow_filter || !in_shape) return word;
        return word & ~cov[el * (P.ngw + 1) + w];
    };
    // kept-bit count of every word, dealt over the splits and shared through LDS: each of the WPE splits needs the counts of
    // ALL words (list length, start of its rank range), and vector instructions -- not barriers -- are what this kernel
    // runs out of, so nothing is counted four times
    if (!slow_filter)
        for (int w = sx; w < W; w += WPE) pc[w * AG + at] = (unsigned char)__popc(kept_word(w));
    __syncthreads();
    STAMP(5);
    EXIT_AT(6);

    // ---- capped sensed list (CPP:236-271) into LDS, and the exploration reward's sums over it (CPP:494-551).
    // Each split emits the slots of its own words (rank = prefix of the kept-bit counts) and accumulates
    // partial sums; the sums are combined in split order below.
    int n_kept = 0;
    for (int w = 0; w < W; ++w) n_kept += pc[w * AG + at];
    n_sel = n_kept > G ? G : n_kept;
    // (1) which RANKS of the kept list survive the cap?  rank(s) = round(s * (n-1)/(G-1)), s = 0..G-1 (CPP:241-245),
    // evaluated in a uniform slot loop (split sx takes slots sx, sx+WPE, ...) and OR-ed into a per-lane bit set.
    // With G-1 odd the value s(n-1)/(G-1) is never within 1/(2(G-1)) of a half-integer, so the reference's fp64
    // round() equals the integer floor((2 s (n-1) + (G-1)) / (2 (G-1))) exactly; for even G-1 (ties possible) the
    // reference's fp64 expression is evaluated as is.
    unsigned *rsel = reinterpret_cast<unsigned *>(smem + P.off_cmask);      // [W+1][AG] (aliases cmask: consumed)
    if (!use_lat) {                       // lattice mode cleared it up front (the region has no earlier user there)
        for (int w = sx; w <= W; w += WPE) rsel[w * AG + at] = 0;
        __syncthreads();
    }
    // The cap is rare (an agent deep inside a fine-celled shape): when no agent of this wave is capped -- the same
    // answer in all WPE splits, they hold the same agents -- ranks are slots and the rank-select bits are skipped.
    const bool any_sub = __any(n_kept > G) != 0;
    int *sub_base = reinterpret_cast<int *>(smem + P.off_partc);   // [WPE][AG] first slot of each split's rank range, capped agents only (part_c is consumed)
    // (for N <= 64 `any_sub` is the same in every wave of the workgroup, so the barriers it guards are uniform)
    if (any_sub)
    for (int rep = 0, reps = REPS(5); rep < reps; ++rep) {
        FENCE();
        const bool sub = n_kept > G;
        // lanes under the cap: rank = slot, i.e. bits [0, n_kept) -- whole words, dealt over the splits
        if (!sub)
            for (int w = sx; w * 32 < G; w += WPE) {
                const int left = n_kept - w * 32;
                if (left > 0) rsel[w * AG + at] = left >= 32 ? 0xFFFFFFFFu : ((1u << left) - 1u);
            }
        // capped lanes (rare, usually one or two per wave): wave-cooperative, lane = slot -- G/64 passes per capped agent
        // instead of every lane looping over its G slots; the capped agents are dealt over the splits
        u64 sm = __ballot(sub);
        int k = 0;
        while (sm != 0) {
            const int L = __ffsll((unsigned long long)sm) - 1;
            sm &= sm - 1;
            if ((k++ % WPE) != sx) continue;
            const unsigned nm1 = (unsigned)(__builtin_amdgcn_readlane(n_kept, L) - 1);
            const double step = (double)nm1 / (G - 1);
            unsigned *rl = rsel + (at & ~63) + L;                                  // agent thread of lane L in this wave
            const unsigned perL = (nm1 + 1 + WPE - 1) / WPE;                       // ranks per split in the emission below
            int below[WPE];                                                        // selected ranks below split s's first rank
#pragma unroll
            for (int sp_ = 0; sp_ < WPE; ++sp_) below[sp_] = 0;
            for (int q0 = 0; q0 < G; q0 += 64) {
                const int q = q0 + lane;
                unsigned r = 0xFFFFFFFFu;
                if (q < G) {
                    if (P.cap_int) {
                        const unsigned x = 2u * (unsigned)q * nm1 + (unsigned)(G - 1);
                        const unsigned hq = __umulhi(x, P.cap_magic);
                        r = (((x - hq) >> 1) + hq) >> P.cap_shift;                 // x / (2 (G-1)), Granlund-Montgomery
                    } else {
                        r = (unsigned)(int)round(q * step);
                    }
                    atomicOr(&rl[(r >> 5) * AG], 1u << (r & 31));
                }
#pragma unroll
                for (int sp_ = 1; sp_ < WPE; ++sp_) below[sp_] += __popcll(__ballot(r < (unsigned)sp_ * perL));
            }
            if (lane == 0) {
#pragma unroll
                for (int sp_ = 1; sp_ < WPE; ++sp_) sub_base[sp_ * AG + (at & ~63) + L] = below[sp_];
            }
        }
    }
    if (NW > 1 || any_sub) __syncthreads();
    STAMP(18);
    EXIT_AT(7);
    // (2) emit: the kept list of an agent is cut into WPE contiguous RANK ranges, one per split.  Each lane walks its
    // own range bit by bit with a lane-private word pointer, so a wave's trip count is the longest range of any lane
    // (n_kept / WPE) -- not, as with words dealt over the splits, the sum over words of the fullest lane's count.
    int slot_lo = 0, slot_hi = 0;            // the slots this split emitted: its share of the reward sums below
    {
        short *row = sidx + (size_t)at * P.g_stride;
        for (int rep = 0, reps = REPS(11); rep < reps; ++rep) {
        FENCE();
        const int per = (n_kept + WPE - 1) / WPE;
        const int k0 = sx * per < n_kept ? sx * per : n_kept;
        const int k1 = k0 + per < n_kept ? k0 + per : n_kept;
        // the word holding rank k0 and the number of kept bits before k0 inside it
        int ww = 0, within = k0;
        {
            int prefix = 0;
            for (int w = 0; w < W; ++w) {                              // ww = number of words whose running count is <= k0
                prefix += pc[w * AG + at];
                const bool le = prefix <= k0;
                ww += le ? 1 : 0; within = le ? k0 - prefix : within;
            }
        }
        unsigned it = k0 < k1 ? kept_word(ww) : 0u;
        {   // drop the `within` lowest set bits: position of the within-th set bit by a binary search on popcounts
            int p = 0, left = within;
#pragma unroll
            for (int sh = 16; sh >= 1; sh >>= 1) {
                const int c = __popc((it >> p) & ((1u << sh) - 1u));
                const bool go = left >= c;
                left -= go ? c : 0; p += go ? sh : 0;
            }
            it &= ~((1u << p) - 1u);                                   // p <= 31: the within-th set bit exists (k0 < k1)
        }
        // uncapped agent: rank = slot; capped: the rank-select pass counted the selected ranks below each split's range
        int slot = (n_kept > G && sx > 0) ? sub_base[sx * AG + at] : (n_kept > G ? 0 : k0);
        slot_lo = slot;
        // two instantiations of the walk: without a capped agent in the wave every kept bit is taken and no selection
        // word is carried
        auto walk_range = [&](auto capped) {
            constexpr bool CAP = decltype(capped)::value;
            int k = k0;
            unsigned rw = (CAP && k0 < k1) ? rsel[(k0 >> 5) * AG + at] : 0xFFFFFFFFu;      // selection bits of ranks 32 (k >> 5) ...
            while (__any(k < k1)) {
                if (k < k1) {
                    if (it == 0) { ++ww; it = kept_word(ww); }         // next word (more kept bits exist: k < k1 <= n_kept)
                    if (it != 0) {
                        const int b = __ffs(it) - 1;
                        it &= it - 1;
                        const bool take = !CAP || ((rw >> (k & 31)) & 1u) != 0;
                        if (take) row[slot] = (short)(ww * 32 + b);
                        slot += take ? 1 : 0;
                        ++k;
                        if (CAP && (k & 31) == 0 && k < k1) rw = rsel[(k >> 5) * AG + at];
                    }
                }
            }
        };
        if (any_sub) walk_range(std::true_type{}); else walk_range(std::false_type{});
        slot_hi = slot;
        }
        for (int q = n_sel + sx; q < G; q += WPE) row[q] = -1;
    }
    STAMP(19);
    if (NW > 1 || any_sub) __syncthreads();      // the partial sums below overwrite the rank-select bits other splits may still read
    STAMP(20);
    EXIT_AT(8);
    // exploration-reward sums over the capped list (CPP:494-551), fp32 fast path: every split sums the slots it has just
    // emitted itself (its own LDS writes: no barrier in between), two per iteration in packed fp32.  The fp32 result
    // only DECIDES when |v| is outside a guard band around the 0.05 threshold; inside it the sums are redone in fp64 below.
    {
        const float inv_dsen2 = (float)(1.0 / (P.d_sen * P.d_sen));
        float num0 = 0.0f, num1 = 0.0f, den = 0.0f;
        const short *row = sidx + (size_t)at * P.g_stride;
        const int lim = in_shape ? slot_hi : slot_lo;
        for (int rep = 0, reps = REPS(6); rep < reps; ++rep) {
        FENCE();
        f2v n0v = {0.0f, 0.0f}, n1v = {0.0f, 0.0f}, dnv = {0.0f, 0.0f};
        const f2v pxx2 = {pxf, pxf}, pyy2 = {pyf, pyf}, inv2 = {inv_dsen2, inv_dsen2};
        for (int q = slot_lo; __any(q < lim); q += 2) {
            const bool va = q < lim, vb = q + 1 < lim;
            const int ca = va ? row[q] : 0, cb = vb ? row[q + 1] : 0;         // cell 0 stands in for an empty slot (weight zeroed)
            f2v gx, gy;
            if (use_lat) { const double2 ga = gce[ca], gb = gce[cb]; gx = f2v{(float)ga.x, (float)gb.x}; gy = f2v{(float)ga.y, (float)gb.y}; }
            else {
                gx = f2v{cq_e[(ca >> 1) * 4 + (ca & 1)], cq_e[(cb >> 1) * 4 + (cb & 1)]};
                gy = f2v{cq_e[(ca >> 1) * 4 + 2 + (ca & 1)], cq_e[(cb >> 1) * 4 + 2 + (cb & 1)]};
            }
            const f2v x = gx - pxx2, y = gy - pyy2;
            const f2v u = __builtin_elementwise_fma(x, x, y * y) * inv2;
            f2v c = {7.969553699e-04f, 7.969553699e-04f};                      // psi_u_f32, both slots at once
            c = __builtin_elementwise_fma(c, u, f2v{-1.267949212e-02f, -1.267949212e-02f});
            c = __builtin_elementwise_fma(c, u, f2v{1.175149009e-01f, 1.175149009e-01f});
            c = __builtin_elementwise_fma(c, u, f2v{-6.675792336e-01f, -6.675792336e-01f});
            c = __builtin_elementwise_fma(c, u, f2v{2.029347420e+00f, 2.029347420e+00f});
            c = __builtin_elementwise_fma(c, u, f2v{-2.467400551e+00f, -2.467400551e+00f});
            c = __builtin_elementwise_fma(c, u, f2v{1.0f, 1.0f});
            const f2v psi = {va ? c.x : 0.0f, vb ? c.y : 0.0f};
            n0v = __builtin_elementwise_fma(psi, x, n0v); n1v = __builtin_elementwise_fma(psi, y, n1v); dnv = dnv + psi;
        }
        num0 = n0v.x + n0v.y; num1 = n1v.x + n1v.y; den = dnv.x + dnv.y;
        }
        rsum[(sx * 3 + 0) * AG + at] = num0; rsum[(sx * 3 + 1) * AG + at] = num1; rsum[(sx * 3 + 2) * AG + at] = den;
    }
    STAMP(21);
    __syncthreads();
    STAMP(6);
    EXIT_AT(9);

    if (sx == 0) {
        float n0 = 0.0f, n1 = 0.0f, dn = 0.0f;
#pragma unroll
        for (int s = 0; s < WPE; ++s) {
            n0 += rsum[(s * 3 + 0) * AG + at]; n1 += rsum[(s * 3 + 1) * AG + at]; dn += rsum[(s * 3 + 2) * AG + at];
        }
        const bool has = in_shape && n_sel > 0;
        const float v0f = n0 / dn, v1f = n1 / dn;
        const float vf = sqrtf(fmaf(v0f, v0f, v1f * v1f));
        uniform = has && vf < 0.05f;
        unsure = has && (P.force_exact || !(dn > 1e-6f) || !(fabsf(vf - 0.05f) > P.rew_ga * (float)n_sel / dn + P.rew_gb));
    }
    }

    if constexpr (!LAT) {
        // generic path: split A combines the splits' fp32 sums (above) and stores the reward
        if (sx == 0) {
            u64 um = __ballot(unsure);
            while (um != 0) {
                const int L = __ffsll((unsigned long long)um) - 1;               // agent thread, wave-uniform
                um &= um - 1;
                const bool res = exact_uniform((at & ~63) + L, __builtin_amdgcn_readlane(n_sel, L), blockIdx.x * EPB + (NPAD < 64 ? L / NPAD : 0));
                if (lane == L) uniform = res;
            }
            if (act) {
                if (reward != nullptr) __builtin_nontemporal_store((in_shape && !(snei[at * kNeiStride + kTopoMax] != 0) && uniform) ? 1.0f : 0.0f, &reward[(size_t)e * n_a + i]);   // CPP:554-556
                if (done != nullptr) __builtin_nontemporal_store((uint8_t)0, &done[(size_t)e * n_a + i]);                                                         // ENV:480-482
            }
        }
    }
    if (P.export_idx) {
        if constexpr (LAT) __syncthreads();          // export launches only (workgroup-uniform): every wave's lists are complete
        if (sx == 0) {
        if (P.export_idx && act) {
            const short *row = sidx + (size_t)at * P.g_stride;
            int *es_ = P.exp_sensed + ((size_t)e * n_a + i) * G;
            for (int q = 0; q < G; ++q) es_[q] = row[q];
            // occupied list with its own cap, CPP:217-233.  NWD words of occupied bits in ascending cell order: the target-cell
            // words of the generic scan, or the window rows of the lattice path (row-major = ascending cell index)
            const int NWD = LAT ? P.lat_nrs : W;
            const unsigned *ob = LAT ? orow : obits;
            int ob0 = 0, oca0 = 0;
            if constexpr (LAT) { const float4 hq = hdr[at]; ob0 = __float_as_int(hq.z); oca0 = __float_as_int(hq.w); }
            int n_occ = 0;
            for (int w = 0; w < NWD; ++w) n_occ += __popc(ob[w * AG + at]);
            const int O = P.occ_max;
            int *eo = P.exp_occ + ((size_t)e * n_a + i) * O;
            const bool osub = n_occ > O;
            const double ostep = osub ? (double)(n_occ - 1) / (O - 1) : 0.0;
            int os = 0, ok = 0, otarget = 0;
            for (int w = 0; w < NWD; ++w) {
                unsigned it = ob[w * AG + at];
                int cbase = w * 32; unsigned relm = 0xFFFFFFFFu;           // generic: cell = 32 w + bit
                if constexpr (LAT) {
                    const int bq = ob0 + w < 0 ? 0 : (ob0 + w > 63 ? 63 : ob0 + w);
                    const u64 rowm = lrm[el * 64 + bq];
                    cbase = lrs[el * 64 + bq] + __popcll(rowm & ((1ull << oca0) - 1ull)); relm = (unsigned)(rowm >> oca0);
                }
                while (it) {
                    const int b = __ffs(it) - 1;
                    it &= it - 1;
                    const bool sel = osub ? (ok == otarget) : true;
                    if (sel && os < O) { eo[os++] = cbase + __popc(relm & ((1u << b) - 1u)); if (osub) otarget = (int)round(os * ostep); }
                    ++ok;
                }
            }
            for (; os < O; ++os) eo[os] = -1;
        }
        }
    }
    STAMP(22);
    if constexpr (!LAT) {
        EXIT_AT(10);
        prior_policy();
        if (obs != nullptr) {
            head_blocks(false);
            STAMP(9);
            EXIT_AT(11);
            sensed_rows(false, nullptr);
        }
    }
    STAMP(7);
#ifdef SWARM_STAMPS
    if (P.stamps != nullptr && lane == 0) P.stamps[((size_t)blockIdx.x * (T / 64) + (tid >> 6)) * 24 + 23] = sx;   // slot 23 = the wave's split (role)
#endif
}

// -------------------------------------------------------------------------------------------------
// batched reset: AssemblySwarmEnv.reset(), ENV:156-219, for every environment at once.
// Counter-based generator: draw k of environment g in episode ep under `seed` is
//     u = (mix64(mix64(mix64(seed + GOLD*(ep+1)) ^ g) + GOLD*(k+1)) >> 11) * 2^-53   in [0, 1)
// (splitmix64 finaliser), so any env range can be generated on any rank without communication.  Draw slots mirror
// the reference's order: 0 shape index (:160), 1 angle (:175), 2-3 the discarded offset (:182), 4-5 offset (:184-185),
// 6 branch coin (:202), 7-8 cluster centre (:207-208), then per agent x, y (:203-208) and vx, vy (:215).
// -------------------------------------------------------------------------------------------------
struct ShapeSet {
    int n_shapes;
    const double *cells;      // [S][2][ng_max], shape frame (ENV: grid_center_origins[s].T)
    const int *n_g;           // [S]
    const double *l_cell;     // [S]
    const double *c_in;       // [S] in-shape cut-off
    const LatEnv *lat;        // [S] lattice of the un-rotated shape (nrows == 0: not a lattice)
};

__host__ __device__ inline unsigned long long mix64(unsigned long long z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__host__ __device__ inline double reset_u01(unsigned long long key, unsigned k)
{
    return (double)(mix64(key + 0x9E3779B97F4A7C15ull * (unsigned long long)(k + 1)) >> 11) * (1.0 / 9007199254740992.0);
}

__global__ void __launch_bounds__(256)
k_reset(const KP P, const ShapeSet S, const unsigned long long seed, const unsigned long long episode,
        const long long env_offset, double *cells_out, int *ng_out, double *cin_out, LatEnv *lat_out, int *shape_out)
{
    const int e = blockIdx.x, tid = threadIdx.x;
    const unsigned long long key = mix64(mix64(seed + 0x9E3779B97F4A7C15ull * (episode + 1)) ^ (unsigned long long)(env_offset + e));
    const double W = P.w_half, H = P.h_half;
    int s = (int)(reset_u01(key, 0) * S.n_shapes);
    s = s >= S.n_shapes ? S.n_shapes - 1 : s;
    const double ang = M_PI * (2.0 * reset_u01(key, 1) - 1.0);
    const double cs = cos(ang), sn = sin(ang);                       // rotate_matrix = [[c, s], [-s, c]], ENV:177
    const double offx = (-W + 1) + reset_u01(key, 4) * (2 * W - 2);
    const double offy = (-H + 1) + reset_u01(key, 5) * (2 * H - 2);
    const int ng = S.n_g[s];
    const double *sx_ = S.cells + (size_t)s * 2 * P.ng_max, *sy_ = sx_ + P.ng_max;
    double *gx = cells_out + (size_t)e * 2 * P.ng_max, *gy = gx + P.ng_max;
    for (int c = tid; c < P.ng_max; c += blockDim.x) {
        double x = 0.0, y = 0.0;
        if (c < ng) { x = cs * sx_[c] + sn * sy_[c] + offx; y = -sn * sx_[c] + cs * sy_[c] + offy; }   // ENV:178,187
        gx[c] = x; gy[c] = y;
    }
    if (tid == 0) {
        ng_out[e] = ng; cin_out[e] = S.c_in[s]; shape_out[e] = s;
        LatEnv L = S.lat[s];
        if (L.nrows > 0) {          // rotate / shift the shape's lattice: u' = R u, v' = R v, o' = R o + offset
            // shape-frame basis from the stored inverse basis: u = uxi / |uxi|^2
            const double iu = 1.0 / (L.uxi * L.uxi + L.uyi * L.uyi), iv = 1.0 / (L.vxi * L.vxi + L.vyi * L.vyi);
            const double ux = L.uxi * iu, uy = L.uyi * iu, vx = L.vxi * iv, vy = L.vyi * iv;
            const double rux = cs * ux + sn * uy, ruy = -sn * ux + cs * uy;
            const double rvx = cs * vx + sn * vy, rvy = -sn * vx + cs * vy;
            const double rox = cs * L.ox + sn * L.oy + offx, roy = -sn * L.ox + cs * L.oy + offy;
            L.ox = rox; L.oy = roy;
            L.uxi = rux / iu; L.uyi = ruy / iu; L.vxi = rvx / iv; L.vyi = rvy / iv;
        }
        lat_out[e] = L;
    }
    // agents (ENV:202-215)
    const int N = P.n_a;
    const bool spread = (2.0 * reset_u01(key, 6) - 1.0) > 0;
    const double cx = (-W + 1) + reset_u01(key, 7) * (2 * W - 2), cy = (-H + 1) + reset_u01(key, 8) * (2 * H - 2);
    for (int i = tid; i < N; i += blockDim.x) {
        const double ux_ = reset_u01(key, 16 + i), uy_ = reset_u01(key, 16 + N + i);
        double x, y;
        if (spread) { x = -W + ux_ * (2 * W); y = -H + uy_ * (2 * H); }
        else { x = (2.0 * ux_ - 1.0) + cx; y = (2.0 * uy_ - 1.0) + cy; }
        P.p[(size_t)e * 2 * N + i] = x; P.p[(size_t)e * 2 * N + N + i] = y;
        P.dp[(size_t)e * 2 * N + i] = -0.5 + reset_u01(key, 16 + 2 * N + i);
        P.dp[(size_t)e * 2 * N + N + i] = -0.5 + reset_u01(key, 16 + 3 * N + i);
    }
}

// -------------------------------------------------------------------------------------------------
// evaluation metrics: AssemblySwarmWrapper.coverage_rate / distribution_uniformity /
// voronoi_based_uniformity, /root/reference/cus_gym/gym/wrappers/customized_envs/assembly_wrapper.py:48-128, per env.
// fp64 in numpy's operation order, including np.var's two-pass form and numpy's pairwise summation (blocks of 8
// accumulators up to 128 elements, recursive halves above), so the values are bit-identical to the Python loops.
// -------------------------------------------------------------------------------------------------
__device__ double np_pairwise_sum(const double *a, int n)
{
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        int i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

// (np.var(v) - min(v)) / (max(v) - min(v)), assembly_wrapper.py:96-99,125-126; `tmp` holds n doubles of scratch
__device__ double np_var_metric(const double *v, double *tmp, int n)
{
    const double mean = np_pairwise_sum(v, n) / n;
    double mn = v[0], mx = v[0];
    for (int i = 0; i < n; ++i) {
        const double d = v[i] - mean;
        tmp[i] = d * d;
        mn = v[i] < mn ? v[i] : mn; mx = v[i] > mx ? v[i] : mx;
    }
    const double var = np_pairwise_sum(tmp, n) / n;
    return (var - mn) / (mx - mn);
}

__global__ void __launch_bounds__(256)
k_metrics(const KP P, double *__restrict__ out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int N = P.n_a, e = blockIdx.x, tid = threadIdx.x;
    double *px = reinterpret_cast<double *>(smem), *py = px + N;       // [N], [N]
    double *val = py + N, *tmp = val + N;                              // [N] per-agent values, [N] scratch
    int *cnt = reinterpret_cast<int *>(tmp + N);                       // [N] Voronoi counts, then [1] coverage count
    const int ng = P.n_g[e];
    const double *gx = P.cells + (size_t)e * 2 * P.ng_max, *gy = gx + P.ng_max;
    for (int i = tid; i < N; i += blockDim.x) {
        px[i] = P.p[(size_t)e * 2 * N + i]; py[i] = P.p[(size_t)e * 2 * N + N + i];
        cnt[i] = 0;
    }
    if (tid == 0) cnt[N] = 0;
    __syncthreads();
    // coverage (assembly_wrapper.py:58-73) and Voronoi owner (:110-121) of every cell
    const double half = P.r_avoid / 2;
    for (int c = tid; c < ng; c += blockDim.x) {
        bool covered = false;
        double best = 0.0; int owner = 0;
        for (int j = 0; j < N; ++j) {
            const double dx = px[j] - gx[c], dy = py[j] - gy[c];
            const double d = sqrt(dx * dx + dy * dy);                 // np.linalg.norm(axis=0)
            covered = covered || (d < half);
            if (j == 0 || d < best) { best = d; owner = j; }          // np.argmin: first minimum
        }
        if (covered) atomicAdd(&cnt[N], 1);
        atomicAdd(&cnt[owner], 1);
    }
    // minimum non-zero distance of every agent (:85-93)
    for (int i = tid; i < N; i += blockDim.x) {
        double m = INFINITY;
        for (int j = 0; j < N; ++j) {
            const double dx = px[j] - px[i], dy = py[j] - py[i];
            const double d = sqrt(dx * dx + dy * dy);
            if (d != 0 && d < m) m = d;
        }
        val[i] = m;
    }
    __syncthreads();
    if (tid == 0) {
        out[(size_t)e * 3 + 0] = (double)cnt[N] / ng;
        out[(size_t)e * 3 + 1] = np_var_metric(val, tmp, N);
    }
    __syncthreads();
    for (int i = tid; i < N; i += blockDim.x) val[i] = (double)cnt[i];
    __syncthreads();
    if (tid == 0) out[(size_t)e * 3 + 2] = np_var_metric(val, tmp, N);
}


// -------------------------------------------------------------------------------------------------
// rule-based expert controller: agent_strategy == 'rule',
// /root/reference/cus_gym/gym/envs/customized_envs/assembly.py:530-601, for the CURRENT state.  Not a hot path (expert
// data collection, collect_expert_data.py): one thread per agent, fp64 in numpy's operation order (np.sum's pairwise
// blocks of 8 included).  It consumes what the observation pass left in HBM: nearest cell / in-shape flag and the
// capped sensed-cell list (`exp_sensed`, the same filter + round(i*step) selection as :544-572).  np.cos is numpy's
// vectorised routine, so v_exp agrees to a few ulp, not bit for bit (tests: 1e-12 absolute on the clipped action).
// -------------------------------------------------------------------------------------------------
template <class F>
__device__ double np_sum_stream(int n, F f)      // np.sum of f(0..n-1) for n <= 128, values generated on the fly
{
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += f(i);
        return res;
    }
    double r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = f(k);
    const int n8 = n - (n % 8);
    int i;
    for (i = 8; i < n8; i += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] += f(i + k);
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += f(i);
    return res;
}

__global__ void __launch_bounds__(256)
k_rule(const KP P, double *__restrict__ out)     // out [E][N][2]
{
    const int N = P.n_a, e = blockIdx.x, G = P.g_max;
    const double *px = P.p + (size_t)e * 2 * N, *py = px + N;
    const double *vx = P.dp + (size_t)e * 2 * N, *vy = vx + N;
    const double *gx = P.cells + (size_t)e * 2 * P.ng_max, *gy = gx + P.ng_max;
    const double k_1 = 1, k_2 = 15, k_3 = 17;                                  // :532
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const double xi = px[i], yi = py[i], ui = vx[i], wi = vy[i];
        const bool in_shape = P.in_flag[(size_t)e * N + i] != 0;
        double ent_x = 0.0, ent_y = 0.0;                                       // :538-541
        if (!in_shape) {
            const int bc = P.near_cell[(size_t)e * N + i];
            const double rx = gx[bc] - xi, ry = gy[bc] - yi;
            const double nr = sqrt(rx * rx + ry * ry) + 1e-8;
            ent_x = k_1 * (rx / nr) + (0.0 - ui);
            ent_y = k_1 * (ry / nr) + (0.0 - wi);
        }
        const int *sel = P.exp_sensed + ((size_t)e * N + i) * G;               // capped list, -1 padded (:561-572)
        int n = 0;
        while (n < G && sel[n] >= 0) ++n;
        double exp_x = 0.0, exp_y = 0.0;                                       // :574-584
        if (n > 0) {
            auto psi = [&](double rx, double ry) {                             // _rho_cos_dec(z, 0, d_sen) :846-850
                const double z = sqrt(rx * rx + ry * ry);
                return z < P.d_sen ? 0.5 * (1.0 + cos(M_PI * (z / P.d_sen - 0) / (1.0 - 0))) : 0.0;
            };
            const double sx = np_sum_stream(n, [&](int q) { const int c = sel[q]; const double rx = gx[c] - xi, ry = gy[c] - yi; return psi(rx, ry) * rx; });
            const double sy = np_sum_stream(n, [&](int q) { const int c = sel[q]; const double rx = gx[c] - xi, ry = gy[c] - yi; return psi(rx, ry) * ry; });
            double den = np_sum_stream(n, [&](int q) { const int c = sel[q]; return psi(gx[c] - xi, gy[c] - yi); });
            if (den == 0) den = 1e-8;
            exp_x = k_2 * sx / den; exp_y = k_2 * sy / den;
        }
        int n_near = 0;                                                        // :587-598
        for (int j = 0; j < N; ++j) {
            const double rx = px[j] - xi, ry = py[j] - yi;
            n_near += (j != i && sqrt(rx * rx + ry * ry) < P.d_sen) ? 1 : 0;
        }
        double int_x = 0.0, int_y = 0.0;
        for (int j = 0; j < N; ++j) {
            const double rx = px[j] - xi, ry = py[j] - yi;
            const double nr = sqrt(rx * rx + ry * ry);
            if (j == i || !(nr < P.d_sen)) continue;
            if (nr < P.r_avoid) {
                const double c = -k_3 * (P.r_avoid / nr - 1);
                int_x += c * rx; int_y += c * ry;
            }
            int_x += 5 * (vx[j] - ui) / n_near; int_y += 5 * (vy[j] - wi) / n_near;
        }
        const double ax = (ent_x + exp_x) + int_x, ay = (ent_y + exp_y) + int_y;
        out[((size_t)e * N + i) * 2 + 0] = fmin(fmax(ax, -1.0), 1.0);          // np.clip :601
        out[((size_t)e * N + i) * 2 + 1] = fmin(fmax(ay, -1.0), 1.0);
    }
}

// (x, y)-interleaved copy of the target cells of envs [e0, e0 + count): the step kernel gathers cells per lane, and one
// 16-byte load per cell costs half the address-unit work of two 8-byte loads from the ABI's [2][ng_max] layout.
__global__ void __launch_bounds__(256)
k_interleave(const double *__restrict__ cells, double2 *__restrict__ out, int ng_max, int e0, int count)
{
    const size_t n = (size_t)count * ng_max;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
        const size_t e = e0 + q / ng_max, c = q % ng_max;
        double2 g; g.x = cells[e * 2 * ng_max + c]; g.y = cells[e * 2 * ng_max + ng_max + c];
        out[e * ng_max + c] = g;
    }
}

// -------------------------------------------------------------------------------------------------
// The reference-shaped host outputs ("a single host-side gather of obs / reward"): the step
// leaves obs [E][N][D], reward [E][N], done [E][N], a_prior [E][N][2] on the device; the numpy API of
// AssemblySwarmEnv.step returns obs (D, n_a) / reward (1, n_a) / a_prior (2, n_a) as float64 and done (1, n_a) as bool with
// the environments side by side on the agent axis (assembly.py:487-666, 227-231, 353, 480-482).  k_export writes exactly
// that block -- widened to double, transposed -- into ONE contiguous device buffer that a single hipMemcpyAsync moves
// into pinned host memory: no per-step allocation, no host-side pass over the data.
// Block layout (doubles): obs D*EN | a_prior 2*EN | reward EN | done EN bytes.
// -------------------------------------------------------------------------------------------------
template <typename OT> __device__ __forceinline__ double wide(OT v) { return (double)v; }
template <> __device__ __forceinline__ double wide<__bf16>(__bf16 v) { return (double)(float)v; }

template <typename OT>
__global__ void __launch_bounds__(256)
k_export(const OT *__restrict__ obs, const float *__restrict__ reward, const uint8_t *__restrict__ done,
         const OT *__restrict__ prior, double *__restrict__ out, const int D, const long long EN, const int with_prior)
{
    // tile: 64 agent rows x 32 features through LDS: reads run along a row (features contiguous), writes along the agent axis
    __shared__ double tile[32][65];
    const int tid = threadIdx.x;
    const long long r0 = (long long)blockIdx.x * 64;
    for (int f0 = 0; f0 < D; f0 += 32) {
        const int fw = D - f0 < 32 ? D - f0 : 32;
        {
            const int f = tid & 31;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int r = (tid >> 5) + 8 * k;
                if (f < fw && r0 + r < EN) tile[f][r] = wide<OT>(obs[(size_t)(r0 + r) * D + f0 + f]);
            }
        }
        __syncthreads();
        {
            const int r = tid & 63;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int f = (tid >> 6) + 4 * k;
                if (f < fw && r0 + r < EN) __builtin_nontemporal_store(tile[f][r], &out[(size_t)(f0 + f) * EN + r0 + r]);
            }
        }
        __syncthreads();
    }
    if (tid < 64 && r0 + tid < EN) {
        const long long a = r0 + tid;
        double *pri = out + (size_t)D * EN, *rew = pri + 2 * EN;
        uint8_t *dn = reinterpret_cast<uint8_t *>(rew + EN);
        if (with_prior) { pri[a] = wide<OT>(prior[2 * a]); pri[EN + a] = wide<OT>(prior[2 * a + 1]); }
        if (reward != nullptr) rew[a] = (double)reward[a];
        if (done != nullptr) dn[a] = done[a];
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
    int attr_smem[24];
    bool half;                     // the half-occupied geometry is in use (set_lattice_mode)
    int n_cu;
    std::vector<char> cells_set;
    std::string err;
    // device buffers
    double *d_p, *d_dp, *d_cells, *d_cin;
    double2 *d_cells_xy;
    LatEnv *d_lat;
    // shape set for the device-side reset
    int n_shapes;
    double *d_shape_cells, *d_shape_l, *d_shape_cin;
    int *d_shape_ng;
    int *d_shape_idx;              // [E] shape index drawn by the last swarm_reset (-1 before / after swarm_set_cells)
    LatEnv *d_shape_lat;
    bool shapes_lattice; float shapes_rmax, shapes_cmax; int shapes_ncols;
    std::vector<char> lat_ok;      // per env: cells are a lattice subset
    std::vector<float> lat_R, lat_Rc;
    std::vector<int> lat_ncols;
    bool lattice_disabled;
    int *d_nei, *d_near, *d_inflag, *d_ng, *d_exp_sensed, *d_exp_occ;
    double2 *d_sf;
    void *d_prior;
    double2 *d_act_next;           // [E][N] the 'llm' strategy's next action (cfg.llm_action)
    // reference-shaped host I/O (swarm_step_host): library-owned step outputs on the device, the export block on the
    // device, two pinned host copies of it (ping-pong: the previous step's arrays stay valid for one more step), a pinned
    // staging buffer for the action
    void *d_io_obs, *d_io_prior; float *d_io_rew; uint8_t *d_io_done;
    double *d_io_block, *h_io_block[2];
    void *h_io_action, *d_io_action;
    size_t io_block_bytes;
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

// Is this cell list a row-major subset of a square lattice (<= 64 x 64)?  Fills `L` (geometry only) if so.
bool detect_lattice(const double *gx, const double *gy, int n, LatEnv &L)
{
    if (n < 2) return false;
    // lattice step: the closest pair among consecutive cells (cells of one row are consecutive and one step apart)
    double l2 = INFINITY; int k0 = -1;
    for (int c = 0; c + 1 < n; ++c) {
        const double dx = gx[c + 1] - gx[c], dy = gy[c + 1] - gy[c], d2 = dx * dx + dy * dy;
        if (d2 < l2) { l2 = d2; k0 = c; }
    }
    if (!(l2 > 0) || k0 < 0) return false;
    // the row direction u is one of the (at most four) distinct unit-step directions between consecutive cells
    // (single-cell rows make consecutive cells vertical neighbours): try each until the order is row-major
    double cand[4][2]; int ncand = 0;
    for (int c = 0; c + 1 < n && ncand < 4; ++c) {
        const double dx = gx[c + 1] - gx[c], dy = gy[c + 1] - gy[c], d2 = dx * dx + dy * dy;
        if (d2 > l2 * (1.0 + 1e-6)) continue;
        bool seen = false;
        for (int q = 0; q < ncand; ++q)
            if (std::fabs(cand[q][0] - dx) + std::fabs(cand[q][1] - dy) < 1e-6 * std::sqrt(l2)) seen = true;
        if (!seen) { cand[ncand][0] = dx; cand[ncand][1] = dy; ++ncand; }
    }
    std::vector<int> ai((size_t)n), bi((size_t)n);
    double ux = 0, uy = 0, vx = 0, vy = 0;
    bool found = false;
    for (int q = 0; q < ncand && !found; ++q) {
        ux = cand[q][0]; uy = cand[q][1]; vx = -uy; vy = ux;
        for (int pass = 0; pass < 2 && !found; ++pass) {
            bool ok = true;
            for (int c = 0; c < n && ok; ++c) {
                const double rx = gx[c] - gx[0], ry = gy[c] - gy[0];
                const double a = (rx * ux + ry * uy) / l2, b = (rx * vx + ry * vy) / l2;
                const double ar = std::nearbyint(a), br = std::nearbyint(b);
                if (std::fabs(a - ar) > 1e-6 || std::fabs(b - br) > 1e-6 || std::fabs(ar) > 4096 || std::fabs(br) > 4096) ok = false;
                ai[(size_t)c] = (int)ar; bi[(size_t)c] = (int)br;
            }
            if (!ok) break;
            // row-major order: within a row the column increases, rows increase
            bool order = true, flip = false;
            for (int c = 0; c + 1 < n; ++c) {
                if (bi[(size_t)c + 1] == bi[(size_t)c]) { if (ai[(size_t)c + 1] <= ai[(size_t)c]) order = false; }
                else if (bi[(size_t)c + 1] < bi[(size_t)c]) { flip = true; order = false; }
            }
            if (order) { found = true; break; }
            if (pass == 0 && flip) { vx = -vx; vy = -vy; continue; }      // rows run the other way: mirror v
            break;
        }
    }
    if (!found) return false;
    int amin = ai[0], amax = ai[0], bmin = bi[0], bmax = bi[0];
    for (int c = 0; c < n; ++c) {
        amin = std::min(amin, ai[(size_t)c]); amax = std::max(amax, ai[(size_t)c]);
        bmin = std::min(bmin, bi[(size_t)c]); bmax = std::max(bmax, bi[(size_t)c]);
    }
    if (amax - amin + 1 > 64 || bmax - bmin + 1 > 64) return false;
    std::memset(&L, 0, sizeof(L));
    L.ncols = amax - amin + 1; L.nrows = bmax - bmin + 1;
    for (int b = 0; b < 64; ++b) L.rowstart[b] = 0;
    int prev_b = -1;
    for (int c = 0; c < n; ++c) {
        const int a = ai[(size_t)c] - amin, b = bi[(size_t)c] - bmin;
        if (b != prev_b) { if (b < prev_b) return false; L.rowstart[b] = (short)c; prev_b = b; }
        if (L.rowmask[b] & (1ull << a)) return false;
        L.rowmask[b] |= 1ull << a;
    }
    L.ox = gx[0] - (ai[0] - amin) * ux - (bi[0] - bmin) * vx;
    L.oy = gy[0] - (ai[0] - amin) * uy - (bi[0] - bmin) * vy;
    L.uxi = ux / l2; L.uyi = uy / l2; L.vxi = vx / l2; L.vyi = vy / l2;
    const double l = std::sqrt(l2);
    L.R = (float)l;                 // caller turns the step length into radii
    return true;
}

template <int NPAD, bool HALF>
void layout_t(KP &k)
{
    typedef Geo<NPAD, HALF> G_;
    constexpr int AG = G_::AG, EPB = G_::EPB, NW = G_::NW, WPE = G_::WPE, T = G_::T;
    k.ngw = (k.ng_max + 31) / 32;
    k.cxy_stride = k.ngw * 32 + 1;            // +1 pair: envs of one wave start on different LDS banks
    int half = (k.g_max + 1) / 2;
    if ((half & 1) == 0) ++half;              // odd dword stride: lane-per-row int16 writes spread over banks
    k.g_stride = 2 * half;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 15) & ~size_t(15); return (int)o; };
    auto max2 = [](size_t a, size_t b) { return a > b ? a : b; };
    const size_t pm_bytes = (size_t)(NW == 1 ? WPE * 4 : 5) * NW * AG * 8;   // partial pair masks (per-split copies for N <= 64, one accumulator set above)
    k.off_cxy = 0;                             // fp64 cells are no longer staged in LDS
    k.off_sp = take((size_t)4 * AG * 8);
    k.cxq_stride = k.ngw * 64 + 4;             // floats: 2 per cell, +1 pair-of-pairs of padding
    if (k.lattice) {
        // row-space lattice path: no per-cell bit sets at all
        constexpr int NRC = 16;
        k.off_hdr = take((size_t)AG * 16);
        k.off_srow = take(max2((size_t)(NRC - 1) * AG * 4, (size_t)NW * 1536));   // window-row words (lat_nrs <= 15 rows) | scratch of the exact reward
        k.off_pcr = take((size_t)AG * NRC);
        k.off_sidx = take(max2((size_t)AG * k.g_stride * 2, pm_bytes));      // sidx | pm
        k.off_partc = take((size_t)WPE * AG * 2);                             // nearest-cell candidates | partial ranks (16-bit)
        k.off_partd = take((size_t)(WPE - 1) * AG * 8);                       // the walking splits' exact squared distances | perm
        k.off_lat = take((size_t)EPB * 64 * (8 + 2));
        k.off_cov = take((size_t)EPB * 64 * 8);
        k.off_flag = take((size_t)AG);                                        // one byte per agent thread
        k.off_snei = take((size_t)AG * kNeiStride * 2);
        k.off_sncf = take((size_t)AG * 4);
        k.off_snear = take(NW > 1 ? (size_t)NW * AG * 8 : 0);                 // (N <= 64: the nearby mask stays in a register)
        k.off_perm = k.off_partd;                                             // T bytes, written two barriers after the distances were consumed
        static_assert((size_t)T <= (size_t)(WPE - 1) * AG * 8, "perm must fit the distance array it reuses");
        k.off_rres = 0;
        k.smem_lat = (int)off;                           // lattice mode, no export
        k.off_orow = take((size_t)NRC * AG * 4);         // only launches that export the index scratch use it
        k.smem_lat_export = (int)off;
        k.off_cmask = k.off_sbits = k.off_obits = k.off_pc = k.off_cxyf = 0;     // generic scan only
        k.smem_generic = 0;
        return;
    }
    k.off_cmask = take(max2(max2((size_t)k.ngw * 32 * NW * 8, (size_t)WPE * 3 * AG * 4 + (NW > 1 ? NW * 1536 : 0)), (size_t)(k.ngw + 1) * AG * 4));   // cmask | rsel | rsum
    k.off_sbits = take((size_t)(k.ngw + 1) * AG * 4);
    k.off_sidx = take(max2(max2((size_t)AG * k.g_stride * 2, pm_bytes), (size_t)14 * AG * 8));
    k.off_partc = take((size_t)WPE * AG * 4);
    k.off_lat = take((size_t)EPB * 64 * (8 + 2));
    k.off_cov = take((size_t)EPB * (k.ngw + 1) * 4);
    k.off_flag = take((size_t)AG * 4);
    k.off_snei = take((size_t)AG * kNeiStride * 2);
    k.off_sncf = take((size_t)AG * 4);
    k.off_snear = take((size_t)NW * AG * 8);
    k.off_pc = take((size_t)k.ngw * AG);
    k.off_obits = take((size_t)k.ngw * AG * 4);
    k.off_cxyf = take((size_t)EPB * k.cxq_stride * 4);   // fp32 cell copy of the generic scan
    k.smem_generic = (int)off;
    k.smem_lat = k.smem_lat_export = 0;
    k.off_hdr = k.off_srow = k.off_pcr = k.off_perm = k.off_rres = k.off_orow = k.off_partd = 0;
}

void layout(KP &k, int npad, bool half)
{
    switch (npad) {
    case 8: if (half) layout_t<8, true>(k); else layout_t<8, false>(k); break;
    case 16: if (half) layout_t<16, true>(k); else layout_t<16, false>(k); break;
    case 32: if (half) layout_t<32, true>(k); else layout_t<32, false>(k); break;
    case 64: layout_t<64, false>(k); break;
    case 128: layout_t<128, false>(k); break;
    default: layout_t<256, false>(k); break;
    }
}

// Decide the cell path of the next launches and lay out its LDS.  The row-space lattice path needs every env's cells to
// be a lattice subset AND a sensing window of at most 15 lattice rows (d_sen < ~7.5 cells: a window row then has at
// most 17 columns -- one 32-bit word -- and a list at most 240 cells -- one byte per row count); anything else takes the
// generic scan, which handles arbitrary cell sets.
void set_lattice_mode(swarm_env *h, bool all_lattice, float rmax, float cmax, int ncols_max)
{
    KP &k = h->kp;
    k.lat_n32 = ncols_max <= 32 ? 1 : 0;
    k.lat_rw = (int)std::ceil(rmax + 0.02f);
    k.lat_cw = (int)std::ceil(cmax + 0.02f);
    k.lat_nrs = (int)std::floor(2.0f * (rmax + 0.01f)) + 1; k.lat_nrc = (int)std::floor(2.0f * (cmax + 0.01f)) + 1;
    k.lattice = (all_lattice && !h->lattice_disabled && k.lat_nrs <= 15) ? 1 : 0;
    {   // guard band of the fp32 reward decision of the lattice path, in lattice steps (R = d_sen / l <= rmax).  Per cell the
        // model coordinate relative to the agent is off by dx: lattice fit tolerance 1.5e-6 steps, fp32 cast of the relative
        // coordinate (|.| <= 17 steps) 2.1e-6, the walk's scaled form c / R - a / R (two products of magnitude <= 17 / R with
        // a 1-ulp reciprocal, cancelling) 5e-6, margin: 1.2e-5.  As in swarm_create: psi is off by
        // dpsi <= (pi^2 / 4) (2 sqrt(2) dx / R) + 1.2e-6, |v| by n (dpsi R + dx + thr dpsi) / den, thr = 0.05 R / d_sen;
        // fp32 accumulation / division / sqrt: 4e-6 relative to d_sen, i.e. 4e-6 R / d_sen steps.  1.3x margin.
        const double dx = 1.2e-5, R = std::fmax(1.0, (double)rmax), thr = 0.05 * R / k.d_sen;
        const double dpsi_R = 2.4675 * 2.0 * std::sqrt(2.0) * dx + 1.2e-6 * R;      // dpsi * R (1.2e-6: degree-5 polynomial 6.5e-7 + the 1-ulp reciprocal scaling)
        k.rew_ga_lat = (float)(1.3 * (dpsi_R + dx + thr * dpsi_R / R));
        k.rew_gb_lat = (float)(4e-6 * R / k.d_sen);
    }
    // a small batch of small environments (N < 64) that leaves at least half of the chip's workgroup slots empty: the
    // half-occupied geometry (Geo<NPAD, true>) -- twice the workgroups, eight lanes per agent in the list phase
    {
        const int epb_full = h->npad < 64 ? 64 / h->npad : 1;
        const long long grid_full = ((long long)h->cfg.n_env + epb_full - 1) / epb_full;
        h->half = k.lattice && h->npad < 64 && epb_full >= 2 && !(h->cfg.debug_flags & 4) && 2 * grid_full <= (long long)h->n_cu * 6;
    }
    layout(k, h->npad, h->half);
}

template <int NPAD, typename OT, bool DO_STEP, bool LAT, bool HALF>
int launch_t(swarm_env *h, const void *action, int act_f64, void *obs, float *reward, uint8_t *done, void *a_prior)
{
    constexpr int T = Geo<NPAD, HALF>::T, EPB = Geo<NPAD, HALF>::EPB;
    auto kern = k_env<NPAD, OT, DO_STEP, LAT, HALF>;
    int smem = !LAT ? h->kp.smem_generic : (h->kp.export_idx ? h->kp.smem_lat_export : h->kp.smem_lat);
#ifdef SWARM_EXTRA_SMEM
    smem += SWARM_EXTRA_SMEM;                            // occupancy experiments only
#endif
    int &attr = h->attr_smem[(DO_STEP ? 1 : 0) + (sizeof(OT) == 8 ? 2 : sizeof(OT) == 2 ? 4 : 0) + (LAT ? 6 : 0) + (HALF ? 12 : 0)];   // raise the dynamic-LDS cap once per instantiation
    if (attr < smem) {
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr = smem;
    }
    const int grid = (h->cfg.n_env + EPB - 1) / EPB;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(T), smem, h->stream, h->kp, action, act_f64,
                       static_cast<OT *>(obs), reward, done, static_cast<OT *>(a_prior));
    HIP_TRY(h, hipGetLastError());
    return SWARM_OK;
}

template <int NPAD, typename OT, bool DO_STEP>
int launch_l(swarm_env *h, const void *action, int act_f64, void *obs, float *reward, uint8_t *done, void *a_prior)
{
    if constexpr (NPAD < 64) {
        if (h->kp.lattice && h->half) return launch_t<NPAD, OT, DO_STEP, true, true>(h, action, act_f64, obs, reward, done, a_prior);
    }
    return h->kp.lattice ? launch_t<NPAD, OT, DO_STEP, true, false>(h, action, act_f64, obs, reward, done, a_prior)
                         : launch_t<NPAD, OT, DO_STEP, false, false>(h, action, act_f64, obs, reward, done, a_prior);
}

template <int NPAD>
int launch_n(swarm_env *h, bool do_step, const void *action, int act_f64, void *obs, float *reward, uint8_t *done,
             void *a_prior)
{
    const int dt = h->cfg.obs_dtype;
    if (do_step) {
        return dt == SWARM_F64 ? launch_l<NPAD, double, true>(h, action, act_f64, obs, reward, done, a_prior)
             : dt == SWARM_BF16 ? launch_l<NPAD, __bf16, true>(h, action, act_f64, obs, reward, done, a_prior)
                                : launch_l<NPAD, float, true>(h, action, act_f64, obs, reward, done, a_prior);
    }
    return dt == SWARM_F64 ? launch_l<NPAD, double, false>(h, action, act_f64, obs, reward, done, a_prior)
         : dt == SWARM_BF16 ? launch_l<NPAD, __bf16, false>(h, action, act_f64, obs, reward, done, a_prior)
                            : launch_l<NPAD, float, false>(h, action, act_f64, obs, reward, done, a_prior);
}

int launch(swarm_env *h, bool do_step, const void *action, int act_f64, void *obs, float *reward, uint8_t *done,
           void *a_prior)
{
    // (SWARM_ONLY_NPAD: development builds with a single instantiation -- quick register / code-size checks and A/B runs)
#ifndef SWARM_ONLY_NPAD
#define SWARM_ONLY_NPAD 0
#endif
#define SWARM_CASE(n) case n: if (SWARM_ONLY_NPAD == 0 || SWARM_ONLY_NPAD == n) return launch_n<(SWARM_ONLY_NPAD == 0 || SWARM_ONLY_NPAD == n) ? n : (SWARM_ONLY_NPAD)>(h, do_step, action, act_f64, obs, reward, done, a_prior); break
    switch (h->npad) {
    SWARM_CASE(8); SWARM_CASE(16); SWARM_CASE(32); SWARM_CASE(64); SWARM_CASE(128); SWARM_CASE(256);
    }
#undef SWARM_CASE
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
    c->prior_gain[0] = 2.0; c->prior_gain[1] = 3.0; c->prior_gain[2] = 2.0;      // AssemblyEnv.cpp:1128-1132
    c->llm_repulsion = 1.0; c->llm_action = 0;                                   // assembly.py:895
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
    if (cfg->obs_dtype != SWARM_F32 && cfg->obs_dtype != SWARM_F64 && cfg->obs_dtype != SWARM_BF16) return fail(nullptr, SWARM_ERR_INVALID, "obs_dtype must be SWARM_F32, SWARM_F64 or SWARM_BF16");
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
    for (int &a : h->attr_smem) a = -1;
    h->half = false; h->n_cu = 256;
    h->d_p = h->d_dp = h->d_cells = h->d_cin = nullptr; h->d_cells_xy = nullptr;
    h->d_lat = nullptr;
    h->n_shapes = 0; h->d_shape_cells = h->d_shape_l = h->d_shape_cin = nullptr; h->d_shape_ng = nullptr; h->d_shape_lat = nullptr; h->d_shape_idx = nullptr;
    h->shapes_lattice = false; h->shapes_rmax = h->shapes_cmax = 0.0f; h->shapes_ncols = 0;
    h->lat_ok.assign((size_t)cfg->n_env, 0);
    h->lat_R.assign((size_t)cfg->n_env, 0.0f); h->lat_Rc.assign((size_t)cfg->n_env, 0.0f);
    h->lat_ncols.assign((size_t)cfg->n_env, 0);
    h->lattice_disabled = (cfg->debug_flags & 2) != 0;
    h->d_nei = h->d_near = h->d_inflag = h->d_ng = h->d_exp_sensed = h->d_exp_occ = nullptr; h->d_sf = nullptr; h->d_prior = nullptr;
    h->d_act_next = nullptr;
    h->d_io_obs = h->d_io_prior = nullptr; h->d_io_rew = nullptr; h->d_io_done = nullptr;
    h->d_io_block = nullptr; h->h_io_block[0] = h->h_io_block[1] = nullptr; h->h_io_action = h->d_io_action = nullptr; h->io_block_bytes = 0;
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
    k.pk_att = cfg->prior_gain[0]; k.pk_rep = cfg->prior_gain[1]; k.pk_ali = cfg->prior_gain[2];
    k.pk_llm = cfg->llm_repulsion; k.llm = cfg->llm_action ? 1 : 0;
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
    k.c_close2 = std::fmin(k.c_sen, (3.0 * k.r_avoid) * (3.0 * k.r_avoid));
    k.c_close = std::fmin(k.c_sen, (1.9 * k.r_avoid) * (1.9 * k.r_avoid));   // 1.9: fewest insertion trips on the 64-agent workload (measured)
    {   // fp32 pre-filter bands.  With |coordinates| <= S, a float-converted coordinate is off by <= 2^-24 S and
        // their float difference by another 2^-24 S at most: dr = 4 * 2^-24 * S bounds each component of the fp32
        // relative position (1.33x margin).  Then |d2_32 - d2_64| <= 2 sqrt(2) |r| dr + O(2^-23 d2) <= 3 sqrt(d2) dr + 2^-21 d2.
        double S = 0.0;
        for (int q = 0; q < 4; ++q) S = std::fmax(S, std::fabs(cfg->boundary[q]));
        S = 1.5 * S + 1.0;
        const double dr = 4.0 * std::ldexp(1.0, -24) * S;
        auto band = [&](double c) { return 3.0 * std::sqrt(c) * dr + std::ldexp(1.0, -21) * c + 1e-12; };
        auto f_below = [](double v) { float f = (float)v; while ((double)f > v) f = std::nextafterf(f, -INFINITY); return f; };
        auto f_above = [](double v) { float f = (float)v; while ((double)f < v) f = std::nextafterf(f, INFINITY); return f; };
        k.csen_lo = f_below(k.c_sen - band(k.c_sen)); k.csen_hi = f_above(k.c_sen + band(k.c_sen));
        k.cocc_lo = f_below(k.c_occ - band(k.c_occ)); k.cocc_hi = f_above(k.c_occ + band(k.c_occ));
        k.coord_lim = (float)S;
        k.min_tol_a = (float)(2.0 * 3.0 * dr); k.min_tol_b = (float)(2.0 * std::ldexp(1.0, -21));
        {   // error bound of the fp32 reward sums near the 0.05 threshold, v = |sum psi r| / sum psi over n list entries:
            // each fp32 component of r is off by dx (two float conversions + the subtraction), u = |r|^2 / d_sen^2 by
            // du <= 2 sqrt(2) dx / d_sen, psi by |dpsi/du| du + the polynomial's 4e-7 with |dpsi/du| <= pi^2 / 4; hence
            // |dv| <= n (dpsi d_sen + dx + 0.05 dpsi) / den + (fp32 accumulation, division, sqrt: < 3e-6).  1.3x margin.
            const double dx = 2.1 * std::ldexp(1.0, -24) * S;
            const double dpsi = 2.4675 * (2.0 * std::sqrt(2.0) * dx / k.d_sen) + 4e-7;
            k.rew_ga = (float)(1.3 * (dpsi * k.d_sen + dx + 0.0505 * dpsi));
            k.rew_gb = 4e-6f;
        }
        k.rew_thr_k = (float)(0.05 / k.d_sen);
        k.rew_ga_lat = k.rew_gb_lat = 0.0f;
        k.force_exact = (cfg->debug_flags & 1) ? 1 : 0;
        {   // unsigned division by D = 2 (G-1) (Granlund-Montgomery round-up method, exact for every 32-bit x)
            const unsigned D = 2u * (unsigned)(k.g_max - 1);
            int l = 0;
            while ((1ull << l) < D) ++l;
            k.cap_magic = (unsigned)((((1ull << l) - D) << 32) / D + 1);
            k.cap_shift = l - 1;
            // numerators stay below 2^32: 2 (G-1) (n_cells_max-1) + (G-1)
            const unsigned long long xmax = 2ull * (k.g_max - 1) * (unsigned long long)k.ng_max + k.g_max;
            k.cap_int = (((k.g_max - 1) & 1) == 1 && xmax < (1ull << 32) && l >= 1) ? 1 : 0;
        }
        k.dbg_phase = (cfg->debug_flags >> 8) & 0xF;
        k.dbg_extra = (cfg->debug_flags >> 12) & 0xF;
    }
    layout(k, h->npad, false);

    DeviceGuard g(dev);
    if (!g.ok) { delete h; return fail(nullptr, SWARM_ERR_HIP, "hipSetDevice failed"); }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { delete h; return fail(nullptr, SWARM_ERR_HIP, "hipGetDeviceProperties failed"); }
    h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if ((size_t)k.smem_generic > 160 * 1024) {
        delete h;
        return fail(nullptr, SWARM_ERR_INVALID, "configuration needs more LDS per workgroup than the device has (reduce n_cells_max / num_obs_grid_max)");
    }
    const size_t E = (size_t)cfg->n_env, N = (size_t)cfg->n_agents;
    hipError_t a = hipSuccess;
    auto alloc = [&](void **p, size_t bytes) { if (a == hipSuccess) a = hipMalloc(p, bytes); };
    alloc((void **)&h->d_p, E * 2 * N * 8); alloc((void **)&h->d_dp, E * 2 * N * 8);
    alloc((void **)&h->d_cells, E * 2 * (size_t)k.ng_max * 8); alloc((void **)&h->d_cin, E * 8);
    alloc((void **)&h->d_cells_xy, E * (size_t)k.ng_max * 16);
    alloc((void **)&h->d_ng, E * 4); alloc((void **)&h->d_shape_idx, E * 4);
    alloc((void **)&h->d_prior, E * N * 16);
    if (cfg->llm_action) alloc((void **)&h->d_act_next, E * N * 16);
    alloc((void **)&h->d_lat, E * sizeof(LatEnv));
    alloc((void **)&h->d_nei, E * N * (size_t)k.topo * 4); alloc((void **)&h->d_near, E * N * 4);
    alloc((void **)&h->d_inflag, E * N * 4); alloc((void **)&h->d_sf, E * N * 16);
    if (a == hipSuccess) a = hipMemset(h->d_ng, 0, E * 4);
    if (a == hipSuccess) a = hipMemset(h->d_prior, 0, E * N * 16);
    if (a == hipSuccess && h->d_act_next) a = hipMemset(h->d_act_next, 0, E * N * 16);
    if (a == hipSuccess) a = hipMemset(h->d_shape_idx, 0xFF, E * 4);
    if (a == hipSuccess) a = hipMemset(h->d_nei, 0xFF, E * N * (size_t)k.topo * 4);
    if (a == hipSuccess) a = hipMemset(h->d_near, 0, E * N * 4);
    if (a == hipSuccess) a = hipMemset(h->d_inflag, 0, E * N * 4);
    if (a == hipSuccess) a = hipMemset(h->d_sf, 0, E * N * 16);
    if (a == hipSuccess) a = hipMemset(h->d_cells, 0, E * 2 * (size_t)k.ng_max * 8);
    if (a == hipSuccess) a = hipMemset(h->d_cells_xy, 0, E * (size_t)k.ng_max * 16);
    if (a == hipSuccess) a = hipEventCreate(&h->ev0);
    if (a == hipSuccess) a = hipEventCreate(&h->ev1);
    if (a != hipSuccess) {
        std::string m = std::string("device allocation failed: ") + hipGetErrorString(a);
        swarm_destroy(h);
        return fail(nullptr, SWARM_ERR_HIP, m);
    }
    k.p = h->d_p; k.dp = h->d_dp; k.nei = h->d_nei; k.near_cell = h->d_near; k.in_flag = h->d_inflag; k.sf_next = h->d_sf;
    k.prior_next = h->d_prior; k.act_next = h->d_act_next;
    k.cells = h->d_cells; k.cells_xy = h->d_cells_xy; k.n_g = h->d_ng; k.c_in = h->d_cin;
    k.lat = h->d_lat; k.lattice = 0; k.lat_rw = k.lat_cw = 0; k.lat_nrs = k.lat_nrc = 0; k.lat_n32 = 0;
    k.c_near_hi = k.c_near * (1.0 + 1e-9);
    *out = h;
    return SWARM_OK;
}

int swarm_destroy(swarm_env_t *h)
{
    if (!h) return SWARM_OK;
    {
        DeviceGuard g(h->device);
        (void)hipStreamSynchronize(h->stream);
        (void)hipFree(h->d_p); (void)hipFree(h->d_dp); (void)hipFree(h->d_cells); (void)hipFree(h->d_cin); (void)hipFree(h->d_cells_xy);
        (void)hipFree(h->d_ng); (void)hipFree(h->d_nei); (void)hipFree(h->d_near); (void)hipFree(h->d_inflag); (void)hipFree(h->d_sf);
        (void)hipFree(h->d_exp_sensed); (void)hipFree(h->d_exp_occ); (void)hipFree(h->d_lat); (void)hipFree(h->d_shape_idx); (void)hipFree(h->d_prior);
        (void)hipFree(h->d_shape_cells); (void)hipFree(h->d_shape_l); (void)hipFree(h->d_shape_cin); (void)hipFree(h->d_shape_ng); (void)hipFree(h->d_shape_lat);
        (void)hipFree(h->d_act_next); (void)hipFree(h->d_io_obs); (void)hipFree(h->d_io_prior); (void)hipFree(h->d_io_rew); (void)hipFree(h->d_io_done);
        (void)hipFree(h->d_io_block); (void)hipFree(h->d_io_action);
        if (h->h_io_block[0]) (void)hipHostFree(h->h_io_block[0]);
        if (h->h_io_block[1]) (void)hipHostFree(h->h_io_block[1]);
        if (h->h_io_action) (void)hipHostFree(h->h_io_action);
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
    {
        const size_t n = (size_t)count * h->kp.ng_max;
        hipLaunchKernelGGL(k_interleave, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, h->stream,
                           h->d_cells, h->d_cells_xy, h->kp.ng_max, env_begin, count);
        HIP_TRY(h, hipGetLastError());
    }
    HIP_TRY(h, hipMemcpyAsync(h->d_ng + env_begin, n_g, (size_t)count * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->d_shape_idx + env_begin, 0xFF, (size_t)count * 4, h->stream));   // no longer a shape of the set
    HIP_TRY(h, hipMemcpyAsync(h->d_cin + env_begin, cin.data(), (size_t)count * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));           // cin is a host temporary
    {   // lattice detection on a host copy of what was uploaded (`cells` may be a device pointer)
        std::vector<double> hc((size_t)count * row);
        HIP_TRY(h, hipMemcpy(hc.data(), h->d_cells + (size_t)env_begin * row, (size_t)count * row * 8, hipMemcpyDeviceToHost));
        std::vector<LatEnv> lat((size_t)count);
        for (int k = 0; k < count; ++k) {
            const double *gx = hc.data() + (size_t)k * row, *gy = gx + h->kp.ng_max;
            LatEnv &L = lat[(size_t)k];
            std::memset(&L, 0, sizeof(L));
            const bool ok = !h->lattice_disabled && detect_lattice(gx, gy, n_g[k], L);
            h->lat_ok[(size_t)(env_begin + k)] = ok ? 1 : 0;
            if (ok) {
                const double l = L.R;
                L.R = (float)(h->kp.d_sen / l); L.Rc = (float)((h->kp.r_avoid / 2.0) / l);
                h->lat_R[(size_t)(env_begin + k)] = L.R; h->lat_Rc[(size_t)(env_begin + k)] = L.Rc;
                h->lat_ncols[(size_t)(env_begin + k)] = L.ncols;
            }
        }
        HIP_TRY(h, hipMemcpy(h->d_lat + env_begin, lat.data(), (size_t)count * sizeof(LatEnv), hipMemcpyHostToDevice));
        bool all = true; float rmax = 0.0f, cmax = 0.0f; int ncmax = 0;
        for (int e2 = 0; e2 < h->cfg.n_env; ++e2) {
            if (!h->lat_ok[(size_t)e2]) { all = false; break; }
            rmax = std::max(rmax, h->lat_R[(size_t)e2]); cmax = std::max(cmax, h->lat_Rc[(size_t)e2]);
            ncmax = std::max(ncmax, h->lat_ncols[(size_t)e2]);
        }
        set_lattice_mode(h, all, rmax, cmax, ncmax);
    }
    for (int k = 0; k < count; ++k) h->cells_set[(size_t)(env_begin + k)] = 1;
    h->have_cells = true;
    for (char c : h->cells_set) if (!c) { h->have_cells = false; break; }
    h->observed = false;
    return SWARM_OK;
}

int swarm_set_shapes(swarm_env_t *h, int n_shapes, const double *shape_cells, const int32_t *n_g, const double *l_cell)
{
    if (!h) return SWARM_ERR_INVALID;
    if (n_shapes < 1 || !shape_cells || !n_g || !l_cell) return fail(h, SWARM_ERR_INVALID, "swarm_set_shapes: bad argument");
    const size_t row = (size_t)2 * h->kp.ng_max;
    std::vector<double> cin((size_t)n_shapes);
    std::vector<LatEnv> lat((size_t)n_shapes);
    bool all = true; float rmax = 0.0f, cmax = 0.0f; int ncmax = 0;
    for (int k = 0; k < n_shapes; ++k) {
        if (n_g[k] < 1 || n_g[k] > h->cfg.n_cells_max) return fail(h, SWARM_ERR_INVALID, "swarm_set_shapes: n_g must be in [1, n_cells_max]");
        if (!(l_cell[k] > 0)) return fail(h, SWARM_ERR_INVALID, "swarm_set_shapes: l_cell must be positive");
        cin[(size_t)k] = cut_lt(std::sqrt(2) * l_cell[k] / 2);
        LatEnv &L = lat[(size_t)k];
        std::memset(&L, 0, sizeof(L));
        const double *gx = shape_cells + (size_t)k * row, *gy = gx + h->kp.ng_max;
        if (!h->lattice_disabled && detect_lattice(gx, gy, n_g[k], L)) {
            const double l = L.R;
            L.R = (float)(h->kp.d_sen / l); L.Rc = (float)((h->kp.r_avoid / 2.0) / l);
            rmax = std::max(rmax, L.R); cmax = std::max(cmax, L.Rc); ncmax = std::max(ncmax, L.ncols);
        } else { std::memset(&L, 0, sizeof(L)); all = false; }
    }
    DeviceGuard g(h->device);
    (void)hipFree(h->d_shape_cells); (void)hipFree(h->d_shape_l); (void)hipFree(h->d_shape_cin); (void)hipFree(h->d_shape_ng); (void)hipFree(h->d_shape_lat);
    h->d_shape_cells = h->d_shape_l = h->d_shape_cin = nullptr; h->d_shape_ng = nullptr; h->d_shape_lat = nullptr;
    HIP_TRY(h, hipMalloc((void **)&h->d_shape_cells, (size_t)n_shapes * row * 8));
    HIP_TRY(h, hipMalloc((void **)&h->d_shape_l, (size_t)n_shapes * 8));
    HIP_TRY(h, hipMalloc((void **)&h->d_shape_cin, (size_t)n_shapes * 8));
    HIP_TRY(h, hipMalloc((void **)&h->d_shape_ng, (size_t)n_shapes * 4));
    HIP_TRY(h, hipMalloc((void **)&h->d_shape_lat, (size_t)n_shapes * sizeof(LatEnv)));
    HIP_TRY(h, hipMemcpy(h->d_shape_cells, shape_cells, (size_t)n_shapes * row * 8, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_shape_l, l_cell, (size_t)n_shapes * 8, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_shape_cin, cin.data(), (size_t)n_shapes * 8, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_shape_ng, n_g, (size_t)n_shapes * 4, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_shape_lat, lat.data(), (size_t)n_shapes * sizeof(LatEnv), hipMemcpyHostToDevice));
    h->n_shapes = n_shapes; h->shapes_lattice = all; h->shapes_rmax = rmax; h->shapes_cmax = cmax; h->shapes_ncols = ncmax;
    return SWARM_OK;
}

int swarm_reset(swarm_env_t *h, uint64_t seed, uint64_t episode, int64_t env_offset, void *obs)
{
    if (!h) return SWARM_ERR_INVALID;
    if (h->n_shapes < 1) return fail(h, SWARM_ERR_STATE, "swarm_reset: no shape set (swarm_set_shapes)");
    DeviceGuard g(h->device);
    ShapeSet S;
    S.n_shapes = h->n_shapes; S.cells = h->d_shape_cells; S.n_g = h->d_shape_ng; S.l_cell = h->d_shape_l;
    S.c_in = h->d_shape_cin; S.lat = h->d_shape_lat;
    hipLaunchKernelGGL(k_reset, dim3(h->cfg.n_env), dim3(256), 0, h->stream, h->kp, S, (unsigned long long)seed,
                       (unsigned long long)episode, (long long)env_offset, h->d_cells, h->d_ng, h->d_cin, h->d_lat, h->d_shape_idx);
    HIP_TRY(h, hipGetLastError());
    {
        const size_t n = (size_t)h->cfg.n_env * h->kp.ng_max;
        hipLaunchKernelGGL(k_interleave, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, h->stream,
                           h->d_cells, h->d_cells_xy, h->kp.ng_max, 0, h->cfg.n_env);
        HIP_TRY(h, hipGetLastError());
    }
    std::fill(h->cells_set.begin(), h->cells_set.end(), 1);
    std::fill(h->lat_ok.begin(), h->lat_ok.end(), h->shapes_lattice ? 1 : 0);
    // per-env bounds for a later partial swarm_set_cells: the shape set's maxima are valid for every env
    std::fill(h->lat_R.begin(), h->lat_R.end(), h->shapes_rmax); std::fill(h->lat_Rc.begin(), h->lat_Rc.end(), h->shapes_cmax);
    std::fill(h->lat_ncols.begin(), h->lat_ncols.end(), h->shapes_ncols);
    h->have_cells = h->have_state = true;
    set_lattice_mode(h, h->shapes_lattice, h->shapes_rmax, h->shapes_cmax, h->shapes_ncols);
    h->observed = false;
    return swarm_observe(h, obs);
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

int swarm_metrics(swarm_env_t *h, double *out)
{
    if (!h || !out) return SWARM_ERR_INVALID;
    if (!h->have_cells || !h->have_state) return fail(h, SWARM_ERR_STATE, "swarm_metrics: cells / state not set");
    DeviceGuard g(h->device);
    const size_t smem = (size_t)h->cfg.n_agents * (4 * 8 + 4) + 16;
    hipLaunchKernelGGL(k_metrics, dim3(h->cfg.n_env), dim3(256), smem, h->stream, h->kp, out);
    HIP_TRY(h, hipGetLastError());
    return SWARM_OK;
}

int swarm_get_cells(swarm_env_t *h, double *cells, int32_t *n_g)
{
    if (!h) return SWARM_ERR_INVALID;
    DeviceGuard g(h->device);
    if (cells) HIP_TRY(h, hipMemcpyAsync(cells, h->d_cells, (size_t)h->cfg.n_env * 2 * h->kp.ng_max * 8, hipMemcpyDefault, h->stream));
    if (n_g) HIP_TRY(h, hipMemcpyAsync(n_g, h->d_ng, (size_t)h->cfg.n_env * 4, hipMemcpyDefault, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SWARM_OK;
}

int swarm_get_shape_index(swarm_env_t *h, int32_t *shape_index)
{
    if (!h || !shape_index) return SWARM_ERR_INVALID;
    DeviceGuard g(h->device);
    HIP_TRY(h, hipMemcpyAsync(shape_index, h->d_shape_idx, (size_t)h->cfg.n_env * 4, hipMemcpyDefault, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
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
    if (!action) {
        // agent_strategy == 'llm' (assembly.py:525-529): the action is the Python twin of the prior policy, which the
        // previous pass evaluated on this very state
        if (!h->cfg.llm_action) return fail(h, SWARM_ERR_INVALID, "swarm_step: null action (only a handle created with llm_action may pass NULL)");
        action = h->d_act_next; action_dtype = SWARM_F64;
    }
    if (action_dtype != SWARM_F32 && action_dtype != SWARM_F64) return fail(h, SWARM_ERR_INVALID, "swarm_step: bad action_dtype");
    if (!h->observed) return fail(h, SWARM_ERR_STATE, "swarm_step: call swarm_observe after setting cells/state (the reference's reset() ends with _get_obs())");
    DeviceGuard g(h->device);
    return launch(h, true, action, action_dtype == SWARM_F64, obs, reward, done, a_prior);
}

namespace {
int io_alloc(swarm_env *h)
{
    if (h->d_io_block) return SWARM_OK;
    const size_t EN = (size_t)h->cfg.n_env * h->cfg.n_agents, D = (size_t)h->kp.obs_dim;
    const size_t so = h->cfg.obs_dtype == SWARM_F64 ? 8 : h->cfg.obs_dtype == SWARM_BF16 ? 2 : 4;
    h->io_block_bytes = (D * EN + 2 * EN + EN) * 8 + ((EN + 15) & ~size_t(15));
    HIP_TRY(h, hipMalloc(&h->d_io_obs, EN * D * so));
    HIP_TRY(h, hipMalloc(&h->d_io_prior, EN * 2 * so));
    HIP_TRY(h, hipMalloc((void **)&h->d_io_rew, EN * 4));
    HIP_TRY(h, hipMalloc((void **)&h->d_io_done, EN));
    HIP_TRY(h, hipMalloc((void **)&h->d_io_block, h->io_block_bytes));
    HIP_TRY(h, hipMalloc(&h->d_io_action, EN * 16));
    HIP_TRY(h, hipHostMalloc((void **)&h->h_io_block[0], h->io_block_bytes, hipHostMallocDefault));
    HIP_TRY(h, hipHostMalloc((void **)&h->h_io_block[1], h->io_block_bytes, hipHostMallocDefault));
    HIP_TRY(h, hipHostMalloc(&h->h_io_action, EN * 16, hipHostMallocDefault));
    HIP_TRY(h, hipMemset(h->d_io_block, 0, h->io_block_bytes));
    std::memset(h->h_io_block[0], 0, h->io_block_bytes); std::memset(h->h_io_block[1], 0, h->io_block_bytes);
    return SWARM_OK;
}

int io_export(swarm_env *h, int slot, bool stepped)
{
    const long long EN = (long long)h->cfg.n_env * h->cfg.n_agents;
    const int D = h->kp.obs_dim, wp = (stepped && h->kp.with_prior) ? 1 : 0;
    const unsigned grid = (unsigned)((EN + 63) / 64);
    const float *rew = stepped ? h->d_io_rew : nullptr;
    const uint8_t *dn = stepped ? h->d_io_done : nullptr;
    if (h->cfg.obs_dtype == SWARM_F64)
        hipLaunchKernelGGL(k_export<double>, dim3(grid), dim3(256), 0, h->stream, (const double *)h->d_io_obs, rew, dn, (const double *)h->d_io_prior, h->d_io_block, D, EN, wp);
    else if (h->cfg.obs_dtype == SWARM_BF16)
        hipLaunchKernelGGL(k_export<__bf16>, dim3(grid), dim3(256), 0, h->stream, (const __bf16 *)h->d_io_obs, rew, dn, (const __bf16 *)h->d_io_prior, h->d_io_block, D, EN, wp);
    else
        hipLaunchKernelGGL(k_export<float>, dim3(grid), dim3(256), 0, h->stream, (const float *)h->d_io_obs, rew, dn, (const float *)h->d_io_prior, h->d_io_block, D, EN, wp);
    HIP_TRY(h, hipGetLastError());
    // obs only (reset / observe) moves the obs part; a step moves the whole block
    const size_t bytes = stepped ? h->io_block_bytes : (size_t)D * EN * 8;
    HIP_TRY(h, hipMemcpyAsync(h->h_io_block[slot], h->d_io_block, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SWARM_OK;
}
}  // namespace

int swarm_get_llm_action(swarm_env_t *h, double *action)
{
    if (!h || !action) return SWARM_ERR_INVALID;
    if (!h->d_act_next) return fail(h, SWARM_ERR_STATE, "swarm_get_llm_action: handle was not created with llm_action");
    if (!h->observed) return fail(h, SWARM_ERR_STATE, "swarm_get_llm_action: nothing observed yet");
    DeviceGuard g(h->device);
    HIP_TRY(h, hipMemcpyAsync(action, h->d_act_next, (size_t)h->cfg.n_env * h->cfg.n_agents * 16, hipMemcpyDefault, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SWARM_OK;
}

int swarm_host_outputs(swarm_env_t *h, int slot, swarm_host_out_t *out)
{
    if (!h || !out || slot < 0 || slot > 1) return SWARM_ERR_INVALID;
    DeviceGuard g(h->device);
    int rc = io_alloc(h);
    if (rc != SWARM_OK) return rc;
    const size_t EN = (size_t)h->cfg.n_env * h->cfg.n_agents, D = (size_t)h->kp.obs_dim;
    double *b = h->h_io_block[slot];
    out->obs = b; out->a_prior = b + D * EN; out->reward = b + D * EN + 2 * EN;
    out->done = reinterpret_cast<uint8_t *>(b + D * EN + 3 * EN);
    return SWARM_OK;
}

int swarm_observe_host(swarm_env_t *h, int slot)
{
    if (!h || slot < 0 || slot > 1) return SWARM_ERR_INVALID;
    DeviceGuard g(h->device);
    int rc = io_alloc(h);
    if (rc != SWARM_OK) return rc;
    rc = swarm_observe(h, h->d_io_obs);
    if (rc != SWARM_OK) return rc;
    return io_export(h, slot, false);
}

int swarm_step_host(swarm_env_t *h, const void *action, int action_dtype, int action_on_device, int slot)
{
    if (!h || slot < 0 || slot > 1) return SWARM_ERR_INVALID;
    if (action && action_dtype != SWARM_F32 && action_dtype != SWARM_F64) return fail(h, SWARM_ERR_INVALID, "swarm_step_host: bad action_dtype");
    if (!h->observed) return fail(h, SWARM_ERR_STATE, "swarm_step_host: call swarm_observe(_host) after setting cells/state");
    DeviceGuard g(h->device);
    int rc = io_alloc(h);
    if (rc != SWARM_OK) return rc;
    const size_t EN = (size_t)h->cfg.n_env * h->cfg.n_agents;
    const void *act = action; int mode = 0;
    if (!action) {
        if (!h->cfg.llm_action) return fail(h, SWARM_ERR_INVALID, "swarm_step_host: null action");
        act = h->d_act_next; mode = 1;                                   // agent-major doubles
    } else if (action_on_device) {
        mode = action_dtype == SWARM_F64 ? 1 : 0;                        // [E][N][2] device tensor, as swarm_step
    } else {
        // the reference's (2, n_a) host array: through the pinned staging buffer, read component-major by the kernel
        const size_t bytes = EN * 2 * (action_dtype == SWARM_F64 ? 8 : 4);
        std::memcpy(h->h_io_action, action, bytes);
        HIP_TRY(h, hipMemcpyAsync(h->d_io_action, h->h_io_action, bytes, hipMemcpyHostToDevice, h->stream));
        act = h->d_io_action; mode = 2 | (action_dtype == SWARM_F64 ? 1 : 0);
    }
    rc = launch(h, true, act, mode, h->d_io_obs, h->d_io_rew, h->d_io_done, h->kp.with_prior ? h->d_io_prior : nullptr);
    if (rc != SWARM_OK) return rc;
    return io_export(h, slot, true);
}

int swarm_get_indices(swarm_env_t *h, int32_t *neighbor_index, int32_t *in_flags, int32_t *sensed_index, int32_t *occupied_index)
{
    if (!h) return SWARM_ERR_INVALID;
    if (!h->observed) return fail(h, SWARM_ERR_STATE, "swarm_get_indices: nothing observed yet");
    DeviceGuard g(h->device);
    const size_t EN = (size_t)h->cfg.n_env * h->cfg.n_agents;
    const bool lists = sensed_index || occupied_index;
    if (lists && !h->d_exp_sensed) {
        HIP_TRY(h, hipMalloc((void **)&h->d_exp_sensed, EN * (size_t)h->kp.g_max * 4));
        HIP_TRY(h, hipMalloc((void **)&h->d_exp_occ, EN * (size_t)h->kp.occ_max * 4));
    }
    {
        // re-run the observation pass on the current state with the export switched on (the step keeps the index
        // scratch in LDS and writes none of it to HBM); it recomputes the same caches from the same state, so it is
        // idempotent.
        h->kp.export_small = 1;
        if (lists) { h->kp.export_idx = 1; h->kp.exp_sensed = h->d_exp_sensed; h->kp.exp_occ = h->d_exp_occ; }
        int rc = launch(h, false, nullptr, 0, nullptr, nullptr, nullptr, nullptr);
        h->kp.export_idx = 0; h->kp.export_small = 0;
        if (rc != SWARM_OK) return rc;
        if (sensed_index) HIP_TRY(h, hipMemcpyAsync(sensed_index, h->d_exp_sensed, EN * (size_t)h->kp.g_max * 4, hipMemcpyDefault, h->stream));
        if (occupied_index) HIP_TRY(h, hipMemcpyAsync(occupied_index, h->d_exp_occ, EN * (size_t)h->kp.occ_max * 4, hipMemcpyDefault, h->stream));
    }
    if (neighbor_index) HIP_TRY(h, hipMemcpyAsync(neighbor_index, h->d_nei, EN * (size_t)h->kp.topo * 4, hipMemcpyDefault, h->stream));
    if (in_flags) HIP_TRY(h, hipMemcpyAsync(in_flags, h->d_inflag, EN * 4, hipMemcpyDefault, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SWARM_OK;
}

int swarm_rule_action(swarm_env_t *h, double *action)
{
    if (!h || !action) return SWARM_ERR_INVALID;
    if (!h->observed) return fail(h, SWARM_ERR_STATE, "swarm_rule_action: nothing observed yet");
    if (h->kp.g_max > 128) return fail(h, SWARM_ERR_INVALID, "swarm_rule_action: num_obs_grid_max > 128 not supported");
    DeviceGuard g(h->device);
    const size_t EN = (size_t)h->cfg.n_env * h->cfg.n_agents;
    if (!h->d_exp_sensed) {
        HIP_TRY(h, hipMalloc((void **)&h->d_exp_sensed, EN * (size_t)h->kp.g_max * 4));
        HIP_TRY(h, hipMalloc((void **)&h->d_exp_occ, EN * (size_t)h->kp.occ_max * 4));
    }
    // observation pass on the current state with the index export switched on (idempotent, see swarm_get_indices)
    h->kp.export_idx = 1; h->kp.export_small = 1; h->kp.exp_sensed = h->d_exp_sensed; h->kp.exp_occ = h->d_exp_occ;
    int rc = launch(h, false, nullptr, 0, nullptr, nullptr, nullptr, nullptr);
    h->kp.export_idx = 0; h->kp.export_small = 0;
    if (rc != SWARM_OK) return rc;
    hipLaunchKernelGGL(k_rule, dim3(h->cfg.n_env), dim3(h->cfg.n_agents <= 64 ? 64 : 256), 0, h->stream, h->kp, action);
    HIP_TRY(h, hipGetLastError());
    return SWARM_OK;
}

int swarm_lattice_envs(const swarm_env_t *h)
{
    if (!h) return -1;
    int n = 0;
    for (char c : h->lat_ok) n += c ? 1 : 0;
    return n;
}

double swarm_step_algorithmic_bytes(const swarm_env_t *h)
{
    if (!h) return 0.0;
    // Per agent-step: action 2*4 r, state p/dp 4*8 r + 4*8 w (fp64 here), obs D*sizeof w, reward 4 + done 1 +
    // prior 2*sizeof w; per env: target cells 2*n_g_max*8 r.  (with this build's dtypes.)
    const double so = h->cfg.obs_dtype == SWARM_F64 ? 8.0 : h->cfg.obs_dtype == SWARM_BF16 ? 2.0 : 4.0;
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


#ifdef SWARM_STAMPS
// Diagnostic build only: run one step with per-wave phase clocks; out[grid][waves per workgroup][24] (host), returns grid size.
int swarm_debug_stamps(swarm_env_t *h, const void *action, int action_dtype, void *obs, float *reward, uint8_t *done,
                       void *a_prior, long long *out, int max_blocks)
{
    if (!h || !out) return -1;
    DeviceGuard g(h->device);
    const int epb = h->npad < 64 ? (64 / h->npad) / (h->half ? 2 : 1) : 1;
    const int grid = (h->cfg.n_env + epb - 1) / epb;   // same for every Geo<NPAD>
    const int wpb = (h->npad < 64 ? 64 : h->npad) * 4 / 64;      // waves per workgroup
    if (grid > max_blocks) return -1;
    long long *d = nullptr;
    if (hipMalloc((void **)&d, (size_t)grid * wpb * 24 * sizeof(long long)) != hipSuccess) return -1;
    (void)hipMemset(d, 0, (size_t)grid * wpb * 24 * sizeof(long long));
    h->kp.stamps = d;
    int rc = launch(h, true, action, action_dtype == SWARM_F64, obs, reward, done, a_prior);
    h->kp.stamps = nullptr;
    if (rc == SWARM_OK && hipMemcpy(out, d, (size_t)grid * wpb * 24 * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) rc = -1;
    (void)hipFree(d);
    return rc == SWARM_OK ? grid : -1;
}
#endif

}  // extern "C"
