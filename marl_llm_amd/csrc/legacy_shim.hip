// legacy_shim.hip -- the reference's five extern "C" entry points, exact signatures and buffer contract
// (AssemblyEnv.h:13-34,35-58,64-73,75-81,98-109; called from assembly.py:234-255,357-380,460-466,495-504,
// 613-624), so an unmodified assembly.py can load libswarmenv.so where it loads libAssemblyEnv.so.
//
// Contract kept from the reference: host pointers, caller-owned pre-allocated buffers written in place,
// void returns, no state between calls.  Each call stages its inputs to the GPU, runs HIP kernels, copies the
// results back and synchronises.  _get_observation drives the same fused kernel as the batched ABI on a
// private one-environment handle; the other four are small one-thread-per-agent kernels (they take the
// reference's intermediate matrices / index lists as INPUTS, which the fused path never materialises).
// No CPU fallback: without a HIP device the call prints an error and fills its outputs with NaN / -1.
//
// Compile with -ffp-contract=off (see swarm_env.hip).

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <mutex>
#include <vector>

#include "swarm_env.h"

namespace {

std::mutex g_mu;

// The reference's five entry points return void (AssemblyEnv.h:13-109), so a failure cannot be reported through them:
// outputs are poisoned with NaN, a line goes to stderr, and the status / message of the last legacy call of this thread
// can be read through swarm_legacy_status() / swarm_legacy_last_error().
thread_local int g_legacy_status = 0;
thread_local char g_legacy_msg[256] = "";
void legacy_fail(const char *fn, const char *why)
{
    g_legacy_status = 1;
    std::snprintf(g_legacy_msg, sizeof(g_legacy_msg), "%s: %s", fn, why);
    std::fprintf(stderr, "libswarmenv: %s: %s\n", fn, why);
}

struct Arena {          // grow-only device scratch, reset per call
    char *base = nullptr;
    size_t cap = 0, used = 0;
    bool reserve(size_t bytes)
    {
        if (bytes <= cap) { used = 0; return true; }
        if (base) (void)hipFree(base);
        base = nullptr; cap = 0; used = 0;
        if (hipMalloc((void **)&base, bytes) != hipSuccess) return false;
        cap = bytes;
        return true;
    }
    template <typename T> T *take(size_t n)
    {
        used = (used + 255) & ~size_t(255);
        T *p = reinterpret_cast<T *>(base + used);
        used += n * sizeof(T);
        return p;
    }
} g_arena;

bool device_ok(const char *fn)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
        std::fprintf(stderr, "libswarmenv: %s: no HIP device available; this library has no CPU path\n", fn);
        return false;
    }
    return true;
}

bool ok(hipError_t e, const char *fn, const char *what)
{
    if (e == hipSuccess) return true;
    std::fprintf(stderr, "libswarmenv: %s: %s failed: %s\n", fn, what, hipGetErrorString(e));
    return false;
}

__device__ __forceinline__ void wrap_rel(double &x, double &y, double wh, double hh)
{   // AssemblyEnv.cpp:700-715
    if (x < -wh) x += 2 * wh; else if (x > wh) x -= 2 * wh;
    if (y < -hh) y += 2 * hh; else if (y > hh) y -= 2 * hh;
}

// _get_reward, AssemblyEnv.cpp:452-559, one thread per agent.
__global__ void k_legacy_reward(const double *p, double *reward, const double *grid, const int *nei, const int *in_flags,
                                const int *sensed, double d_sen, double r_avoid, int topo, int gmax, int n_a, int n_g,
                                int periodic, int pen_inter, int pen_explore, double wh, double hh)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_a) return;
    const double px = p[i], py = p[n_a + i];
    bool collision = false, uniform = false;
    if (pen_inter) {
        for (int k = 0; k < topo; ++k) {
            const int j = nei[i * topo + k];
            if (j == -1) continue;
            double x = p[j] - px, y = p[n_a + j] - py;
            if (periodic) wrap_rel(x, y, wh, hh);
            if (r_avoid > sqrt(x * x + y * y)) { collision = true; break; }
        }
    }
    double r = 0.0;
    if (pen_explore) {
        if (in_flags[i] == 1) {
            double num0 = 0.0, num1 = 0.0, den = 0.0; bool any = false;
            for (int s = 0; s < gmax; ++s) {
                const int c = sensed[(size_t)i * gmax + s];
                if (c == -1) continue;
                any = true;
                const double x = grid[c] - px, y = grid[n_g + c] - py;
                const double z = sqrt(x * x + y * y);
                double psi;
                if (z < 0.0 * d_sen) psi = 1.0;
                else if (z < d_sen) psi = (1.0 / 2.0) * (1.0 + cos(M_PI * (z / d_sen - 0.0) / (1.0 - 0.0)));
                else psi = 0.0;
                num0 += psi * x; num1 += psi * y; den += psi;
            }
            if (any) {
                if (den == 0) den = 1E-8;
                const double v0 = 1.0 * num0 / den, v1 = 1.0 * num1 / den;
                if (sqrt(v0 * v0 + v1 * v1) < 0.05) uniform = true;
            }
        }
        if (in_flags[i] == 1 && !collision && uniform) r += 1.0;
    }
    reward[i] = r;
}

// _sf_b2b_all, AssemblyEnv.cpp:735-815, one thread per agent i: sum over k (index order) of the
// antisymmetric pair table built from the caller's d_edge / d_center / collide matrices.
__global__ void k_legacy_sf_b2b(const double *p, double *sf, const double *d_edge, const unsigned char *collide,
                                const double *d_center, int n_a, double k_ball, int periodic, double wh, double hh)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_a) return;
    double sx = 0.0, sy = 0.0;
    for (int k = 0; k < n_a; ++k) {
        if (k == i) continue;
        const int a = i > k ? i : k, b = i > k ? k : i;       // the reference evaluates the pair as (a > b)
        double x = p[b] - p[a], y = p[n_a + b] - p[n_a + a];
        if (periodic) wrap_rel(x, y, wh, hh);
        const double ux = x / d_center[(size_t)a * n_a + b], uy = y / d_center[(size_t)a * n_a + b];
        const double c = (double)collide[(size_t)a * n_a + b];
        double fx = c * d_edge[(size_t)a * n_a + b] * k_ball * (-ux);
        double fy = c * d_edge[(size_t)a * n_a + b] * k_ball * (-uy);
        if (i < k) { fx = -fx; fy = -fy; }
        sx += fx; sy += fy;
    }
    sf[i] = sx; sf[n_a + i] = sy;
}

// _get_dist_b2w, AssemblyEnv.cpp:817-855.
__global__ void k_legacy_b2w(const double *p, const double *r, double *d_b2w, unsigned char *collide, int n_a,
                             double bx0, double by1, double bx2, double by3)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_a) return;
    double d[4];
    d[0] = p[i] - r[i] - bx0;
    d[1] = by1 - (p[n_a + i] + r[i]);
    d[2] = bx2 - (p[i] + r[i]);
    d[3] = p[n_a + i] - r[i] - by3;
    for (int w = 0; w < 4; ++w) {
        collide[w * n_a + i] = d[w] < 0;
        d_b2w[w * n_a + i] = fabs(d[w]);
    }
}

// calculateActionPrior + robotPolicy + _get_target_grid_state, AssemblyEnv.cpp:1061-1196,858-908.
__global__ void k_legacy_prior(const double *p, const double *dp, double *a_prior, const double *grid, const int *nei,
                               double r_avoid, double in_thr, int topo, int n_a, int n_g)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_a) return;
    const double px = p[i], py = p[n_a + i];
    double best = 0.0; int bc = 0;
    for (int c = 0; c < n_g; ++c) {
        const double x = grid[c] - px, y = grid[n_g + c] - py;
        const double d = sqrt(x * x + y * y);
        if (c == 0 || d < best) { best = d; bc = c; }
    }
    double tx, ty;
    if (n_g > 0 && best < in_thr) { tx = px - px; ty = py - py; }
    else { tx = grid[bc] - px; ty = grid[n_g + bc] - py; }
    double fx = 0.0, fy = 0.0;
    const double dt = sqrt(tx * tx + ty * ty);
    if (dt > 0) { fx += 2.0 * tx / dt; fy += 2.0 * ty / dt; }
    double avx = 0.0, avy = 0.0; int cnt = 0;
    for (int k = 0; k < topo; ++k) {
        const int j = nei[i * topo + k];
        if (j == -1) continue;
        const double x = px - p[j], y = py - p[n_a + j];
        const double d = sqrt(x * x + y * y);
        if (d > 0 && d < r_avoid) {
            const double ux = x / d, uy = y / d;
            const double factor = 3.0 * (r_avoid / d - 1.0);
            fx += factor * ux; fy += factor * uy;
        }
        avx += dp[j]; avy += dp[n_a + j]; ++cnt;
    }
    if (cnt > 0) {
        avx /= cnt; avy /= cnt;
        fx += 2.0 * (avx - dp[i]); fy += 2.0 * (avy - dp[n_a + i]);
    }
    double m = (1.0 < fx) ? 1.0 : fx; a_prior[i] = (-1.0 < m) ? m : -1.0;
    m = (1.0 < fy) ? 1.0 : fy; a_prior[n_a + i] = (-1.0 < m) ? m : -1.0;
}

// one cached private handle for _get_observation
struct ObsCtx {
    swarm_env_t *h = nullptr;
    swarm_config_t cfg;
    double *d_obs = nullptr;
    int32_t *d_nei = nullptr, *d_inf = nullptr, *d_sen = nullptr, *d_occ = nullptr;
    void drop()
    {
        if (h) swarm_destroy(h);
        h = nullptr;
        (void)hipFree(d_obs); (void)hipFree(d_nei); (void)hipFree(d_inf); (void)hipFree(d_sen); (void)hipFree(d_occ);
        d_obs = nullptr; d_nei = d_inf = d_sen = d_occ = nullptr;
    }
} g_obs;

void fill_nan(double *a, size_t n) { for (size_t k = 0; k < n; ++k) a[k] = std::numeric_limits<double>::quiet_NaN(); }

}  // namespace

extern "C" {

void _get_observation(double *p_input, double *dp_input, double *heading_input, double *obs_input,
                      double *boundary_pos_input, double *grid_center_input, int *neighbor_index_input,
                      int *in_flags_input, int *sensed_index_input, int *occupied_index_input, double d_sen,
                      double r_avoid, double l_cell, double Vel_max, int topo_nei_max, int num_obs_grid_max,
                      int num_occupied_grid_max, int n_a, int n_g, int obs_dim_agent, int dim, bool *condition)
{
    (void)heading_input; (void)Vel_max;
    std::lock_guard<std::mutex> lk(g_mu);
    g_legacy_status = 0; g_legacy_msg[0] = 0;
    const char *fn = "_get_observation";
    const size_t nobs = (size_t)obs_dim_agent * n_a;
    auto bail = [&](const char *why) {
        legacy_fail(fn, why);
        fill_nan(obs_input, nobs);
    };
    if (dim != 2) return bail("only dim == 2 is supported");
    if (!condition[1]) return bail("only Cartesian dynamics are supported (the reference's Python refuses the rest too)");
    if (!device_ok(fn)) return bail("no device");
    swarm_config_t c;
    swarm_default_config(&c);
    c.n_env = 1; c.n_agents = n_a; c.n_cells_max = n_g;
    c.topo_nei_max = topo_nei_max; c.num_obs_grid_max = num_obs_grid_max; c.num_occupied_grid_max = num_occupied_grid_max;
    c.is_boundary = condition[0] ? 0 : 1; c.with_self_state = condition[2] ? 1 : 0; c.with_prior = 0;
    c.obs_dtype = SWARM_F64; c.device = -1; c.d_sen = d_sen; c.r_avoid = r_avoid;
    for (int k = 0; k < 4; ++k) c.boundary[k] = boundary_pos_input[k];
    if (!g_obs.h || std::memcmp(&c, &g_obs.cfg, sizeof(c)) != 0) {
        g_obs.drop();
        if (swarm_create(&c, &g_obs.h) != SWARM_OK) { g_obs.h = nullptr; return bail(swarm_last_error(nullptr)); }
        g_obs.cfg = c;
        const int D = swarm_obs_dim(g_obs.h);
        if (D != obs_dim_agent) { g_obs.drop(); return bail("obs_dim_agent does not match 4*(topo+1+self)+2*num_obs_grid_max"); }
        bool a = ok(hipMalloc((void **)&g_obs.d_obs, nobs * 8), fn, "hipMalloc") &&
                 ok(hipMalloc((void **)&g_obs.d_nei, (size_t)n_a * topo_nei_max * 4), fn, "hipMalloc") &&
                 ok(hipMalloc((void **)&g_obs.d_inf, (size_t)n_a * 4), fn, "hipMalloc") &&
                 ok(hipMalloc((void **)&g_obs.d_sen, (size_t)n_a * num_obs_grid_max * 4), fn, "hipMalloc") &&
                 ok(hipMalloc((void **)&g_obs.d_occ, (size_t)n_a * num_occupied_grid_max * 4), fn, "hipMalloc");
        if (!a) { g_obs.drop(); return bail("device allocation failed"); }
    }
    swarm_env_t *h = g_obs.h;
    int32_t ng = n_g;
    if (swarm_set_cells(h, 0, 1, grid_center_input, &ng, &l_cell) != SWARM_OK) return bail(swarm_last_error(h));
    if (swarm_set_state(h, p_input, dp_input) != SWARM_OK) return bail(swarm_last_error(h));
    if (swarm_observe(h, g_obs.d_obs) != SWARM_OK) return bail(swarm_last_error(h));
    if (swarm_get_indices(h, g_obs.d_nei, g_obs.d_inf, g_obs.d_sen, g_obs.d_occ) != SWARM_OK) return bail(swarm_last_error(h));
    std::vector<double> rows(nobs);
    bool a = ok(hipMemcpy(rows.data(), g_obs.d_obs, nobs * 8, hipMemcpyDeviceToHost), fn, "hipMemcpy") &&
             ok(hipMemcpy(neighbor_index_input, g_obs.d_nei, (size_t)n_a * topo_nei_max * 4, hipMemcpyDeviceToHost), fn, "hipMemcpy") &&
             ok(hipMemcpy(in_flags_input, g_obs.d_inf, (size_t)n_a * 4, hipMemcpyDeviceToHost), fn, "hipMemcpy") &&
             ok(hipMemcpy(sensed_index_input, g_obs.d_sen, (size_t)n_a * num_obs_grid_max * 4, hipMemcpyDeviceToHost), fn, "hipMemcpy") &&
             ok(hipMemcpy(occupied_index_input, g_obs.d_occ, (size_t)n_a * num_occupied_grid_max * 4, hipMemcpyDeviceToHost), fn, "hipMemcpy");
    if (!a) return bail("copy back failed");
    for (int i = 0; i < n_a; ++i)                 // rows [N][D] -> the reference's (D, N), AssemblyEnv.cpp:324-328
        for (int r = 0; r < obs_dim_agent; ++r) obs_input[(size_t)r * n_a + i] = rows[(size_t)i * obs_dim_agent + r];
}

void _get_reward(double *p_input, double *dp_input, double *heading_input, double *act_input, double *reward_input,
                 double *boundary_pos_input, double *grid_center_input, int *neighbor_index_input, int *in_flags_input,
                 int *sensed_index_input, int *occupied_index_input, double d_sen, double r_avoid, double l_cell,
                 int topo_nei_max, int num_obs_grid_max, int num_occupied_grid_max, int n_a, int n_g, int dim,
                 bool *condition, bool *is_collide_b2b_input, bool *is_collide_b2w_input, double *coefficients)
{
    (void)dp_input; (void)heading_input; (void)act_input; (void)occupied_index_input; (void)l_cell;
    (void)num_occupied_grid_max; (void)is_collide_b2b_input; (void)is_collide_b2w_input; (void)coefficients;
    std::lock_guard<std::mutex> lk(g_mu);
    g_legacy_status = 0; g_legacy_msg[0] = 0;
    const char *fn = "_get_reward";
    auto bail = [&](const char *why) { legacy_fail(fn, why); fill_nan(reward_input, (size_t)n_a); };
    if (dim != 2) return bail("only dim == 2 is supported");
    if (!device_ok(fn)) return bail("no device");
    const size_t N = (size_t)n_a;
    if (!g_arena.reserve(N * 2 * 8 + N * 8 + (size_t)n_g * 2 * 8 + N * topo_nei_max * 4 + N * 4 + N * num_obs_grid_max * 4 + 4096))
        return bail("device allocation failed");
    double *d_p = g_arena.take<double>(2 * N), *d_r = g_arena.take<double>(N), *d_g = g_arena.take<double>(2 * (size_t)n_g);
    int *d_nei = g_arena.take<int>(N * topo_nei_max), *d_inf = g_arena.take<int>(N), *dv_sen = g_arena.take<int>(N * num_obs_grid_max);
    const double wh = (boundary_pos_input[2] - boundary_pos_input[0]) / 2.0, hh = (boundary_pos_input[1] - boundary_pos_input[3]) / 2.0;
    bool a = ok(hipMemcpy(d_p, p_input, 2 * N * 8, hipMemcpyHostToDevice), fn, "hipMemcpy") &&
             ok(hipMemcpy(d_g, grid_center_input, 2 * (size_t)n_g * 8, hipMemcpyHostToDevice), fn, "hipMemcpy") &&
             ok(hipMemcpy(d_nei, neighbor_index_input, N * topo_nei_max * 4, hipMemcpyHostToDevice), fn, "hipMemcpy") &&
             ok(hipMemcpy(d_inf, in_flags_input, N * 4, hipMemcpyHostToDevice), fn, "hipMemcpy") &&
             ok(hipMemcpy(dv_sen, sensed_index_input, N * num_obs_grid_max * 4, hipMemcpyHostToDevice), fn, "hipMemcpy");
    if (!a) return bail("copy in failed");
    hipLaunchKernelGGL(k_legacy_reward, dim3((n_a + 63) / 64), dim3(64), 0, 0, d_p, d_r, d_g, d_nei, d_inf, dv_sen, d_sen,
                       r_avoid, topo_nei_max, num_obs_grid_max, n_a, n_g, (int)condition[0], (int)condition[3],
                       (int)condition[4], wh, hh);
    if (!ok(hipGetLastError(), fn, "launch") || !ok(hipMemcpy(reward_input, d_r, N * 8, hipMemcpyDeviceToHost), fn, "hipMemcpy"))
        return bail("kernel failed");
}

void _sf_b2b_all(double *p_input, double *sf_b2b_input, double *d_b2b_edge_input, bool *is_collide_b2b_input,
                 double *boundary_pos_input, double *d_b2b_center_input, int n_a, int dim, double k_ball, bool is_periodic)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_legacy_status = 0; g_legacy_msg[0] = 0;
    const char *fn = "_sf_b2b_all";
    auto bail = [&](const char *why) { legacy_fail(fn, why); fill_nan(sf_b2b_input, (size_t)2 * n_a); };
    if (dim != 2) return bail("only dim == 2 is supported");
    if (!device_ok(fn)) return bail("no device");
    const size_t N = (size_t)n_a;
    if (!g_arena.reserve(4 * N * 8 + 2 * N * N * 8 + N * N + 4096)) return bail("device allocation failed");
    double *d_p = g_arena.take<double>(2 * N), *d_sf = g_arena.take<double>(2 * N);
    double *d_de = g_arena.take<double>(N * N), *d_dc = g_arena.take<double>(N * N);
    unsigned char *d_c = g_arena.take<unsigned char>(N * N);
    const double wh = (boundary_pos_input[2] - boundary_pos_input[0]) / 2.0, hh = (boundary_pos_input[1] - boundary_pos_input[3]) / 2.0;
    bool a = ok(hipMemcpy(d_p, p_input, 2 * N * 8, hipMemcpyHostToDevice), fn, "hipMemcpy") &&
             ok(hipMemcpy(d_de, d_b2b_edge_input, N * N * 8, hipMemcpyHostToDevice), fn, "hipMemcpy") &&
             ok(hipMemcpy(d_dc, d_b2b_center_input, N * N * 8, hipMemcpyHostToDevice), fn, "hipMemcpy") &&
             ok(hipMemcpy(d_c, is_collide_b2b_input, N * N, hipMemcpyHostToDevice), fn, "hipMemcpy");
    if (!a) return bail("copy in failed");
    hipLaunchKernelGGL(k_legacy_sf_b2b, dim3((n_a + 63) / 64), dim3(64), 0, 0, d_p, d_sf, d_de, d_c, d_dc, n_a, k_ball,
                       (int)is_periodic, wh, hh);
    if (!ok(hipGetLastError(), fn, "launch") || !ok(hipMemcpy(sf_b2b_input, d_sf, 2 * N * 8, hipMemcpyDeviceToHost), fn, "hipMemcpy"))
        return bail("kernel failed");
}

void _get_dist_b2w(double *p_input, double *r_input, double *d_b2w_input, bool *isCollision_input, int dim, int n_a,
                   double *boundary_pos)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_legacy_status = 0; g_legacy_msg[0] = 0;
    const char *fn = "_get_dist_b2w";
    auto bail = [&](const char *why) { legacy_fail(fn, why); fill_nan(d_b2w_input, (size_t)4 * n_a); };
    if (dim != 2) return bail("only dim == 2 is supported");
    if (!device_ok(fn)) return bail("no device");
    const size_t N = (size_t)n_a;
    if (!g_arena.reserve(8 * N * 8 + 4 * N + 4096)) return bail("device allocation failed");
    double *d_p = g_arena.take<double>(2 * N), *d_r = g_arena.take<double>(N), *d_d = g_arena.take<double>(4 * N);
    unsigned char *d_c = g_arena.take<unsigned char>(4 * N);
    bool a = ok(hipMemcpy(d_p, p_input, 2 * N * 8, hipMemcpyHostToDevice), fn, "hipMemcpy") &&
             ok(hipMemcpy(d_r, r_input, N * 8, hipMemcpyHostToDevice), fn, "hipMemcpy");
    if (!a) return bail("copy in failed");
    hipLaunchKernelGGL(k_legacy_b2w, dim3((n_a + 63) / 64), dim3(64), 0, 0, d_p, d_r, d_d, d_c, n_a, boundary_pos[0],
                       boundary_pos[1], boundary_pos[2], boundary_pos[3]);
    if (!ok(hipGetLastError(), fn, "launch") || !ok(hipMemcpy(d_b2w_input, d_d, 4 * N * 8, hipMemcpyDeviceToHost), fn, "hipMemcpy") ||
        !ok(hipMemcpy(isCollision_input, d_c, 4 * N, hipMemcpyDeviceToHost), fn, "hipMemcpy"))
        return bail("kernel failed");
}

void calculateActionPrior(double *p_input, double *dp_input, double *a_prior_input, double *grid_center_input,
                          int *neighbor_index_input, double d_sen, double r_avoid, double l_cell, int topo_nei_max,
                          int n_a, int n_g, int dim)
{
    (void)d_sen;
    std::lock_guard<std::mutex> lk(g_mu);
    g_legacy_status = 0; g_legacy_msg[0] = 0;
    const char *fn = "calculateActionPrior";
    auto bail = [&](const char *why) { legacy_fail(fn, why); fill_nan(a_prior_input, (size_t)2 * n_a); };
    if (dim != 2) return bail("only dim == 2 is supported");
    if (!device_ok(fn)) return bail("no device");
    const size_t N = (size_t)n_a;
    if (!g_arena.reserve(6 * N * 8 + 2 * (size_t)n_g * 8 + N * topo_nei_max * 4 + 4096)) return bail("device allocation failed");
    double *d_p = g_arena.take<double>(2 * N), *d_dp = g_arena.take<double>(2 * N), *d_a = g_arena.take<double>(2 * N);
    double *d_g = g_arena.take<double>(2 * (size_t)n_g);
    int *d_nei = g_arena.take<int>(N * topo_nei_max);
    bool a = ok(hipMemcpy(d_p, p_input, 2 * N * 8, hipMemcpyHostToDevice), fn, "hipMemcpy") &&
             ok(hipMemcpy(d_dp, dp_input, 2 * N * 8, hipMemcpyHostToDevice), fn, "hipMemcpy") &&
             ok(hipMemcpy(d_g, grid_center_input, 2 * (size_t)n_g * 8, hipMemcpyHostToDevice), fn, "hipMemcpy") &&
             ok(hipMemcpy(d_nei, neighbor_index_input, N * topo_nei_max * 4, hipMemcpyHostToDevice), fn, "hipMemcpy");
    if (!a) return bail("copy in failed");
    hipLaunchKernelGGL(k_legacy_prior, dim3((n_a + 63) / 64), dim3(64), 0, 0, d_p, d_dp, d_a, d_g, d_nei, r_avoid,
                       std::sqrt(2) * l_cell / 2, topo_nei_max, n_a, n_g);
    if (!ok(hipGetLastError(), fn, "launch") || !ok(hipMemcpy(a_prior_input, d_a, 2 * N * 8, hipMemcpyDeviceToHost), fn, "hipMemcpy"))
        return bail("kernel failed");
}

int swarm_legacy_status(void) { return g_legacy_status; }
const char *swarm_legacy_last_error(void) { return g_legacy_msg; }

}  // extern "C"
