"""ctypes bindings for the parity oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  It binds two libraries:

* ``oracle/liboracle.so``  -- our plain-C restatement (``assembly_oracle.c``), class :class:`Oracle`.
* ``oracle/_ref/libAssemblyEnv.so`` -- the reference's own C++ compiled unmodified by
  ``oracle/Makefile`` (present only if it was built in the build container), class :class:`RefLib`,
  called exactly the way the reference's Python does it
  (/root/reference/cus_gym/gym/envs/customized_envs/assembly.py:234-255,357-380,460-466,495-504,613-624),
  plus :func:`ref_step`, a restatement of the numpy glue of ``AssemblySwarmEnv.step`` (assembly.py:487-666)
  around those five calls, used to time / check "reference C++ + glue" where the reference's
  Python files are absent (the GPU box).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "liboracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libAssemblyEnv.so")

# constants of the reference env (assembly.py:27-81,128-130,199)
TOPO = 6
G_MAX = 80
OCC_MAX = 200
SIZE_A = 0.035
K_BALL = 30.0
K_WALL = 100.0
C_WALL = 5.0
VEL_MAX = 0.8
DT = 0.1
D_SEN = 0.4
BOUNDARY = np.array([-2.4, 2.4, 2.4, -2.4], dtype=np.float64)


def build_oracle(force=False):
    """Compile oracle/liboracle.so (and oracle/_ref when the reference sources are present)."""
    src = os.path.join(_HERE, "assembly_oracle.c")
    if force or not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    if not os.path.exists(REF_SO):
        subprocess.call(["make", "-s", "-C", _HERE, "ref"])


def _d(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _i(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def _b(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte))


def obs_dim(with_self=True):
    return 2 * 2 * (TOPO + 1 + (1 if with_self else 0)) + 2 * G_MAX


def r_avoid_for(n_a, n_gs, l_cells):
    """assembly.py:124."""
    return round(float(np.sqrt(4 * np.min(n_gs) / (n_a * np.pi)) * np.min(l_cells)), 2)


class Oracle:
    """The C restatement.  All arrays float64 / int32 / uint8, C-contiguous, reference layouts."""

    def __init__(self):
        build_oracle()
        self.lib = ctypes.CDLL(ORACLE_SO)
        for name in ("orc_get_observation", "orc_get_reward", "orc_get_dist_b2b", "orc_sf_b2b_all",
                     "orc_get_dist_b2w", "orc_action_prior", "orc_step", "orc_step_batch"):
            getattr(self.lib, name).restype = None

    def get_observation(self, p, dp, grid, l_cell, r_avoid, d_sen=D_SEN, boundary=BOUNDARY,
                        is_periodic=False, with_self=True, topo=TOPO, g_max=G_MAX, occ_max=OCC_MAX):
        p = np.ascontiguousarray(p, np.float64); dp = np.ascontiguousarray(dp, np.float64)
        grid = np.ascontiguousarray(grid, np.float64)
        n_a, n_g = p.shape[1], grid.shape[1]
        od = 2 * 2 * (topo + 1 + (1 if with_self else 0)) + 2 * g_max
        obs = np.zeros((od, n_a)); nei = np.empty((n_a, topo), np.int32)
        inf = np.empty(n_a, np.int32); sen = np.empty((n_a, g_max), np.int32)
        occ = np.empty((n_a, occ_max), np.int32)
        cond = np.array([is_periodic, True, with_self, False], np.uint8)
        self.lib.orc_get_observation(_d(p), _d(dp), _d(obs), _d(np.ascontiguousarray(boundary, np.float64)), _d(grid),
                                     _i(nei), _i(inf), _i(sen), _i(occ),
                                     ctypes.c_double(d_sen), ctypes.c_double(r_avoid), ctypes.c_double(l_cell),
                                     ctypes.c_int(topo), ctypes.c_int(g_max), ctypes.c_int(occ_max),
                                     ctypes.c_int(n_a), ctypes.c_int(n_g), ctypes.c_int(od), _b(cond))
        return dict(obs=obs, neighbor_index=nei, in_flags=inf, sensed_index=sen, occupied_index=occ)

    def get_reward(self, p, grid, neighbor_index, in_flags, sensed_index, r_avoid, d_sen=D_SEN,
                   boundary=BOUNDARY, is_periodic=False):
        p = np.ascontiguousarray(p, np.float64); grid = np.ascontiguousarray(grid, np.float64)
        n_a, n_g = p.shape[1], grid.shape[1]
        rew = np.zeros((1, n_a))
        cond = np.array([is_periodic, True, True, True, True], np.uint8)
        self.lib.orc_get_reward(_d(p), _d(rew), _d(np.ascontiguousarray(boundary, np.float64)), _d(grid),
                                _i(np.ascontiguousarray(neighbor_index, np.int32)),
                                _i(np.ascontiguousarray(in_flags, np.int32)),
                                _i(np.ascontiguousarray(sensed_index, np.int32)),
                                ctypes.c_double(d_sen), ctypes.c_double(r_avoid),
                                ctypes.c_int(neighbor_index.shape[1]), ctypes.c_int(sensed_index.shape[1]),
                                ctypes.c_int(n_a), ctypes.c_int(n_g), _b(cond))
        return rew

    def dist_b2b(self, p, boundary=BOUNDARY, is_periodic=False, size_a=SIZE_A):
        p = np.ascontiguousarray(p, np.float64)
        n_a = p.shape[1]
        size = np.full(n_a, size_a)
        dc = np.empty((n_a, n_a)); de = np.empty((n_a, n_a)); col = np.empty((n_a, n_a), np.uint8)
        self.lib.orc_get_dist_b2b(_d(p), _d(size), _d(np.ascontiguousarray(boundary, np.float64)), ctypes.c_int(n_a),
                                  ctypes.c_int(int(is_periodic)), _d(dc), _d(de), _b(col))
        return dc, de, col.astype(bool)

    def sf_b2b_all(self, p, d_edge, collide, d_center, boundary=BOUNDARY, is_periodic=False, k_ball=K_BALL):
        p = np.ascontiguousarray(p, np.float64)
        n_a = p.shape[1]
        sf = np.zeros((2, n_a))
        self.lib.orc_sf_b2b_all(_d(p), _d(sf), _d(np.ascontiguousarray(d_edge)), _b(np.ascontiguousarray(collide, np.uint8)),
                                _d(np.ascontiguousarray(boundary, np.float64)), _d(np.ascontiguousarray(d_center)),
                                ctypes.c_int(n_a), ctypes.c_double(k_ball), ctypes.c_int(int(is_periodic)))
        return sf

    def dist_b2w(self, p, boundary=BOUNDARY, size_a=SIZE_A):
        p = np.ascontiguousarray(p, np.float64)
        n_a = p.shape[1]
        d = np.empty((4, n_a)); c = np.empty((4, n_a), np.uint8)
        self.lib.orc_get_dist_b2w(_d(p), _d(np.full(n_a, size_a)), _d(d), _b(c), ctypes.c_int(n_a),
                                  _d(np.ascontiguousarray(boundary, np.float64)))
        return d, c.astype(bool)

    def action_prior(self, p, dp, grid, neighbor_index, l_cell, r_avoid, d_sen=D_SEN):
        p = np.ascontiguousarray(p, np.float64); dp = np.ascontiguousarray(dp, np.float64)
        grid = np.ascontiguousarray(grid, np.float64)
        n_a, n_g = p.shape[1], grid.shape[1]
        ap = np.zeros((2, n_a))
        nei = np.ascontiguousarray(neighbor_index, np.int32)
        self.lib.orc_action_prior(_d(p), _d(dp), _d(ap), _d(grid), _i(nei), ctypes.c_double(d_sen),
                                  ctypes.c_double(r_avoid), ctypes.c_double(l_cell), ctypes.c_int(nei.shape[1]),
                                  ctypes.c_int(n_a), ctypes.c_int(n_g))
        return ap

    def step(self, p, dp, a, grid, neighbor_index, l_cell, r_avoid, d_sen=D_SEN, boundary=BOUNDARY,
             is_boundary=True, with_self=True, with_prior=True):
        """One env step.  Returns a dict with the NEW p, dp and every output (inputs are not modified)."""
        p = np.array(p, np.float64, order="C"); dp = np.array(dp, np.float64, order="C")
        a = np.ascontiguousarray(a, np.float64); grid = np.ascontiguousarray(grid, np.float64)
        n_a, n_g = p.shape[1], grid.shape[1]
        nei = np.array(neighbor_index, np.int32, order="C")
        od = obs_dim(with_self)
        obs = np.zeros((od, n_a)); rew = np.zeros((1, n_a)); ap = np.zeros((2, n_a))
        inf = np.empty(n_a, np.int32); sen = np.empty((n_a, G_MAX), np.int32); occ = np.empty((n_a, OCC_MAX), np.int32)
        co = np.array([not is_boundary, True, with_self, False], np.uint8)
        cr = np.array([not is_boundary, True, True, True, True], np.uint8)
        self.lib.orc_step(_d(p), _d(dp), _d(a), _d(obs), _d(rew), _d(ap) if with_prior else None,
                          _d(np.ascontiguousarray(boundary, np.float64)), _d(grid), _i(nei), _i(inf), _i(sen), _i(occ),
                          ctypes.c_double(d_sen), ctypes.c_double(r_avoid), ctypes.c_double(l_cell),
                          ctypes.c_double(SIZE_A), ctypes.c_double(K_BALL), ctypes.c_double(K_WALL),
                          ctypes.c_double(C_WALL), ctypes.c_double(VEL_MAX), ctypes.c_double(DT),
                          ctypes.c_int(TOPO), ctypes.c_int(G_MAX), ctypes.c_int(OCC_MAX),
                          ctypes.c_int(n_a), ctypes.c_int(n_g), ctypes.c_int(od), ctypes.c_int(int(is_boundary)),
                          _b(co), _b(cr))
        return dict(p=p, dp=dp, obs=obs, reward=rew, a_prior=ap if with_prior else None,
                    neighbor_index=nei, in_flags=inf, sensed_index=sen, occupied_index=occ,
                    done=np.zeros((1, n_a), bool))

    def step_batch(self, p, dp, a, grid, n_g, l_cell, neighbor_index, r_avoid, d_sen=D_SEN, boundary=BOUNDARY,
                   is_boundary=True, with_self=True):
        """E envs: p, dp, a [E,2,N]; grid [E,2,NG_MAX]; n_g, l_cell [E]; neighbor_index [E,N,6].
        Advances p, dp, neighbor_index IN PLACE (timing leg), returns (obs[E,od,N], reward[E,N], a_prior[E,2,N])."""
        E, _, n_a = p.shape
        od = obs_dim(with_self)
        obs = np.zeros((E, od, n_a)); rew = np.zeros((E, n_a)); ap = np.zeros((E, 2, n_a))
        inf = np.empty((E, n_a), np.int32); sen = np.empty((E, n_a, G_MAX), np.int32)
        occ = np.empty((E, n_a, OCC_MAX), np.int32)
        co = np.array([not is_boundary, True, with_self, False], np.uint8)
        cr = np.array([not is_boundary, True, True, True, True], np.uint8)
        n_g = np.ascontiguousarray(n_g, np.int32); l_cell = np.ascontiguousarray(l_cell, np.float64)
        self.lib.orc_step_batch(ctypes.c_int(E), _d(p), _d(dp), _d(a), _d(obs), _d(rew), _d(ap),
                                _d(np.ascontiguousarray(boundary, np.float64)), _d(grid), _i(n_g), _d(l_cell),
                                ctypes.c_int(grid.shape[2]), _i(neighbor_index), _i(inf), _i(sen), _i(occ),
                                ctypes.c_double(d_sen), ctypes.c_double(r_avoid), ctypes.c_double(SIZE_A),
                                ctypes.c_double(K_BALL), ctypes.c_double(K_WALL), ctypes.c_double(C_WALL),
                                ctypes.c_double(VEL_MAX), ctypes.c_double(DT), ctypes.c_int(TOPO), ctypes.c_int(G_MAX),
                                ctypes.c_int(OCC_MAX), ctypes.c_int(n_a), ctypes.c_int(od),
                                ctypes.c_int(int(is_boundary)), _b(co), _b(cr))
        return obs, rew, ap


class RefLib:
    """The reference's own libAssemblyEnv.so (compiled unmodified), called like assembly.py does."""

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def __init__(self):
        if not os.path.exists(REF_SO):
            build_oracle()
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO)
        self.lib = ctypes.CDLL(REF_SO)

    @staticmethod
    def _bb(a):
        return a.ctypes.data_as(ctypes.POINTER(ctypes.c_bool))

    def get_observation(self, p, dp, grid, l_cell, r_avoid, d_sen=D_SEN, boundary=BOUNDARY,
                        is_periodic=False, with_self=True, topo=TOPO, g_max=G_MAX, occ_max=OCC_MAX):
        p = np.ascontiguousarray(p, np.float64); dp = np.ascontiguousarray(dp, np.float64)
        grid = np.ascontiguousarray(grid, np.float64)
        n_a, n_g = p.shape[1], grid.shape[1]
        od = 2 * 2 * (topo + 1 + (1 if with_self else 0)) + 2 * g_max
        heading = np.zeros((2, n_a))
        obs = np.zeros((od, n_a)); nei = -np.ones((n_a, topo), np.int32)       # assembly.py:227-231
        inf = np.zeros(n_a, np.int32); sen = -np.ones((n_a, g_max), np.int32)
        occ = -np.ones((n_a, occ_max), np.int32)
        cond = np.array([is_periodic, True, with_self, False])
        self.lib._get_observation(_d(p), _d(dp), _d(heading), _d(obs), _d(np.ascontiguousarray(boundary, np.float64)),
                                  _d(grid), _i(nei), _i(inf), _i(sen), _i(occ),
                                  ctypes.c_double(d_sen), ctypes.c_double(r_avoid), ctypes.c_double(l_cell),
                                  ctypes.c_double(VEL_MAX), ctypes.c_int(topo), ctypes.c_int(g_max),
                                  ctypes.c_int(occ_max), ctypes.c_int(n_a), ctypes.c_int(n_g), ctypes.c_int(od),
                                  ctypes.c_int(2), self._bb(cond))
        return dict(obs=obs, neighbor_index=nei, in_flags=inf, sensed_index=sen, occupied_index=occ)

    def get_reward(self, p, grid, neighbor_index, in_flags, sensed_index, r_avoid, d_sen=D_SEN,
                   boundary=BOUNDARY, is_periodic=False, occupied_index=None):
        p = np.ascontiguousarray(p, np.float64); grid = np.ascontiguousarray(grid, np.float64)
        n_a, n_g = p.shape[1], grid.shape[1]
        nei = np.ascontiguousarray(neighbor_index, np.int32)
        sen = np.ascontiguousarray(sensed_index, np.int32)
        if occupied_index is None:
            occupied_index = -np.ones((n_a, OCC_MAX), np.int32)
        occ = np.ascontiguousarray(occupied_index, np.int32)
        rew = np.zeros((1, n_a)); zeros = np.zeros((2, n_a))
        coef = np.array([0.05]); cond = np.array([is_periodic, True, True, True, True], dtype=bool)
        cb2b = np.zeros((n_a, n_a), bool); cb2w = np.zeros((4, n_a), bool)
        self.lib._get_reward(_d(p), _d(zeros), _d(zeros.copy()), _d(zeros.copy()), _d(rew),
                             _d(np.ascontiguousarray(boundary, np.float64)), _d(grid), _i(nei),
                             _i(np.ascontiguousarray(in_flags, np.int32)), _i(sen), _i(occ),
                             ctypes.c_double(d_sen), ctypes.c_double(r_avoid), ctypes.c_double(0.0),
                             ctypes.c_int(nei.shape[1]), ctypes.c_int(sen.shape[1]), ctypes.c_int(occ.shape[1]),
                             ctypes.c_int(n_a), ctypes.c_int(n_g), ctypes.c_int(2), self._bb(cond),
                             self._bb(cb2b), self._bb(cb2w), _d(coef))
        return rew

    def sf_b2b_all(self, p, d_edge, collide, d_center, boundary=BOUNDARY, is_periodic=False, k_ball=K_BALL):
        p = np.ascontiguousarray(p, np.float64)
        n_a = p.shape[1]
        sf = np.zeros((2, n_a))
        self.lib._sf_b2b_all(_d(p), _d(sf), _d(np.ascontiguousarray(d_edge)), self._bb(np.ascontiguousarray(collide, bool)),
                             _d(np.ascontiguousarray(boundary, np.float64)), _d(np.ascontiguousarray(d_center)),
                             ctypes.c_int(n_a), ctypes.c_int(2), ctypes.c_double(k_ball), ctypes.c_bool(bool(is_periodic)))
        return sf

    def dist_b2w(self, p, boundary=BOUNDARY, size_a=SIZE_A):
        p = np.ascontiguousarray(p, np.float64)
        n_a = p.shape[1]
        d = np.ones((4, n_a)); c = np.zeros((4, n_a), bool)
        self.lib._get_dist_b2w(_d(p), _d(np.full(n_a, size_a)), _d(d), self._bb(c), ctypes.c_int(2), ctypes.c_int(n_a),
                               _d(np.ascontiguousarray(boundary, np.float64)))
        return d, c

    def action_prior(self, p, dp, grid, neighbor_index, l_cell, r_avoid, d_sen=D_SEN):
        p = np.ascontiguousarray(p, np.float64); dp = np.ascontiguousarray(dp, np.float64)
        grid = np.ascontiguousarray(grid, np.float64)
        n_a, n_g = p.shape[1], grid.shape[1]
        nei = np.ascontiguousarray(neighbor_index, np.int32)
        ap = np.zeros((2, n_a))
        self.lib.calculateActionPrior(_d(p), _d(dp), _d(ap), _d(grid), _i(nei), ctypes.c_double(d_sen),
                                      ctypes.c_double(r_avoid), ctypes.c_double(l_cell), ctypes.c_int(nei.shape[1]),
                                      ctypes.c_int(n_a), ctypes.c_int(n_g), ctypes.c_int(2))
        return ap


def numpy_dist_b2b(p, is_periodic=False, w_half=2.4, h_half=2.4, size_a=SIZE_A):
    """AssemblySwarmEnv._get_dist_b2b, assembly.py:442-457, including its rows-0/1-only periodic wrap."""
    n_a = p.shape[1]
    all_pos = np.tile(p, (n_a, 1))
    my_pos = np.tile(p.T.reshape(2 * n_a, 1), (1, n_a))
    rel = all_pos - my_pos
    if is_periodic:
        rel[0, rel[0, :] < -w_half] += 2 * w_half
        rel[0, rel[0, :] > w_half] -= 2 * w_half
        rel[1, rel[1, :] < -h_half] += 2 * h_half
        rel[1, rel[1, :] > h_half] -= 2 * h_half
    d_center = np.sqrt(rel[::2, :] ** 2 + rel[1::2, :] ** 2)
    size = np.full(n_a, size_a)
    sizes = np.tile(size.reshape(n_a, 1), (1, n_a))
    sizes = sizes + sizes.T
    sizes[np.arange(n_a), np.arange(n_a)] = 0
    d_edge = d_center - sizes
    collide = d_edge < 0
    return d_center, np.abs(d_edge), collide


def ref_step(ref, p, dp, a, grid, neighbor_index, l_cell, r_avoid, d_sen=D_SEN, boundary=BOUNDARY,
             is_boundary=True, with_self=True):
    """AssemblySwarmEnv.step (assembly.py:487-666; 'input' strategy, 'llm_rl' method) restated with the
    five native calls going to the REAL reference library `ref` (a RefLib).  Inputs are not modified."""
    p = np.array(p, np.float64, order="C"); dp = np.array(dp, np.float64, order="C")
    is_periodic = not is_boundary
    w_half = (boundary[2] - boundary[0]) / 2; h_half = (boundary[1] - boundary[3]) / 2
    d_center, d_edge, collide = numpy_dist_b2b(p, is_periodic, w_half, h_half)
    sf_b2b = ref.sf_b2b_all(p, d_edge, collide, d_center, boundary, is_periodic)
    if is_boundary:
        d_b2w, c_b2w = ref.dist_b2w(p, boundary)
        sf_b2w = np.array([[1, 0, -1, 0], [0, -1, 0, 1]]).dot(c_b2w * d_b2w) * 100
        df_b2w = np.array([[-1, 0, -1, 0], [0, -1, 0, -1]]).dot(c_b2w * np.concatenate((dp, dp), axis=0)) * 5
    a_prior = ref.action_prior(p, dp, grid, neighbor_index, l_cell, r_avoid, d_sen)
    F = 1 * a + sf_b2b + sf_b2w + df_b2w if is_boundary else 1 * a + sf_b2b
    ddp = F / np.ones(p.shape[1], dtype=int)
    dp += ddp * DT
    dp = np.clip(dp, -VEL_MAX, VEL_MAX)
    p += dp * DT
    if is_periodic:
        p[0, p[0, :] < boundary[0]] += 2 * w_half
        p[0, p[0, :] > boundary[2]] -= 2 * w_half
        p[1, p[1, :] < boundary[3]] += 2 * h_half
        p[1, p[1, :] > boundary[1]] -= 2 * h_half
    o = ref.get_observation(p, dp, grid, l_cell, r_avoid, d_sen, boundary, is_periodic, with_self)
    rew = ref.get_reward(p, grid, o["neighbor_index"], o["in_flags"], o["sensed_index"], r_avoid, d_sen,
                         boundary, is_periodic, o["occupied_index"])
    o.update(p=p, dp=dp, reward=rew, a_prior=a_prior, done=np.zeros((1, p.shape[1]), bool))
    return o


def wrapper_metrics(p, grid, r_avoid):
    """AssemblySwarmWrapper.coverage_rate / distribution_uniformity / voronoi_based_uniformity restated
    (/root/reference/cus_gym/gym/wrappers/customized_envs/assembly_wrapper.py:48-128): Python loops over cells and
    agents with numpy norms, np.var, np.argmin -- the same calls in the same order."""
    n_a, n_g = p.shape[1], grid.shape[1]
    occupied = 0
    for gi in range(n_g):                                              # :58-69
        if (np.linalg.norm(p - grid[:, [gi]], axis=0) < r_avoid / 2).any():
            occupied += 1
    m1 = occupied / n_g
    min_dist = []
    for i in range(n_a):                                               # :85-93
        d = np.linalg.norm(p - p[:, [i]], axis=0)
        min_dist.append(np.min(d[d != 0]))
    with np.errstate(all="ignore"):
        m2 = (np.var(min_dist) - np.min(min_dist)) / (np.max(min_dist) - np.min(min_dist))      # :96-97
        counts = np.zeros(n_a)
        for c in range(n_g):                                           # :110-121
            counts[np.argmin(np.linalg.norm(p - grid[:, [c]], axis=0))] += 1
        m3 = (np.var(counts) - np.min(counts)) / (np.max(counts) - np.min(counts))              # :124-126
    return np.array([m1, m2, m3])


def rule_action(p, dp, grid, l_cell, r_avoid, d_sen=D_SEN, g_max=G_MAX):
    """The rule-based expert controller, agent_strategy == 'rule' (assembly.py:530-601), restated with the same numpy
    calls in the same order.  Returns a (2, n_a) clipped to [-1, 1]."""
    n_a = p.shape[1]
    a = np.zeros((2, n_a))
    k_1, k_2, k_3 = 1, 15, 17                                                   # :532
    for i in range(n_a):
        rel_pos = grid - p[:, [i]]                                              # _get_trgt_grid_state :828-844
        rel_pos_norm = np.linalg.norm(rel_pos, axis=0)
        min_index = np.argmin(rel_pos_norm)
        if rel_pos_norm[min_index] < np.sqrt(2) * l_cell / 2:
            in_flag, target_pos, target_vel = 1, p[:, i], dp[:, i]
        else:
            in_flag, target_pos, target_vel = 0, grid[:, min_index], np.array([0, 0])
        sensed_indices = np.where(rel_pos_norm < d_sen)[0]
        target_pos_rel = target_pos - p[:, i]; target_vel_rel = target_vel - dp[:, i]
        if in_flag == 1:                                                        # :538-541
            v_ent = np.zeros(2)
        else:
            v_ent = k_1 * (target_pos_rel / (np.linalg.norm(target_pos_rel) + 1e-8)) + target_vel_rel
        if len(sensed_indices) > 0:                                             # :544-558
            sensed_grid = grid[:, sensed_indices]
            if in_flag == 1:
                agent_pos_rel_norm = np.linalg.norm(p - p[:, [i]], axis=0)
                for nb in np.where(agent_pos_rel_norm < (d_sen + r_avoid / 2))[0]:
                    mask = np.where(np.linalg.norm(sensed_grid - p[:, [nb]], axis=0) > r_avoid / 2)[0]
                    sensed_grid = sensed_grid[:, mask]; sensed_indices = sensed_indices[mask]
        n_s = len(sensed_indices)                                               # :561-572
        if n_s > g_max:
            step = (n_s - 1) / (g_max - 1)
            final = np.array(sensed_indices)[np.round(np.arange(0, g_max) * step).astype(int)]
            sensed_grid_pos = grid[:, final]
        elif n_s > 0:
            sensed_grid_pos = grid[:, sensed_indices]
        else:
            sensed_grid_pos = None
        v_exp = np.zeros(2)                                                     # :574-584
        if sensed_grid_pos is not None:
            rel = sensed_grid_pos - p[:, [i]]
            z = np.linalg.norm(rel, axis=0)
            psi = np.where(z < 0 * d_sen, 1.0, np.where(z < d_sen, 0.5 * (1.0 + np.cos(np.pi * (z / d_sen - 0) / (1.0 - 0))), 0.0))
            num = np.sum(psi * rel, axis=1); den = np.sum(psi)
            if den == 0:
                den = 1e-8
            v_exp += k_2 * num / den
        agent_pos_rel = p - p[:, [i]]; agent_vel_rel = dp - dp[:, [i]]          # :587-598
        nrm = np.linalg.norm(agent_pos_rel, axis=0)
        nearby = np.where(nrm < d_sen)[0]
        nearby = nearby[nearby != i]
        v_int = np.zeros(2)
        for nb in nearby:
            if nrm[nb] < r_avoid:
                v_int += -k_3 * (r_avoid / nrm[nb] - 1) * agent_pos_rel[:, nb]
            v_int += 5 * agent_vel_rel[:, nb] / len(nearby)
        a[:, i] = np.clip(v_ent + v_exp + v_int, -1, 1)                         # :600-601
    return a
