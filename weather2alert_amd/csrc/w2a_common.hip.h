// w2a_common.hip.h -- geometry, packed per-env state, device RNG, DPP reductions, episode draw, observation tile
// Part of libw2a.so; included only by w2a_kernels.hip (one translation unit, see the file comment there).
#ifndef W2A_COMMON_HIP_H
#define W2A_COMMON_HIP_H

#include "w2a_bookkeeping.h"

#ifndef LANES
#define LANES 4  // lanes per env: 8, 4, 2 or 1 (A/B-tested on MI355X)
#endif
#define ROWF 32
#define QUADS (ROWF / 4 / LANES)  // float4 per lane per 32-float row
#ifndef BLOCK
#define BLOCK 256  // threads per workgroup (measured at 1 M envs: 64..256 within 2 %, 512 is 4 % slower)
#endif
#define ENVS_PER_BLOCK (BLOCK / LANES)
#define ENVS_PER_WAVE (64 / LANES)
#define HDR_BYTES 256
#ifndef W2A_NT_OBS
#define W2A_NT_OBS 1  // observation rows leave with non-temporal stores (they are not re-read by the env)
#endif
#ifndef W2A_NT_W
#define W2A_NT_W 0      // A/B: non-temporal loads for the gathered coefficient rows
#endif
#ifndef W2A_XCD_SWIZZLE
#define W2A_XCD_SWIZZLE 1  // consecutive env tiles on the same XCD (workgroups are dealt round-robin over 8 XCDs)
#endif
// Logical tile of a workgroup. With the swizzle, XCD k (blockIdx % 8 == k, observed placement; only speed
// depends on it) walks the k-th contiguous eighth of the env range, so neighbouring envs share an L2: partial
// output lines (reward, done) merge there, and with episode_order="sorted" each XCD touches one eighth of W / L.
__device__ __forceinline__ uint32_t logical_block(uint32_t b, uint32_t per_xcd) {
#if W2A_XCD_SWIZZLE
  return (b & 7u) * per_xcd + (b >> 3);
#else
  return b;
#endif
}
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_w(const float4 *p) {
#if W2A_NT_W
  v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}
#define RT_QUAD 6    // float4 index of the run-time slots 24..27
#define GATE_QUAD 7  // float4 index holding the gate copy (slot 30 = .z)

// ----------------------------------------------------------------------------------------
// packed state
// ----------------------------------------------------------------------------------------
// Three arrays (40 B per env; a step streams 24 B in and 12 B out):
//   cold  (uint4, reset-time record, not read by the plain step kernel):
//           x = ep_row (county_w*Y + year_i), y = coef_col << 12 | sample (n_samples <= 4096),
//           z = sticky budget (int, -1 unset), w = episode number
//   stepc (3 x u32, read-only while an episode runs): budget (int), ep_row, ep_w  (copies of cold.x/.y)
//   hot3  (3 x u32, read and written every step):
//           dyn0: t[0:10) used[10:20) streak[20:30) last_actual[30] at_budget[31]
//           dyn1: hist14[0:14) n_days[16:26) finished[31]
//           episode return (f32 bits)
// Kernels work on the logical views  cold = {ep_row, ep_w, sticky, episode_no},  hot = {dyn0, dyn1, ret, budget}.
struct u3 { uint32_t a, b, c; };
struct StateArrays {
  uint4 *cold;
  u3 *hot3;
  u3 *stepc;
  // lock-step mirror of hot3 / stepc, 8 + 8 B per env (k_step64's packed variant streams 20 B in and 8 B out per
  // env-step instead of 28 and 12). In lock step the day t and the episode length are the same for every env: the
  // length (a table constant) travels as a kernel argument, the day lives in ONE word per 64-env tile (pk_day: read
  // and advanced by the wave that owns the tile, 4 B per 64 envs) -- in device memory, so that a step recorded into a
  // hipGraph finds the right day on every replay:
  //   pk_hot.x: used[0:8) streak[8:16) hist14[16:30) finished[30]      pk_hot.y: episode return (f32 bits)
  //   pk_c.x:   budget16[0:16) coef_col[16:32)                          pk_c.y:   ep_row[0:22) sample[22:32)
  //   pk_day[tile]: the day of envs 64 tile .. 64 tile + 63, or W2A_PK_DAY_POISON (w2a_bookkeeping.h, graph_packed)
  // (last_actual = hist14 & 1; at_budget is derived). Valid for T <= 255, S < 65536, n_samples <= 1024,
  // S_w * Y < 2^22 -- table properties, checked once by w2a_create; the host tracks which of the two forms is current.
  // Budgets of ANY size are served: budget16 holds budgets 0 .. 65534 themselves and PK_BUDGET_ESCAPE for everything
  // else (65535 and above; negative values a caller handed over), and the packed kernel then reads the env's canonical
  // stepc word -- written by every reset path (store_episode), read-only while an episode runs, so never stale. Until
  // round 5 the host kept an upper bound of every budget in the buffer instead; four of its eight bookkeeping holes were
  // in that bound.
  uint2 *pk_hot;
  uint2 *pk_c;
  uint32_t *pk_day;
};
#define W2A_PK_DAY_POISON 0xFFFFFFFFu
#define PK_BUDGET_ESCAPE 0xFFFFu
__device__ __forceinline__ uint32_t pk_budget16(uint32_t budget) { return budget < PK_BUDGET_ESCAPE ? budget : PK_BUDGET_ESCAPE; }
#define PK_USED(w) ((w) & 255u)
#define PK_STREAK(w) (((w) >> 8) & 255u)
#define PK_HIST(w) (((w) >> 16) & 0x3FFFu)
#define PK_FIN(w) (((w) >> 30) & 1u)
__device__ __forceinline__ uint32_t pk_pack_hot(uint32_t used, uint32_t streak, uint32_t hist, uint32_t fin) {
  return (used & 255u) | ((streak & 255u) << 8) | ((hist & 0x3FFFu) << 16) | (fin << 30);
}
// per-step view: hot complete, cold.x/.y valid (cold.z/.w = 0: load_cold() when the sticky budget / episode
// number are needed)
__device__ __forceinline__ void load_step_state(const StateArrays &s, uint32_t e, uint4 &cold, uint4 &hot) {
  const u3 h = s.hot3[e];
  const u3 c = s.stepc[e];
  hot = make_uint4(h.a, h.b, h.c, c.a);
  cold = make_uint4(c.b, c.c, 0u, 0u);
}
__device__ __forceinline__ uint4 load_cold(const StateArrays &s, uint32_t e) { return s.cold[e]; }
__device__ __forceinline__ void store_hot(const StateArrays &s, uint32_t e, const uint4 hot) {
  u3 h; h.a = hot.x; h.b = hot.y; h.c = hot.z;
  s.hot3[e] = h;
}
// a new episode: all arrays
__device__ __forceinline__ void store_episode(const StateArrays &s, uint32_t e, const uint4 cold, const uint4 hot) {
  s.cold[e] = cold;
  u3 c; c.a = hot.w; c.b = cold.x; c.c = cold.y;
  s.stepc[e] = c;
  store_hot(s, e, hot);
}
#define D0_T(d) ((d) & 1023u)
#define D0_USED(d) (((d) >> 10) & 1023u)
#define D0_STREAK(d) (((d) >> 20) & 1023u)
#define D0_LAST(d) (((d) >> 30) & 1u)
#define D0_ATB(d) (((d) >> 31) & 1u)
#define D1_HIST(d) ((d) & 0x3FFFu)
#define D1_NDAYS(d) (((d) >> 16) & 1023u)
#define D1_FIN(d) (((d) >> 31) & 1u)

__device__ __forceinline__ uint32_t pack_d0(uint32_t t, uint32_t used, uint32_t streak, uint32_t last, uint32_t atb) {
  return (t & 1023u) | ((used > 1023u ? 1023u : used) << 10) | ((streak > 1023u ? 1023u : streak) << 20) |
         (last << 30) | (atb << 31);
}
__device__ __forceinline__ uint32_t pack_d1(uint32_t hist, uint32_t ndays, uint32_t fin) {
  return (hist & 0x3FFFu) | ((ndays & 1023u) << 16) | (fin << 31);
}

struct DevTables {
  const float4 *X;
  const int32_t *n_days;
  const int32_t *B0;
  const float4 *W;
  const int32_t *fips_to_weather;
  const int32_t *sim_cnt;
  const uint32_t *gate_bits;       // [T][gate_words] bitmap of the slot-30 gate flags; nullable
  int32_t gate_words;
  const int32_t *sim_ptr;          // [S+1] CSR of similar(county) ∩ fips_list (only for W2A_FIX_AUGMENT); nullable
  const int32_t *sim_idx;
  int32_t T, S_w, Y, S, n_samples, n_obs;
  int32_t slot_hist2w;             // table slot of the historical 'alerts_2wks' column (-1: none)
  uint32_t fixes;                  // W2A_FIX_* bits: opt-in corrections of reference quirks (0 = faithful)
};
// overwrite component `idx` (0 .. 4*QUADS-1) of a lane's row fragment without dynamic register indexing
__device__ __forceinline__ void set_comp(float4 *x, int idx, float v) {
#pragma unroll
  for (int q = 0; q < QUADS; ++q) {
    if (idx == 4 * q) x[q].x = v;
    if (idx == 4 * q + 1) x[q].y = v;
    if (idx == 4 * q + 2) x[q].z = v;
    if (idx == 4 * q + 3) x[q].w = v;
  }
}
#define SAMPLE_BITS 12
#define PACK_W(coef_col, sample) (((uint32_t)(coef_col) << SAMPLE_BITS) | (uint32_t)(sample))
#define W_COL(y) ((y) >> SAMPLE_BITS)
#define W_SAMPLE(y) ((y) & ((1u << SAMPLE_BITS) - 1u))

struct ResetCfg {
  uint64_t seed;
  int32_t location;
  int32_t augment;
  int32_t budget_kw;
  int32_t sample_mode;
  int32_t sticky;
};

struct w2a_env {
  DevTables tb;
  int64_t n;
  int64_t gid0;
  const int32_t *slot_obs;  // [32] slot -> obs column (-1 none), in the state header
  StateArrays st;
  int32_t *status;
  ResetCfg autoreset;
  int has_autoreset;
  int32_t obs_slot_host[ROWF];
  const uint32_t *perm;  // env ids sorted by coefficient column (w2a_group_by_column), valid until the next reset
  uint4 *prep;           // per-step scratch of the posterior-mean path, inside the same workspace
  const uint32_t *inv;   // sorted position of every env, in the same workspace
  const double *wd;      // fp64 copy of W scaled by -log2(e), in the same workspace
  const uint4 *tiles;    // tile list of the posterior-mean kernel, in the same workspace
  const uint32_t *n_tiles;
  // int8 matrix-core form of the posterior-mean reward (w2a_posterior_i8.hip.h), in the same workspace
  const uint4 *tiles_i8;
  const uint32_t *n_tiles_i8;
  const uint32_t *wq;
  const float *wscale;
  const uint32_t *colflag;
  const float *xs;
  const void *xmax_ws;   // workspace whose slot maxima (once-per-table scan) are valid
  // matrix-core rollout (w2a_rollout_mfma.hip.h): tile list by feature row + digit table of W, in its own workspace
  // order workspace (w2a_rollout_order_attach / w2a_rollout_order): envs per feature row, each env's rank inside its row,
  // first position and first 64-env tile of every row (rows + 1 entries each)
  void *order_ws;
  uint32_t *order_cnt, *order_rank, *order_start, *order_tile_start;
  const void *rm_ws;             // workspace whose W digit table is built
  const uint4 *rm_tiles;
  const uint32_t *rm_n_tiles;
  const uint32_t *rm_wq;
  const float *rm_xs;
  const uint32_t *order; // visiting order of k_rollout (w2a_rollout_order), any permutation is correct; NULL = identity
  // which form of the per-env step state is current, what is known about days and budgets, which derived structures
  // are still valid: w2a_bookkeeping.h (plain C++, exercised on the CPU under sanitizers by tests/test_bookkeeping_cpu.py)
  W2aBook bk;
  int pm_kernel;         // W2A_PM_* : which posterior-mean reward kernel w2a_posterior_mean_reward launches
  int w_tail_used;       // some coefficient row uses slot 28, 30 or 31 (scanned once by w2a_create)
};

static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, const char *a = "") {
  snprintf(g_err, sizeof(g_err), fmt, a);
  return code;
}

// ----------------------------------------------------------------------------------------
// counter-based RNG (restated in oracle/heatalert_oracle.py: devrng_*)
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t w2a_mix64(uint64_t z) {
  z ^= z >> 30;
  z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27;
  z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return z;
}
__device__ __forceinline__ uint64_t rng_stream(uint64_t seed, uint64_t gid, uint64_t episode) {
  uint64_t h = w2a_mix64(seed + 0x9E3779B97F4A7C15ull * (gid + 1));
  return w2a_mix64(h ^ (episode * 0xBF58476D1CE4E5B9ull + 0x94D049BB133111EBull));
}
// uniform integer in [0, n): multiply-shift on the high 32 bits of the slot's word
__device__ __forceinline__ uint32_t rng_bounded(uint64_t stream, uint32_t slot, uint32_t n) {
  uint64_t u = w2a_mix64(stream + (uint64_t)(slot + 1) * 0x9E3779B97F4A7C15ull) >> 32;
  return (uint32_t)((u * (uint64_t)n) >> 32);
}

// ----------------------------------------------------------------------------------------
// cross-lane helpers (LANES-lane groups inside a DPP row of 16)
// ----------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// all-reduce (sum) over the LANES lanes of a group; every lane ends with the total
__device__ __forceinline__ double group_sum(double v) {
  if (LANES >= 2) v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]  : lane ^ 1
  if (LANES >= 4) v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]  : lane ^ 2
  if (LANES >= 8) v += dpp_f64<0x141>(v);  // row_half_mirror      : lane -> 7 - lane (other quad of the group)
  return v;
}

__device__ __forceinline__ float sigmoid_f32(float z) {
  // 1/(1+exp(-z)); exp(-z) overflows to +inf for z << 0 which gives exactly 0, and z = -inf
  // (closed effectiveness gate) also gives exactly 0.
  float e = __expf(-z);                      // v_mul + v_exp_f32
  return __builtin_amdgcn_rcpf(1.0f + e);    // v_rcp_f32 (1 ulp); rcp(+inf) = 0
}

// env.py:211-221 from the two float64 logits: f32 sigmoids (v_exp_f32 / v_rcp_f32) and reward. The reference's epilogue is
// float64 expit; against it this form measures <= 1.1e-6 on the reward (bar 1e-5; a float64 exp / divide build variant
// existed until round 3 -- +13 % kernel time for 2.4e-7 -- and was removed untested rather than kept unexercised).
// exp(-z) -> +inf gives exactly 0 (closed gate: ze = -inf).
__device__ __forceinline__ float reward_from_logits(double zb, double ze, uint32_t actual) {
  const float base = sigmoid_f32((float)zb);
  const float eff = sigmoid_f32((float)ze);
  return -(1000.0f / 152.0f) * base * (1.0f - eff * (float)actual);
}

// ----------------------------------------------------------------------------------------
// episode draw shared by the reset kernel and the same-step autoreset (env.py:145-178)
// ----------------------------------------------------------------------------------------
struct Episode {
  uint32_t ep_row, ep_w, ndays;
  int32_t budget, sticky;
  uint32_t bad;
};

__device__ __forceinline__ Episode draw_episode(const DevTables &tb, const ResetCfg &rc, uint64_t gid,
                                                uint32_t episode_no, int32_t sticky_in) {
  Episode e;
  uint64_t st = rng_stream(rc.seed, gid, episode_no);
  uint32_t bad = 0;
  uint32_t county = rc.location < 0 ? rng_bounded(st, 0, (uint32_t)tb.S) : (uint32_t)rc.location;
  if (county >= (uint32_t)tb.S) { county = 0; bad = 1; }
  uint32_t coef_col = county;
  if (rc.augment) {
    int32_t ns = tb.sim_cnt[county];
    if (ns <= 0) { bad = 1; ns = 1; }
    coef_col = rng_bounded(st, 1, (uint32_t)ns);  // position inside the filtered list
    if ((tb.fixes & W2A_FIX_AUGMENT) && tb.sim_idx) {
      // corrected augmentation: the drawn similar county supplies both the weather and the coefficients.
      // An empty list (county absent from the confounders: the reference raises KeyError, datautils.py:123)
      // is never indexed; the looked-up county is range-checked before it indexes any table.
      if (!bad) county = (uint32_t)tb.sim_idx[tb.sim_ptr[county] + (int32_t)coef_col];
      if (county >= (uint32_t)tb.S) { county = 0; bad = 1; }
      coef_col = county;
    }
  }
  uint32_t year_i = rng_bounded(st, 2, (uint32_t)tb.Y);
  uint32_t sample = rng_bounded(st, 3, (uint32_t)tb.n_samples);
  int32_t cw = tb.fips_to_weather[county];
  if (cw < 0) { cw = 0; bad = 1; }
  e.ep_row = (uint32_t)cw * (uint32_t)tb.Y + year_i;
  e.ep_w = PACK_W(coef_col, sample);
  int32_t nd = tb.n_days[e.ep_row];
  if (nd <= 0) { bad = 1; nd = 1; }
  e.ndays = (uint32_t)nd;
  int32_t b = (rc.sticky && sticky_in >= 0) ? sticky_in : (rc.budget_kw < 0 ? tb.B0[e.ep_row] : rc.budget_kw);
  if (b < 0) b = 0;
  if (rc.sample_mode == W2A_BUDGET_LESS_THAN) {
    b = (int32_t)rng_bounded(st, 4, (uint32_t)b + 1u);
  } else if (rc.sample_mode == W2A_BUDGET_CENTERED) {
    // rng.integers(0.5*b, 1.5*b + 1): NumPy truncates the float bounds
    int32_t lo = (int32_t)(0.5 * (double)b), hi = (int32_t)(1.5 * (double)b + 1.0);
    b = lo + (int32_t)rng_bounded(st, 4, (uint32_t)(hi - lo));
  }
  e.budget = b;
  e.sticky = rc.sticky ? b : -1;  // self.budget keeps the (sampled) value (env.py:167-178, Q9)
  e.bad = bad;
  return e;
}

// ----------------------------------------------------------------------------------------
// observation tile: wave-level transpose through LDS, 16-B coalesced stores
// ----------------------------------------------------------------------------------------
// x        : this lane's QUADS float4 of the row (slots 4*(l*QUADS+q)..), run-time fields already patched
// so       : obs column of each of those slots (-1 = not part of the observation)
// write_me : this env's row must be written (false -> keep what is in memory)
__device__ __forceinline__ void store_obs_tile(void *__restrict__ obs_any, float *tile, int64_t wave_env0, int64_t n,
                                               int n_obs, int lane, int grp, const float4 *x, const int4 *so,
                                               bool write_me) {
  float *obs = reinterpret_cast<float *>(obs_any);
  // branch-free scatter: slots that are not observation columns (bias, gate flag, pads; so < 0) go to a scratch
  // word behind the packed rows (the tile has ENVS_PER_WAVE*32 floats, the rows use ENVS_PER_WAVE*n_obs <= 30*16)
  const int base = grp * n_obs;
  const int trash = ENVS_PER_WAVE * n_obs + (lane & 15);
#pragma unroll
  for (int q = 0; q < QUADS; ++q) {
    tile[so[q].x >= 0 ? base + so[q].x : trash] = x[q].x;
    tile[so[q].y >= 0 ? base + so[q].y : trash] = x[q].y;
    tile[so[q].z >= 0 ? base + so[q].z : trash] = x[q].z;
    tile[so[q].w >= 0 ? base + so[q].w : trash] = x[q].w;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const bool full = (wave_env0 + ENVS_PER_WAVE <= n);
  const bool all_write = __all(write_me || (wave_env0 + grp >= n));
  float *dst = obs + wave_env0 * n_obs;
  if (full && all_write) {
    const int chunks = (ENVS_PER_WAVE * n_obs) >> 2;  // ENVS_PER_WAVE*n_obs floats is a multiple of 4
#pragma unroll
    for (int c0 = 0; c0 < (ENVS_PER_WAVE * ROWF) / 4; c0 += 64) {
      const int c = c0 + lane;
      if (c < chunks) {
        v4f v = reinterpret_cast<const v4f *>(tile)[c];
#if W2A_NT_OBS
        __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(dst) + c);
#else
        reinterpret_cast<v4f *>(dst)[c] = v;
#endif
      }
    }
  } else {
    // ragged tail or some env of the wave keeps its stale row: element-wise, masked
    const int total = ENVS_PER_WAVE * n_obs;
    const unsigned long long wm = __ballot(write_me);  // taken before the loop: every lane still active
    for (int i = lane; i < total; i += 64) {
      int g = i / n_obs;
      bool w = (wm >> (g * LANES)) & 1ull;
      if (w && wave_env0 + g < n) dst[i] = tile[i];
    }
  }
  __builtin_amdgcn_wave_barrier();
}

#endif  // W2A_COMMON_HIP_H
