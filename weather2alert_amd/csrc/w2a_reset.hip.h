// w2a_reset.hip.h -- k_reset / k_init_state / k_get_state (env.py:133-184, 228-236)
// Part of libw2a.so; included only by w2a_kernels.hip (one translation unit, see the file comment there).
#ifndef W2A_RESET_HIP_H
#define W2A_RESET_HIP_H

// ----------------------------------------------------------------------------------------
// reset kernels (same lane-group geometry as k_step so the observation tile code is shared)
// ----------------------------------------------------------------------------------------
struct ResetArgs {
  DevTables tb;
  const int32_t *slot_obs;
  StateArrays st;
  const int32_t *county_w, *year_i, *coef_col, *sample, *budget;  // host-tuple mode
  const uint8_t *mask;
  void *obs;
  int32_t *status;
  int64_t n;
  int64_t gid0;
  ResetCfg rc;
  int32_t from_tuples;  // 0: device RNG draw, 1: caller's tuples, 2: observe only (state untouched), 3: device RNG draw of
                        // the env src_idx[e] (w2a_reset_device_rng_sorted: the new episodes land in coefficient-row order)
  const uint32_t *src_idx;  // [n] which env's draw index e receives (a permutation)
  const uint2 *src_zw;      // [n] by SOURCE env: {sticky budget, episode number of the new episode} (k_reset_keys)
  int32_t restart;      // device RNG draw: 1 = the per-env episode counter restarts at 0 (explicit re-seed)
  // whole-batch resets of a handle with an attached order workspace (w2a_rollout_order_attach; both NULL otherwise): envs
  // per feature row (zeroed by the caller before the launch) and each env's position inside its row -- the first pass
  // of the visiting order's counting sort (w2a_rollout.hip.h), one returning atomic per env behind the observation stores
  uint32_t *order_cnt, *order_rank;
};

__global__ __launch_bounds__(BLOCK) void k_reset(const ResetArgs a) {
  __shared__ __attribute__((aligned(16))) float s_tile[BLOCK / 64][ENVS_PER_WAVE * ROWF];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = lane & (LANES - 1);
  const int grp = lane / LANES;
  const uint32_t lb = logical_block(blockIdx.x, gridDim.x >> 3);
  const int64_t wave_env0 = ((int64_t)lb * (BLOCK / 64) + wave) * ENVS_PER_WAVE;
  if (wave_env0 >= a.n) return;
  const int64_t env = wave_env0 + grp;
  const bool valid = env < a.n;
  const uint32_t e = (uint32_t)(valid ? env : (a.n - 1));
  const bool sel = a.mask ? (a.mask[e] != 0) : true;
  uint4 cold = load_cold(a.st, e);
  uint32_t bad = 0;
  Episode ep;
  if (a.from_tuples == 2) {
    // observe only (w2a_observe): first observation of an already reset env, state untouched
    uint4 c2, hot;
    load_step_state(a.st, e, c2, hot);
    ep.ep_row = cold.x;
    ep.budget = (int32_t)hot.w;
    if (D0_T(hot.x) != 0) bad = 4;
  } else if (a.from_tuples == 1) {
    int32_t cw = a.county_w[e], yi = a.year_i[e], cc = a.coef_col[e], sm = a.sample[e];
    if (cw < 0 || cw >= a.tb.S_w) { cw = 0; bad = 1; }
    if (yi < 0 || yi >= a.tb.Y) { yi = 0; bad = 1; }
    if (cc < 0 || cc >= a.tb.S) { cc = 0; bad = 1; }
    if (sm < 0 || sm >= a.tb.n_samples) { sm = 0; bad = 1; }
    ep.ep_row = (uint32_t)cw * (uint32_t)a.tb.Y + (uint32_t)yi;
    ep.ep_w = PACK_W(cc, sm);
    int32_t nd = a.tb.n_days[ep.ep_row];
    if (nd <= 0) { nd = 1; bad |= 1; }
    ep.ndays = (uint32_t)nd;
    ep.budget = a.budget ? a.budget[e] : a.tb.B0[ep.ep_row];
    ep.sticky = (int32_t)cold.z;
  } else if (a.from_tuples == 3) {
    // the episode env `src` draws -- its global id, its sticky budget, its episode number: exactly what k_reset draws for
    // it in the iid order -- lands on index e: the relabelling of episode_order="sorted" without moving any record
    uint32_t src = a.src_idx[e];
    if (src >= (uint32_t)a.n) { src = e; bad = 1; }  // (cannot happen with the sorted indices; never index past the workspace)
    const uint2 zw = a.src_zw[src];
    ep = draw_episode(a.tb, a.rc, (uint64_t)(a.gid0 + src), zw.y, (int32_t)zw.x);
    bad = ep.bad;
    cold.w = zw.y - 1u;  // (stored below as cold.w + 1)
  } else {
    // an explicit re-seed restarts the per-env episode counter, so equal seeds give equal episodes
    // (env.py:143-145 re-creates the Generator); autoresets keep counting up from there
    if (a.restart) cold.w = 0xFFFFFFFFu;
    ep = draw_episode(a.tb, a.rc, (uint64_t)(a.gid0 + e), cold.w + 1, (int32_t)cold.z);
    bad = ep.bad;
  }
  float4 x[QUADS];
  int4 so[QUADS];
#pragma unroll
  for (int q = 0; q < QUADS; ++q) {
    x[q] = a.tb.X[ep.ep_row * (ROWF / 4) + l * QUADS + q];  // day 0
    so[q] = reinterpret_cast<const int4 *>(a.slot_obs)[l * QUADS + q];
  }
  if (l == RT_QUAD / QUADS) x[RT_QUAD % QUADS] = make_float4(0.0f, 0.0f, (float)ep.budget, 0.0f);
  if ((a.tb.fixes & W2A_FIX_ALERTS_2WKS) && a.tb.slot_hist2w >= 0 && l == a.tb.slot_hist2w / (4 * QUADS))
    set_comp(x, a.tb.slot_hist2w % (4 * QUADS), 0.0f);  // the agent's (empty) history replaces the column
  if (valid && sel && l == 0) {
    if (a.from_tuples != 2) {
      store_episode(a.st, e, make_uint4(ep.ep_row, ep.ep_w, (uint32_t)ep.sticky, cold.w + 1),
                    make_uint4(pack_d0(0, 0, 0, 0, 0), pack_d1(0, ep.ndays, 0), __float_as_uint(0.0f), (uint32_t)ep.budget));
    }
    if (bad & 1) atomicOr(a.status, (int)W2A_ST_BAD_EPISODE);
    if (bad & 4) atomicOr(a.status, (int)W2A_ST_STEP_AFTER_DONE);
    if (a.order_cnt) a.order_rank[e] = atomicAdd(&a.order_cnt[ep.ep_row], 1u);
  }
  if (a.obs) store_obs_tile(a.obs, s_tile[wave], wave_env0, a.n, a.tb.n_obs, lane, grp, x, so, sel);
}

// w2a_reset_device_rng_sorted, first pass: the key of the episode every env is about to draw -- its coefficient row
// (column, draw): 17 bits on the reference's tables, two 9-bit passes of the radix sort -- with the env's index, and what
// the second pass needs of the env's old record once other indices' records have been overwritten: its sticky budget and
// the new episode's number. (Envs of one coefficient row keep their index order: the sort is stable.)
__global__ void k_reset_keys(DevTables tb, ResetCfg rc, StateArrays st, int64_t n, int64_t gid0, int32_t restart,
                             uint32_t *keys, uint32_t *idx, uint2 *zw) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4 cold = st.cold[i];
  const uint32_t epno = restart ? 0u : cold.w + 1u;
  const Episode ep = draw_episode(tb, rc, (uint64_t)(gid0 + i), epno, (int32_t)cold.z);
  keys[i] = W_COL(ep.ep_w) * (uint32_t)tb.n_samples + W_SAMPLE(ep.ep_w);
  idx[i] = (uint32_t)i;
  zw[i] = make_uint2(cold.z, epno);
}

__global__ void k_init_state(StateArrays st, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)  // sticky = -1, episode_no = -1 (first reset -> 0); finished, so a step before reset() is flagged
    store_episode(st, (uint32_t)i, make_uint4(0u, 0u, 0xFFFFFFFFu, 0xFFFFFFFFu), make_uint4(0u, pack_d1(0, 1, 1), 0u, 0u));
}

__global__ void k_get_state(StateArrays st, int64_t n, int32_t Y, int32_t n_samples, w2a_state_view v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint4 c2, h;
  load_step_state(st, (uint32_t)i, c2, h);
  const uint4 c = load_cold(st, (uint32_t)i);
  if (v.t) v.t[i] = (int32_t)D0_T(h.x);
  if (v.used) v.used[i] = (int32_t)D0_USED(h.x);
  if (v.streak) v.streak[i] = (int32_t)D0_STREAK(h.x);
  if (v.last_actual) v.last_actual[i] = (int32_t)D0_LAST(h.x);
  if (v.at_budget) v.at_budget[i] = (int32_t)D0_ATB(h.x);
  if (v.hist14) v.hist14[i] = (int32_t)D1_HIST(h.y);
  if (v.n_days) v.n_days[i] = (int32_t)D1_NDAYS(h.y);
  if (v.budget) v.budget[i] = (int32_t)h.w;
  if (v.episode_return) v.episode_return[i] = __uint_as_float(h.z);
  if (v.county_w) v.county_w[i] = (int32_t)(c.x / (uint32_t)Y);
  if (v.year_i) v.year_i[i] = (int32_t)(c.x % (uint32_t)Y);
  if (v.coef_col) v.coef_col[i] = (int32_t)W_COL(c.y);
  if (v.sample) v.sample[i] = (int32_t)W_SAMPLE(c.y);
  if (v.sticky_budget) v.sticky_budget[i] = (int32_t)c.z;
  if (v.episode_no) v.episode_no[i] = (int32_t)c.w;
  if (v.finished) v.finished[i] = (int32_t)D1_FIN(h.y);
}

#endif  // W2A_RESET_HIP_H
