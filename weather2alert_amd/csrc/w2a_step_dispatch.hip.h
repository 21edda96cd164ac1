// w2a_step_dispatch.hip.h -- w2a_step: which step kernel serves a call (env.py:238-262 for all envs), and in which form
// the per-env state is streamed. Part of libw2a.so; included only by w2a_kernels.hip inside its extern "C" block.
// Kept apart from the rest of the C ABI so that weather2alert_amd.build.source_sha() -- the hash that ties a
// step-kernel profile (profiles/traffic_latest.json) to the sources it was collected on -- covers exactly what decides
// the step kernels: w2a_common / w2a_step / w2a_step64 and this file.
#ifndef W2A_STEP_DISPATCH_HIP_H
#define W2A_STEP_DISPATCH_HIP_H

int w2a_step(w2a_env *env, const void *actions, int action_dtype, float *obs, float *reward, uint8_t *done,
             float *last_return, int flags, void *stream) {
  if (!env || !actions || !reward || !done) return fail(W2A_ERR_ARG, "w2a_step: NULL argument");
  if (action_dtype < W2A_ACT_I32 || action_dtype > W2A_ACT_U8) return fail(W2A_ERR_ARG, "w2a_step: bad action_dtype");
  const bool no_obs = (flags & W2A_STEP_NO_OBS) != 0;
  const bool autoreset = (flags & W2A_STEP_AUTORESET) != 0;
  const bool given = (flags & W2A_STEP_REWARD_GIVEN) != 0;
  if (given && (autoreset || env->tb.fixes || (flags & W2A_STEP_CLASSIC)))
    return fail(W2A_ERR_ARG, "w2a_step: W2A_STEP_REWARD_GIVEN is served by the 64-envs-per-wave kernel only (no in-kernel "
                             "autoreset, no corrected-semantics flags)");
  if (!no_obs && !obs) return fail(W2A_ERR_ARG, "w2a_step: obs is NULL (pass W2A_STEP_NO_OBS for reward-only)");
  if (!no_obs && ((uintptr_t)obs & 15)) return fail(W2A_ERR_ARG, "w2a_step: obs must be 16-B aligned");
  if (autoreset && !env->has_autoreset) return fail(W2A_ERR_ARG, "w2a_step: W2A_STEP_AUTORESET needs w2a_set_autoreset first");
  StepArgs a;
  memset(&a, 0, sizeof(a));
  a.tb = env->tb; a.slot_obs = env->slot_obs; a.st = env->st;
  a.actions = actions; a.obs = obs; a.reward = reward; a.done = done; a.last_return = last_return;
  a.status = env->status; a.n = env->n; a.gid0 = env->gid0; a.rc = env->autoreset; a.act_dtype = action_dtype;
  a.skip_finished = (flags & W2A_STEP_SKIP_FINISHED) ? 1 : 0;
  a.next_step = (flags & W2A_STEP_NEXT_STEP) ? 1 : 0;
  if (a.next_step && !autoreset) return fail(W2A_ERR_ARG, "w2a_step: W2A_STEP_NEXT_STEP goes with W2A_STEP_AUTORESET");
  if (a.skip_finished && !given)
    return fail(W2A_ERR_ARG, "w2a_step: W2A_STEP_SKIP_FINISHED goes with W2A_STEP_REWARD_GIVEN (policy loops)");
  dim3 grid(grid_for(env->n)), block(BLOCK);
  hipStream_t s = (hipStream_t)stream;
  // Stream capture (hipGraph): a captured step is replayed later without the host's bookkeeping being run again, so
  // nothing that depends on it may be baked into the graph: every step kernel reads the day from device memory (the
  // packed variant from the mirror's per-tile day word), and no conversion between the two forms of the state is ever
  // recorded -- the recorded kernel's form stays the handle's primary form from then on (w2a_bookkeeping.h: bk_step,
  // graph_canon / graph_packed, bk_end_call).
  const bool capturing = stream_is_capturing(s);
  if (capturing && (flags & W2A_STEP_NO_CAPTURE))
    return fail(W2A_ERR_STATE, "w2a_step: the stream is recording a hipGraph, but this loop's episode boundaries are driven "
                               "from the host (W2A_STEP_NO_CAPTURE): a reset between two steps cannot be recorded. Let the "
                               "step kernel restart finished envs itself (W2A_STEP_AUTORESET; HeatAlertVecEnv(lockstep=False))");
  // measured on MI355X (profiles/r02/nsweep.log): below ~128 K envs the 4-lanes-per-env kernel wins (more, shorter
  // waves hide the two memory hops better: 5.0 vs 6.2 us at 65 536 envs), from there on the 64-envs-per-wave one
  const bool wide = (given || (flags & W2A_STEP_WIDE) || env->n >= W2A_S64_MIN_ENVS) && !(flags & W2A_STEP_CLASSIC);
  // which kernel, on which form of the per-env state; the form conversions (k_pack_state / k_unpack_state) are launched
  // from inside, and the day every env is on after this call is recorded there
  HipDev dv{env, s};
  const W2aBook before = env->bk;
  const BkStepPlan plan = bk_step(env->bk, dv, wide, autoreset, given, (flags & W2A_STEP_UNPACKED) != 0, capturing);
  if (plan.kernel < 0)
    return fail(W2A_ERR_STATE, "w2a_step: this step runs on the canonical state words, which are not current (the state is "
                               "in its packed lock-step form), and a conversion cannot be recorded into a hipGraph; call "
                               "w2a_get_state before capturing -- or, on a handle with a recorded packed step, do not "
                               "record a step of another kind");
  // a launch that fails leaves the state as it was: the bookkeeping goes back too (a conversion that ran stays noted)
#define W2A_STEP_TRY(expr) \
  do { hipError_t _e = (expr); \
       if (_e != hipSuccess) { bk_step_rollback(env->bk, before, plan); if (!capturing) bk_end_call(env->bk, dv); \
                               return fail(W2A_ERR_HIP, #expr ": %s", hipGetErrorString(_e)); } } while (0)
  if (plan.kernel != W2A_BK_STEP_CLASSIC) {
    // the lean 64-envs-per-wave form (w2a_step64.hip.h); a workgroup covers BLOCK * W2A_S64_TILES envs, the grid is a
    // multiple of 8 workgroups
    const int64_t per_wg = (int64_t)BLOCK * W2A_S64_TILES;
    const int64_t tiles = (env->n + per_wg - 1) / per_wg;
    dim3 grid64((unsigned)(((tiles + 7) / 8) * 8));
    if (plan.kernel == W2A_BK_STEP_PACKED) {
      // lock-step mirror (StateArrays::pk_hot / pk_c): 20 B in and 8 B out of per-env state instead of 28 and 12
      a.uni_nd = plan.uni_nd;
      if (autoreset) {  // a lock-step batch whose episodes restart inside the kernel (recorded loops, lockstep=False by choice)
        if (env->tb.fixes) {
          if (no_obs) hipLaunchKernelGGL((k_step64<false, false, true, true, true>), grid64, block, 0, s, a);
          else hipLaunchKernelGGL((k_step64<true, false, true, true, true>), grid64, block, 0, s, a);
        } else if (no_obs) hipLaunchKernelGGL((k_step64<false, false, true, true>), grid64, block, 0, s, a);
        else hipLaunchKernelGGL((k_step64<true, false, true, true>), grid64, block, 0, s, a);
      } else if (env->tb.fixes) {  // corrected-semantics flags: their own variants, the faithful kernels carry none of the code
        if (no_obs) hipLaunchKernelGGL((k_step64<false, false, true, false, true>), grid64, block, 0, s, a);
        else hipLaunchKernelGGL((k_step64<true, false, true, false, true>), grid64, block, 0, s, a);
      } else if (no_obs) hipLaunchKernelGGL((k_step64<false, false, true>), grid64, block, 0, s, a);
      else hipLaunchKernelGGL((k_step64<true, false, true>), grid64, block, 0, s, a);
      W2A_STEP_TRY(hipGetLastError());
      bk_end_call(env->bk, dv);
      return W2A_OK;
    }
    if (given) {
      if (no_obs) hipLaunchKernelGGL((k_step64<false, true>), grid64, block, 0, s, a);
      else hipLaunchKernelGGL((k_step64<true, true>), grid64, block, 0, s, a);
    } else if (env->tb.fixes) {
      if (autoreset) {
        if (no_obs) hipLaunchKernelGGL((k_step64<false, false, false, true, true>), grid64, block, 0, s, a);
        else hipLaunchKernelGGL((k_step64<true, false, false, true, true>), grid64, block, 0, s, a);
      } else {
        if (no_obs) hipLaunchKernelGGL((k_step64<false, false, false, false, true>), grid64, block, 0, s, a);
        else hipLaunchKernelGGL((k_step64<true, false, false, false, true>), grid64, block, 0, s, a);
      }
    } else if (autoreset) {  // batches that left lock step: the finished envs restart inside the kernel (rare epilogue)
      if (no_obs) hipLaunchKernelGGL((k_step64<false, false, false, true>), grid64, block, 0, s, a);
      else hipLaunchKernelGGL((k_step64<true, false, false, true>), grid64, block, 0, s, a);
    } else {
      if (no_obs) hipLaunchKernelGGL((k_step64<false, false>), grid64, block, 0, s, a);
      else hipLaunchKernelGGL((k_step64<true, false>), grid64, block, 0, s, a);
    }
    W2A_STEP_TRY(hipGetLastError());
    if (!capturing) bk_end_call(env->bk, dv);
    return W2A_OK;
  }
#define W2A_LAUNCH(AR, OB) \
  do { if (env->tb.fixes) hipLaunchKernelGGL((k_step<AR, OB, true>), grid, block, 0, s, a); \
       else hipLaunchKernelGGL((k_step<AR, OB, false>), grid, block, 0, s, a); } while (0)
  if (autoreset) { if (no_obs) W2A_LAUNCH(true, false); else W2A_LAUNCH(true, true); }
  else { if (no_obs) W2A_LAUNCH(false, false); else W2A_LAUNCH(false, true); }
#undef W2A_LAUNCH
  W2A_STEP_TRY(hipGetLastError());
#undef W2A_STEP_TRY
  if (!capturing) bk_end_call(env->bk, dv);
  return W2A_OK;
}

#endif  // W2A_STEP_DISPATCH_HIP_H
