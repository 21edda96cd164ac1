"""Episode statistics of on-device rollouts, computed on the device from rollout() outputs: what the reference's SB3
logging callbacks report (/root/reference/src/weather2alert/callbacks.py:61-77 AlertLoggingCallback._on_rollout_end,
:116-157 FinalEvalCallback), same keys and definitions. Exposed as static methods of HeatAlertVecEnv too."""
from __future__ import annotations

import torch


def episode_stats(out: dict) -> dict:
    """Batch summary of finished rollouts: mean return, alerts per episode, alerts attempted over budget per
    episode, the histogram of alert days (needs rollout(..., alert_mask=True)) and, for whole-episode rollouts
    with alert_mask=True, everything the reference's logging callback reports (callback_stats: same keys and
    definitions as callbacks.py:61-77)."""
    s = {"mean_return": float(out["return"].double().mean()),
         "mean_alerts": float(out["alerts"].double().mean()),
         "mean_attempts_over_budget": float(out["attempts_over_budget"].double().mean())}
    if "alert_days" in out:
        s["alert_day_hist"] = out["alert_days"].sum(0).cpu()
        if bool(out["done"].all()) and bool((out["first_day"] == 0).all()):
            s.update(callback_stats(out))
    return s


def callback_stats(out: dict) -> dict:
    """What the reference's AlertLoggingCallback logs at the end of a rollout (callbacks.py:61-77), computed on
    the device from whole-episode rollout outputs (rollout(policy, alert_mask=True) from day 0): same keys, same
    definitions. The callback polls attributes of the legacy env; they map to the current env as listed in
    oracle/heatalert_oracle.py (attempted_alert_buffer, actual_alert_buffer for allowed_alert_buffer, "alert
    attempted at budget" for penalize, the running return for cum_reward). Definitions that follow from its code:
    streaks and alert days are those of ATTEMPTED alerts (:38-46), a streak is recorded when a no-alert day ends
    it (an open streak at the end of the window is dropped), alert days are env.t after the step
    (min(day + 1, n_days - 1)), the 50/80/100 % marks are the first index of the granted-alert list whose
    cumulative fraction reaches the mark, read when env.t == n_days - 2 (:47-57), as is the logged reward."""
    att, act = out["attempt_days"], out["alert_days"]
    n, T = att.shape
    nd = out["n_days"].long()
    day = torch.arange(T, device=att.device)
    live = day[None, :] < nd[:, None]
    att, act = att & live, act & live
    num_steps = int(nd.sum())
    s = {"training_rewards": float(out["return_snapshot"].double().nan_to_num(0.0).mean()),
         "over_budget_freq": float(out["attempts_over_budget"].sum()) / num_steps,
         "alerts_freq": float(att.sum()) / num_steps}
    t_after = torch.minimum(day[None, :] + 1, nd[:, None] - 1).double()
    w = t_after[att]
    s["average_t_alerts"] = float(w.mean()) if w.numel() else 0
    s["stdev_t_alerts"] = float(w.std(unbiased=False)) if w.numel() else 0
    # streaks of attempted alerts that a no-alert day ended inside the episode: run length at the day before
    a = att.to(torch.int32)
    c = a.cumsum(1)
    zero_c = torch.where(a == 0, c, torch.zeros_like(c)).cummax(1).values  # cumsum at the last 0 so far
    run = c - zero_c  # consecutive alerts ending at each day
    ended = (a[:, 1:] == 0) & (a[:, :-1] == 1) & live[:, 1:]
    lens = run[:, :-1][ended].double()
    s["average_streak"] = float(lens.mean()) if lens.numel() else 0
    s["stdev_streak"] = float(lens.std(unbiased=False)) if lens.numel() else 0
    # 50 / 80 / 100 % marks over the granted alerts of days 0 .. n_days-3 (the list when t == n_days - 2)
    seen = day[None, :] < (nd[:, None] - 2)
    g = (act & seen).to(torch.int64)
    cg = g.cumsum(1)
    tot = cg[:, -1:]
    has = tot[:, 0] > 0
    frac = cg.double() / tot.clamp(min=1).double()
    for key, q in (("alert_t_50%", 0.5), ("alert_t_80%", 0.8), ("alert_t_100%", 1.0)):
        hit = ((frac == 1.0) if q == 1.0 else (frac >= q)) & seen
        first = torch.where(hit, day[None, :], torch.full_like(cg, T)).min(1).values
        s[key] = float(first[has].double().mean()) if bool(has.any()) else float("nan")
    return s


CSV_FIELDS = ("year", "alert_budget", "sum_alerts", "reward", "average_t_alerts", "stdev_t_alerts",
              "average_streak", "stdev_streak", "alerts")  # callbacks.py:136-146


def episode_rows(out: dict, chunk: int = 65536) -> list[dict]:
    """One row per env in the format of the reference's FinalEvalCallback (callbacks.py:116-146) from whole-episode
    rollout outputs: year, alert_budget, sum_alerts and reward as read when env.t == n_days - 2 (:128-132), alert
    day / streak statistics over the GRANTED alerts of the whole episode (:118-126), and the granted-alert list.
    Every statistic is computed on the device from the day bitmaps (the same run-length formulation as
    callback_stats, per env instead of pooled); the host only formats the rows. The envs are processed `chunk` at a
    time: the [chunk, T] intermediates stay at ~100 MB however large the batch is."""
    n = out["alert_days"].shape[0]
    rows: list[dict] = []
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        rows.extend(_episode_rows_chunk({k: out[k][lo:hi] for k in (
            "alert_days", "n_days", "year", "budget", "return_snapshot")}))
    return rows


def _episode_rows_chunk(out: dict) -> list[dict]:
    act = out["alert_days"]
    n, T = act.shape
    dev = act.device
    nd = out["n_days"].long()
    day = torch.arange(T, device=dev)
    live = day[None, :] < nd[:, None]
    a = (act & live).to(torch.int32)  # counts <= T <= 1023: exact in int32 and in float64 below
    cnt = a.sum(1)
    # day of each granted alert as the callback sees it: env.t after the step = min(day + 1, n_days - 1)
    t_after = torch.minimum(day[None, :] + 1, nd[:, None] - 1).double()
    af = a.double()
    c = cnt.clamp(min=1).double()
    mean_t = (t_after * af).sum(1) / c
    std_t = (((t_after - mean_t[:, None]) ** 2) * af).sum(1).div(c).sqrt()
    del t_after, af
    # streaks of granted alerts ended by a no-alert day inside the episode: run length at the day before
    cs = a.cumsum(1, dtype=torch.int32)
    zero_c = torch.where(a == 0, cs, torch.zeros_like(cs)).cummax(1).values
    run = (cs - zero_c)[:, :-1].double()
    ended = ((a[:, 1:] == 0) & (a[:, :-1] == 1) & live[:, 1:]).double()
    del cs, zero_c
    ns = ended.sum(1)
    cn = ns.clamp(min=1.0)
    mean_s = (run * ended).sum(1) / cn
    std_s = (((run - mean_s[:, None]) ** 2) * ended).sum(1).div(cn).sqrt()
    del run, ended
    seen = (day[None, :] < (nd[:, None] - 2)).to(torch.int32)
    sum_alerts = (a * seen).sum(1)
    h = {k: v.cpu().numpy() for k, v in dict(
        act=a.to(torch.uint8), nd=nd, year=out["year"], bud=out["budget"], snap=out["return_snapshot"], cnt=cnt,
        mean_t=mean_t, std_t=std_t, ns=ns, mean_s=mean_s, std_s=std_s, sum_alerts=sum_alerts).items()}
    rows = []
    for i in range(n):
        read = h["nd"][i] >= 3  # the callback only fills these when it sees t == n_days - 2
        has_t, has_s = h["cnt"][i] > 0, h["ns"][i] > 0
        rows.append({
            "year": int(h["year"][i]) if read else 0, "alert_budget": int(h["bud"][i]) if read else 0,
            "sum_alerts": int(h["sum_alerts"][i]) if read else 0, "reward": float(h["snap"][i]) if read else 0,
            "average_t_alerts": float(h["mean_t"][i]) if has_t else 0,
            "stdev_t_alerts": float(h["std_t"][i]) if has_t else 0,
            "average_streak": float(h["mean_s"][i]) if has_s else 0,
            "stdev_streak": float(h["std_s"][i]) if has_s else 0,
            "alerts": h["act"][i, : h["nd"][i]].tolist() if read else []})
    return rows


def write_episode_csv(path: str, out: dict) -> None:
    """The per-episode CSV of the reference's FinalEvalCallback (callbacks.py:151-157): header = its field names."""
    import csv

    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(CSV_FIELDS))
        w.writeheader()
        for row in episode_rows(out):
            w.writerow(row)
