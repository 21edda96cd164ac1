"""The info mapping of HeatAlertVecEnv.reset() / step(): the reference's keys (env.py:228-236), fetched from the device on
first access."""
from __future__ import annotations

from collections.abc import Mapping, Sequence

import torch

class _LazyStrings(Sequence):
    """The reference's per-env string entries of info (env.py:229-236) built on demand: element i is formatted
    when it is asked for, so iterating over an info mapping of a million envs does not build a million strings."""

    def __init__(self, n: int, fn):
        self._n, self._fn = int(n), fn

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._fn(j) for j in range(*i.indices(self._n))]
        i = int(i)
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        return self._fn(i)

    def __eq__(self, other):
        try:
            return len(other) == self._n and all(a == b for a, b in zip(self, other))
        except TypeError:
            return NotImplemented

    def __repr__(self):
        head = ", ".join(repr(self[i]) for i in range(min(self._n, 3)))
        return f"<{self._n} strings: {head}{', ...' if self._n > 3 else ''}>"


class _LazyInfo(Mapping):
    """info mapping whose entries are fetched from the device on first access (env.py:228-236); it shows the env's
    state at that moment. Device tensors: remaining_budget, at_budget, location_index (coefficient column),
    county_w, year, final_return, t; host values: feature_names, and the reference's string entries episode_index
    ("<fips>_<year>") and location (env.py:118: under augmentation the fips at the drawn position of the filtered
    similar-county list) as lazy sequences of num_envs strings. A real Mapping: get(), values(), items(), `in`
    and len() all agree with iteration."""

    _KEYS = ("remaining_budget", "at_budget", "location_index", "county_w", "year", "feature_names",
             "final_return", "t", "episode_index", "location")
    _HOST_KEYS = ("episode_index", "location")

    def __init__(self, env):
        self._env = env
        self._d: dict = {}
        self._np: dict = {}

    def _fill(self):
        if not self._d:
            e = self._env
            st = e.state()
            years = torch.as_tensor(e.ct.years, dtype=torch.int32, device=e.device)
            self._d.update(
                remaining_budget=st["budget"] - st["used"], at_budget=st["at_budget"].bool(),
                location_index=st["coef_col"], county_w=st["county_w"], year=years[st["year_i"].long()],
                feature_names=e.feature_names, final_return=e._final_return, t=st["t"])

    def _host_arr(self, k):
        if k not in self._np:
            self._np[k] = self._d[k].cpu().numpy()
        return self._np[k]

    def _host(self, k):
        e, ct, n = self._env, self._env.ct, self._env.num_envs
        if k == "episode_index":
            return _LazyStrings(n, lambda i: f"{ct.fips_weather[self._host_arr('county_w')[i]]}_"
                                             f"{self._host_arr('year')[i]}")
        if e.seed_mode == "numpy_parity" and getattr(e, "_info_location", None):
            loc = list(e._info_location)
            return _LazyStrings(n, lambda i: loc[i])
        # device-RNG episodes: whether the last reset augmented is a property of the reset call
        aug = bool(e._reset_cfg[2]) if e._reset_cfg is not None else False
        if not aug or "augment" in e.fixes:
            return _LazyStrings(n, lambda i: ct.fips_list[self._host_arr("location_index")[i]])

        def drawn(i):  # position li of the filtered similar list of the requested county (Q8)
            sl = ct.similar_list(ct.fips_index(ct.fips_weather[self._host_arr("county_w")[i]]))
            return ct.fips_list[int(sl[self._host_arr("location_index")[i]])]

        return _LazyStrings(n, drawn)

    def __getitem__(self, k):
        if k not in self._KEYS:
            raise KeyError(k)
        self._fill()
        if k in self._HOST_KEYS and k not in self._d:
            self._d[k] = self._host(k)
        return self._d[k]

    def __iter__(self):
        return iter(self._KEYS)

    def __len__(self):
        return len(self._KEYS)
