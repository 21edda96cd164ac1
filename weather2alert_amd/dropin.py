"""HeatAlertEnv: the num_envs = 1 drop-in for weather2alert.env.HeatAlertEnv (env.py:17-262) on top of HeatAlertVecEnv."""
from __future__ import annotations

from typing import Literal

import numpy as np
import torch

from . import _ffi
from .env import HeatAlertVecEnv
from .tables import CompiledTables, DeviceTables


class HeatAlertEnv:
    """Drop-in for ``weather2alert.env.HeatAlertEnv`` (env.py:17-262): same constructor,
    ``reset(location, similar_climate_counties, seed, budget, sample_budget, sample_budget_type)``
    and ``step(action) -> (obs, reward, done, False, info)``; one env on the GPU, episode draws
    replayed from NumPy's Generator so identical seeds give identical episodes.

    Differences by construction: observations are float32 arrays of the true width 29 with
    ``alert`` as 0/1 and ``significance`` as a category code (the reference returns an object
    array mixing floats, ints, bools and strings)."""

    def __init__(self, weights: str = "nn_full_medicare_all", years: list | None = None,
                 fips_list: list | None = None, similar_climate_counties: bool = False, budget: int | None = None,
                 data_dir: str | None = None, split: str = "65k", device: str = "cuda:0",
                 tables: CompiledTables | DeviceTables | None = None):
        self._v = HeatAlertVecEnv(1, weights, years, fips_list, similar_climate_counties, budget, data_dir, split,
                                  device, seed_mode="numpy_parity", autoreset="disabled", tables=tables)
        ct = self._v.ct
        self.fips_list, self.valid_years, self.n_samples = ct.fips_list, ct.years, ct.n_samples
        self.similar_climate_counties = similar_climate_counties
        self.observation_space = self._v.single_observation_space
        self.action_space = self._v.single_action_space
        self.feat_names = ct.feature_names
        self._act = torch.zeros(1, dtype=torch.int32, device=self._v.device)

    @property
    def budget(self):
        return self._v._sticky[0]

    def _sync_state(self):
        """One device-to-host copy per call: observation, reward, done, the status word and every decoded state
        field are concatenated on the device (the reference's API returns Python scalars, so each call must
        synchronise once -- but only once)."""
        v = self._v
        buf, _ = v._state_packed()
        pack = torch.cat([v._obs.view(torch.int32).reshape(-1), v._reward.view(torch.int32),
                          v._done.to(torch.int32), v._status, buf.reshape(-1)])
        host = pack.cpu().numpy()
        n_obs = v.ct.n_obs
        self._h_obs = host[:n_obs].view(np.float32).copy()
        self._h_reward = float(host[n_obs: n_obs + 1].view(np.float32)[0])
        self._h_done = bool(host[n_obs + 1])
        bits = int(host[n_obs + 2])
        if bits:  # rare: let check_status() read-and-clear the word and raise what the reference raises
            v.check_status()
        fields = host[n_obs + 3:]
        st = {k: fields[i] for i, k in enumerate(_ffi.STATE_FIELDS)}
        ct = v.ct
        self.t = int(st["t"])
        self.alert_streak = int(st["streak"])
        self.coef_index = int(st["sample"])
        self.location_index = int(st["coef_col"])
        self.remaining_budget = int(st["budget"] - st["used"])
        self.at_budget = bool(st["at_budget"])
        self.n_days = int(st["n_days"])
        self.location = v._info_location[0]
        self.ep_index = ct.fips_weather[int(st["county_w"])] + "_" + str(ct.years[int(st["year_i"])])
        return st

    def _get_info(self):
        return {"episode_index": self.ep_index, "remaining_budget": self.remaining_budget,
                "at_budget": self.at_budget, "feature_names": self.feat_names, "location": self.location,
                "location_index": self.location_index}

    def reset(self, location: str | None = None, similar_climate_counties: bool | None = None,
              seed: int | None = None, budget: int | None = None, sample_budget: bool = False,
              sample_budget_type: Literal["less_than", "centered"] = "less_than"):
        if seed is None:
            seed = np.random.randint(0, 10000)
        obs, _ = self._v.reset(seed=[seed], options=dict(
            location=location, similar_climate_counties=similar_climate_counties, budget=budget,
            sample_budget=sample_budget, sample_budget_type=sample_budget_type))
        self._sync_state()
        self.observation = self._h_obs
        return self.observation, self._get_info()

    def step(self, action: int):
        self._act.fill_(int(action))
        self._v.step(self._act)
        self._sync_state()
        self.observation = self._h_obs
        return self.observation, self._h_reward, self._h_done, False, self._get_info()

    def close(self):
        self._v.close()
