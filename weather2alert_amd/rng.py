"""Host replay of the reference's reset() draws with NumPy's own Generator (seed parity).

``numpy_parity_episode`` makes, for one env, exactly the calls ``HeatAlertEnv.reset`` makes on
``np.random.default_rng(seed)`` (reference ``src/weather2alert/env.py:145-177``) and returns the episode
tuple the device needs. It is pure host code (no GPU) so that the CPU test-suite can check it against the
golden vectors captured from the reference.
"""
from __future__ import annotations

import numpy as np

from .tables import CompiledTables


def numpy_parity_episode(ct: CompiledTables, seed: int, location: str | None, augment: bool,
                         sticky_budget: int | None, budget_kw: int | None, sample_budget: bool,
                         sample_budget_type: str, fix_augment: bool = False):
    """Returns (county_w, year_i, coef_col, sample, budget, info_location).

    Draw order: choice(fips_list) if location is None; choice(range(n_similar)) if augment;
    choice(valid_years); integers(0, n_samples); budget integers if sample_budget. Raises what the reference
    raises: ValueError for a location outside fips_list without augmentation (env.py:121), KeyError for a
    county absent from the confounders (datautils.py:123) or a (county, year) without data (env.py:127)."""
    rng = np.random.default_rng(seed)
    if location is None:
        location = str(rng.choice(ct.fips_list))  # env.py:151-152
    info_location = location
    if augment:
        # env.py:115-118: the requested county is looked up in the confounders only (KeyError, datautils.py:123),
        # never in fips_list -- a county outside fips_list still augments
        sim = ct.similar_for_fips(location)
        if len(sim) == 0:
            raise ValueError("a must be a positive integer unless no samples are taken")  # rng.choice(range(0))
        li = int(rng.choice(range(len(sim))))  # env.py:117: index into the FILTERED list (Q8)
        drawn = int(sim[li])
        info_location = ct.fips_list[drawn]  # env.py:118
        if fix_augment:  # corrected Q8: the drawn county supplies weather and coefficients
            li = drawn
            location = info_location
    else:
        li = ct.fips_index(location)  # env.py:121 list.index -> ValueError
    year = int(rng.choice(ct.years))  # env.py:125
    wpos = ct.__dict__.get("_weather_pos")
    if wpos is None:
        wpos = ct.__dict__["_weather_pos"] = {f: i for i, f in enumerate(ct.fips_weather)}
    w = wpos.get(location, -1)  # merged.loc[(location, year)] (env.py:127): the REQUESTED county's weather
    y_i = ct.years.index(year)
    if w < 0 or ct.n_days[w * ct.Y + y_i] <= 0:
        raise KeyError((location, year))  # env.py:127
    ci = int(rng.integers(0, ct.n_samples))  # env.py:160
    b = sticky_budget
    if b is None:  # env.py:167-170
        b = int(ct.B0[w * ct.Y + y_i]) if budget_kw is None else int(budget_kw)
    if sample_budget:  # env.py:172-177 (NumPy truncates the float bounds of the centered draw)
        if sample_budget_type == "less_than":
            b = int(rng.integers(0, b + 1))
        elif sample_budget_type == "centered":
            b = int(rng.integers(0.5 * b, 1.5 * b + 1))
    return w, y_i, li, ci, b, info_location
