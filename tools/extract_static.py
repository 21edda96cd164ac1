"""One-off extraction of the *data* lists the synthetic generator needs.

Reads (in the build container only) the reference's weight configs and the DoE
climate-zone CSV and writes compact data files under weather2alert_amd/data/:

  fips_linear.txt                  746 county FIPS codes, order of
                                   /root/reference/weights/linear/config.yaml:16-762
  fips_nn_full_medicare_all.txt    720 codes, order of
                                   /root/reference/weights/nn_full_medicare_all/config.yaml:24-744
  ba_zones.csv                     fips,ba_zone for every county of
                                   /root/reference/data/raw/DoE_climate_zones.csv

These are data (public county identifiers / DoE Building-America zones), not code.
The YAML is parsed exactly like the reference does (yaml.safe_load + str(x),
env.py:74-75) so un-quoted entries such as 01049 come out as 5-char strings.
"""
import os
import sys

import pandas as pd
import yaml

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(__file__), "..", "weather2alert_amd", "data")


def main():
    os.makedirs(OUT, exist_ok=True)
    for name in ["linear", "nn_full_medicare_all"]:
        cfg = yaml.safe_load(open(f"{REF}/weights/{name}/config.yaml"))
        fips = [str(x) for x in cfg["fips_list"]]
        assert all(len(f) == 5 for f in fips), [f for f in fips if len(f) != 5][:5]
        with open(f"{OUT}/fips_{name}.txt", "w") as f:
            f.write("\n".join(fips) + "\n")
        print(name, len(fips), "num_samples", cfg.get("num_samples"))
    z = pd.read_csv(f"{REF}/data/raw/DoE_climate_zones.csv", dtype=str)
    z["fips"] = z["State FIPS"].str.zfill(2) + z["County FIPS"].str.zfill(3)
    z = z[["fips", "BA Climate Zone"]].rename(columns={"BA Climate Zone": "ba_zone"})
    z = z.drop_duplicates("fips").sort_values("fips")
    z.to_csv(f"{OUT}/ba_zones.csv", index=False)
    print("zones", len(z), z.ba_zone.value_counts().to_dict())


if __name__ == "__main__":
    sys.exit(main())
