"""profiles/hbm_traffic.json from the per-kernel PMC summaries that
tools/profile_round.sh leaves in profiles/r01_<workload>_pmc_traffic.json.

FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B; confirmed for
8-byte lanes with tools/hbm_calib.hip), WRITE_SIZE is taken as read; launches
of the stage kernel's variants are weighted by how often each was seen."""
import json, sys
out = {}
pm = json.load(open("profiles/r01_rk4_pmc_traffic.json"))
tiles = {k: v for k, v in pm.items() if "k_residual_tile" in k}
n = sum(v["launches_seen"] for v in tiles.values())
fetch = sum(v["FETCH_SIZE"] * v["launches_seen"] for v in tiles.values()) / n * 1024 * 2
write = sum(v["WRITE_SIZE"] * v["launches_seen"] for v in tiles.values()) / n * 1024
out["rk4"] = {
    "cells": 256 ** 3, "kernel": "agx::k_residual_tile<MUSCL,vanAlbada,Roe,fused> "
    "(stage-0 variant that also stores consVarsN weighted 1 in 4)",
    "bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
    "algorithmic_bytes": 296 * 256 ** 3,
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes "
              "(tools/profile_round.sh); FETCH_SIZE KB x1024 x2 (gfx950 counts 128-B "
              "requests at 64 B; factor confirmed for this library's 8-B-per-lane loads "
              "with tools/hbm_calib.hip: 1 GiB streamed read reports 524292.5 KB, 1 GiB "
              "streamed write reports 1048576 KB), WRITE_SIZE KB x1024",
    "source": "profiles/r01_rk4_pmc_traffic.json"}
json.dump(out, open("profiles/hbm_traffic.json", "w"), indent=1)
print(out["rk4"]["bytes_per_launch"] / 1e9, "GB per launch")
