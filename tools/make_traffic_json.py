"""profiles/hbm_traffic.json from the per-kernel PMC summaries that
tools/profile_round.sh leaves in profiles/<round>_<workload>_pmc_traffic.json.

FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B; confirmed for this
library's 8- and 16-byte lanes with tools/hbm_calib.hip), WRITE_SIZE is taken as
read.  Per pass of an implicit iteration: sum over its kernels of (bytes per
launch x launches per iteration)."""
import json
import os
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
ITER = 4            # iterations of the PMC runs (tools/profile_round.sh: --steps 3 --warmup 1)
METHOD = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes "
          "(tools/profile_round.sh); FETCH_SIZE KB x1024 x2 (gfx950 counts 128-B requests "
          "at 64 B; tools/hbm_calib.hip: 1 GiB streamed read reports 524292.5 KB, 1 GiB "
          "streamed write 1048576 KB), WRITE_SIZE KB x1024")
PASS_OF = [("k_residual_tile", "R"), ("k_inv_residual", "R"), ("k_block_diag", "R"),
           ("k_visc", "R"), ("k_lusgs_prepare", "R"), ("k_implicit_begin", "R"),
           ("k_rans_faces", "R"), ("k_rans_cells", "R"),
           ("k_sweep_records", "F+B"), ("k_lusgs_kp", "F+B"), ("k_lusgs_plane", "F+B"),
           ("k_lusgs_pipe", "F+B"),
           ("k_dplur", "4"), ("k_matrix_resid", "M"), ("k_update", "U"), ("k_norm_final", "U")]


def bytes_of(v):
    return v.get("FETCH_SIZE", 0.0) * 1024 * 2 + v.get("WRITE_SIZE", 0.0) * 1024


out = {}
path = f"profiles/{rnd}_rk4_pmc_traffic.json"
if os.path.exists(path):
    pm = json.load(open(path))
    tiles = {k: v for k, v in pm.items() if "k_residual_tile" in k}
    n = sum(v["launches_seen"] for v in tiles.values())
    fetch = sum(v["FETCH_SIZE"] * v["launches_seen"] for v in tiles.values()) / n * 2048
    write = sum(v["WRITE_SIZE"] * v["launches_seen"] for v in tiles.values()) / n * 1024
    out["rk4"] = {
        "cells": 256 ** 3, "kernel": "agx::k_residual_tile<MUSCL,vanAlbada,Roe,fused> "
        "(stage-0 variant that also stores consVarsN weighted 1 in 4)",
        "bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
        "algorithmic_bytes": 296 * 256 ** 3, "method": METHOD, "source": path}
RANS_BYTES = 8 * ((7 + 19 + 7 + 3 + 29) + 2 * (7 * 7 + 19 + 29 + 7) + (7 * 7 + 19 + 29) + 3 * 7)
for wl, cells in (("lusgs", 256 ** 3), ("dplur8", 8 * 128 ** 3), ("rans4", 4 * 128 * 128 * 64)):
    path = f"profiles/{rnd}_{wl}_pmc_traffic.json"
    if not os.path.exists(path):
        continue
    pm = json.load(open(path))
    passes, kernels, total = {}, {}, 0.0
    for k, v in pm.items():
        per_iter = bytes_of(v) * v["launches_seen"] / ITER
        kernels[k] = {"bytes_per_launch": bytes_of(v),
                      "launches_per_iteration": v["launches_seen"] / ITER}
        for pat, ps in PASS_OF:
            if pat in k:
                passes[ps] = passes.get(ps, 0.0) + per_iter
                total += per_iter
                break
    out[wl] = {"cells": cells, "bytes_per_iteration": total, "passes": passes,
               "kernels": kernels,
               "algorithmic_bytes": {"lusgs": 1296, "dplur8": 1960, "rans4": RANS_BYTES}[wl] * cells,
               "method": METHOD, "source": path}
json.dump(out, open("profiles/hbm_traffic.json", "w"), indent=1)
for k, v in out.items():
    print(k, v.get("bytes_per_iteration", v.get("bytes_per_launch")) / 1e9, "GB")
