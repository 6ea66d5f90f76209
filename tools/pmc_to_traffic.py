#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the condensed rocprofv3 counter summaries: bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB;
the gfx950 correction of MI355X_MICROARCH.md, calibrated in round 1/2 on shade_kernel<false>'s known 32 B/pixel read), and the VALU-active
share of shade_kernel<true> from round 2's SQ / GRBM passes.   usage: pmc_to_traffic.py <dir with r03_pmc_*.csv> [--write]"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1]


def counters(path, kernel):
    out = {}
    if not os.path.exists(path):
        return out
    for row in csv.reader(open(path)):
        if len(row) >= 8 and row[0].startswith(kernel):
            out.setdefault(row[5], []).append((int(row[6]), float(row[10]) if len(row) > 10 else float(row[7])))   # steady state: the last 40 dispatches
    # the loop's launches are the entry with most dispatches
    return {k: max(v)[1] for k, v in out.items()}


def bytes_of(tag, kernel, rnd="r03"):
    f = counters(os.path.join(d, f"{rnd}_pmc_{tag}_FETCH_SIZE.csv"), kernel).get("FETCH_SIZE")
    w = counters(os.path.join(d, f"{rnd}_pmc_{tag}_WRITE_SIZE.csv"), kernel).get("WRITE_SIZE")
    return None if f is None or w is None else (2 * f + w) * 1024.0


tj_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
sys.path.insert(0, ROOT)
from materialist_amd.build import sources_digest  # noqa: E402

# the kernels these bytes were counted on: bench.py reports `traffic` only while csrc/ still hashes to this
tj["csrc_sha16"] = sources_digest()
# round 4: the folded, persistent step (lazy_pstep_kernel), the walk launch behind it, the statistics pass
RND = os.environ.get("PMC_ROUND", "r05")
for tag, b in (("b8", 8), ("b1", 1)):
    for name, kern in (("lazy_pstep", "matpbr::lazy_pstep_kernel"), ("lazy_pwalk", "matpbr::lazy_pwalk_kernel"), ("loss_sums2_r04", "loss_sums2_kernel<")):
        v = bytes_of(tag, kern, RND)
        if v is not None:
            tj[f"{name}_512x512_b{b}_spp64"] = v
if os.path.exists(os.path.join(d, f"{RND}_pmc_b8_FETCH_SIZE.csv")):
    tj["source_r04"] = (f"profiles/{RND}_pmc_{{b1,b8}}_{{FETCH_SIZE,WRITE_SIZE}}.csv (rocprofv3 --kernel-trace --pmc, separate passes, tools/pmc_passes_{RND}.sh); "
                        "bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB), the gfx950 correction of MI355X_MICROARCH.md as calibrated in round 2")
sq4 = counters(os.path.join(d, f"{RND}_pmc_b8_sq.csv"), "matpbr::lazy_pstep_kernel")
gr4 = counters(os.path.join(d, f"{RND}_pmc_b8_grbm.csv"), "matpbr::lazy_pstep_kernel")
if "SQ_ACTIVE_INST_VALU" in sq4 and "GRBM_GUI_ACTIVE" in gr4:
    tj["lazy_pstep_valu_active_frac"] = sq4["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * gr4["GRBM_GUI_ACTIVE"] / 8.0)
for tag, b in (("b8", 8), ("b1", 1)):
    for name, kern in (("lazy_step", "matpbr::lazy_step_kernel"), ("loss_sums2", "loss_sums2_kernel<")):
        v = bytes_of(tag, kern)
        if v is not None:
            tj[f"{name}_512x512_b{b}_spp64"] = v
tj["source_r03"] = ("profiles/r03_pmc_{b1,b8}_{FETCH_SIZE,WRITE_SIZE}.csv (rocprofv3 --kernel-trace --pmc, separate passes, tools/pmc_passes_r03.sh); "
                    "bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB), the gfx950 correction of MI355X_MICROARCH.md as calibrated in round 2")
# VALU-active share of the exact forward (round 2's passes; profiles/README.md): SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
pd = os.path.join(ROOT, "profiles")
sq = counters(os.path.join(pd, "r02_pmc_b8_sq.csv"), "matpbr::shade_kernel<true>")
gr = counters(os.path.join(pd, "r02_pmc_b8_grbm.csv"), "matpbr::shade_kernel<true>")
if "SQ_ACTIVE_INST_VALU" in sq and "GRBM_GUI_ACTIVE" in gr:
    tj["shade_kernel_jac_valu_active_frac"] = sq["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * gr["GRBM_GUI_ACTIVE"] / 8.0)
sq3 = counters(os.path.join(d, "r03_pmc_b8_sq.csv"), "matpbr::lazy_step_kernel")
gr3 = counters(os.path.join(d, "r03_pmc_b8_grbm.csv"), "matpbr::lazy_step_kernel")
if "SQ_ACTIVE_INST_VALU" in sq3 and "GRBM_GUI_ACTIVE" in gr3:
    tj["lazy_step_valu_active_frac"] = sq3["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * gr3["GRBM_GUI_ACTIVE"] / 8.0)
print(json.dumps({k: v for k, v in tj.items() if "r03" in k or "r04" in k or "lazy" in k or "valu" in k or "sums2" in k}, indent=1))
if "--write" in sys.argv:
    json.dump(tj, open(tj_path, "w"), indent=1)
