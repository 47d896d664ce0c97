#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc pass of SQ counters per kernel (mean per launch) into a CSV with two derived shares:
VALU issue (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, both in quad-cycles summed over waves) and LDS bank conflicts
(SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = extra LDS cycles per LDS cycle)."""
import collections, csv, sys

src, dst = sys.argv[1:3]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(src)):
    if "agbnp::" not in r["Kernel_Name"]:
        continue
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("agbnp::", "")
    if name.startswith("k_rows"):  # (k_rows<kind, single?, masks?>: Born, chain-rule and GB rows are different kernels)
        inner = name[name.index("<") + 1:name.index(">")].split(",")
        name = f"k_rows<{inner[0].strip()}>" + ("[single]" if len(inner) > 1 and inner[1].strip() == "true" else "")
    else:
        name = name.split("<")[0].replace("k_tree_cavity_five", "k_tree_cavity")  # (the five-launch mode's instantiation)
    vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE",
        "SQ_LDS_BANK_CONFLICT"]
order = ["k_prep", "k_tree_cavity", "k_born_tiles", "k_rows<0>", "k_gb_tiles", "k_rows<2>", "k_dborn_tiles", "k_rows<1>", "k_tree_pseudo", "k_outputs"]
with open(dst, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "launches"] + cols + ["valu_share_of_wave_cycles", "lds_bank_conflict_share"])
    for k in [k for k in order if k in vals] + sorted(set(vals) - set(order)):
        m = {c: (sum(vals[k][c]) / len(vals[k][c]) if vals[k][c] else 0.0) for c in cols}
        n = max(len(v) for v in vals[k].values())
        w.writerow([k, n] + [int(m[c]) for c in cols] +
                   [round(m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"], 3) if m["SQ_WAVE_CYCLES"] else 0,
                    round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], 3) if m["SQ_LDS_IDX_ACTIVE"] else 0])
print(open(dst).read())
