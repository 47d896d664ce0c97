#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/<tag>_pmc_summary.csv and
profiles/traffic_pmc.json (read by bench.py for roofline.traffic).

HBM-byte bookkeeping follows /opt/skills/guides/MI355X_MICROARCH.md (section HBM): the two counters need
separate passes (TCC slots), both are in KiB, and on gfx950 FETCH_SIZE reports half of the bytes of a wide
coalesced stream, so the read side is doubled (an upper bound for our 8-byte-per-lane accesses, which the
guide calls uncalibrated); WRITE_SIZE is exact for streaming stores and float atomics."""
import collections, csv, json, os, sys

fetch_csv, write_csv, tag, system = sys.argv[1:5]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows_name(name):
    """k_rows<kind, single?, masks?> -> k_rows<kind>, + [single] for the FP32 variant (Born, chain-rule and GB rows are different kernels)"""
    inner = name[name.index("<") + 1:name.index(">")].split(",")
    return f"k_rows<{inner[0].strip()}>" + ("[single]" if len(inner) > 1 and inner[1].strip() == "true" else "")


def per_kernel(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "agbnp::" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("agbnp::", "")
            base = rows_name(name) if name.startswith("k_rows") else name.split("<")[0].replace("k_tree_cavity_five", "k_tree_cavity")  # (the five-launch mode's instantiation)
            d[base].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in d.items()}, {k: len(v) for k, v in d.items()}


f, nf = per_kernel(fetch_csv, "FETCH_SIZE")
w, _ = per_kernel(write_csv, "WRITE_SIZE")
rows = []
for k in sorted(f, key=lambda k: -(2 * f[k] + w.get(k, 0))):
    rows.append(dict(kernel=k, launches=nf[k], FETCH_SIZE_KiB=round(f[k], 1), WRITE_SIZE_KiB=round(w.get(k, 0), 1),
                     hbm_bytes_per_launch=int((2 * f[k] + w.get(k, 0)) * 1024)))
out = os.path.join(ROOT, "profiles", tag, "pmc_summary.csv")
with open(out, "w") as fh:
    cw = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
    cw.writeheader()
    cw.writerows(rows)
print(open(out).read())
dom = max(rows, key=lambda r: r["hbm_bytes_per_launch"]) if len(sys.argv) < 6 else next(r for r in rows if r["kernel"] == sys.argv[5])
head_file = os.path.join(ROOT, "profiles", tag, "profile_head.json")
head = json.load(open(head_file)) if os.path.exists(head_file) else None
json.dump(dict(kernel=dom["kernel"], system=system, profile_head=head, hbm_bytes_per_launch=dom["hbm_bytes_per_launch"],
               FETCH_SIZE_KiB=dom["FETCH_SIZE_KiB"], WRITE_SIZE_KiB=dom["WRITE_SIZE_KiB"],
               all_kernels_bytes_per_eval=sum(r["hbm_bytes_per_launch"] for r in rows),
               source=f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; profiles/{tag}/pmc_summary.csv"),
          open(os.path.join(ROOT, "profiles", "traffic_pmc.json"), "w"), indent=1)
