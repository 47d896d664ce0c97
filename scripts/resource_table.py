#!/usr/bin/env python3
"""Per-kernel resource table of the gfx950 code objects (no GPU needed): VGPRs, SGPRs, spills, scratch, static LDS,
occupancy, and the number of scratch_/v_readlane/v_writelane instructions in the body.  Compiles every .hip file of
the engine to device assembly with the product's flags and reads the .amdhsa_ directives / amdhsa.kernels metadata.

  python scripts/resource_table.py [--out profiles/r02/resource_table.csv] [--keep-asm DIR]
"""
import argparse
import csv
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "openmm_agbnp_plugin_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-DAGBNP_TREE_BLOCK=192", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-S", "--cuda-device-only"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), text=True, capture_output=True).stdout.split("\n")
    return dict(zip(names, out))


def parse(asm_path):
    text = open(asm_path).read()
    rows = {}
    # metadata block: one YAML-ish entry per kernel
    for m in re.finditer(r"- \.agpr_count:.*?(?=\n  - \.agpr_count:|\namdhsa\.target)", text, re.S):
        blk = m.group(0)
        get = lambda key: (re.search(r"\." + key + r":\s+(\S+)", blk) or [None, "?"])[1]
        name = get("name")
        rows[name] = dict(vgpr=get("vgpr_count"), agpr=get("agpr_count"), sgpr=get("sgpr_count"), sgpr_spill=get("sgpr_spill_count"),
                          vgpr_spill=get("vgpr_spill_count"), scratch_bytes=get("private_segment_fixed_size"),
                          static_lds=get("group_segment_fixed_size"), max_flat_wg=get("max_flat_workgroup_size"))
    # instruction census of every kernel body
    for name in rows:
        m = re.search(r"^" + re.escape(name) + r":[^\n]*\n(.*?)\n\.Lfunc_end\d+:", text, re.S | re.M)
        body = m.group(1) if m else ""
        ins = [l.strip() for l in body.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        rows[name].update(instructions=len(ins), scratch_ins=sum(i.startswith("scratch_") for i in ins),
                          readlane=sum(i.startswith("v_readlane") for i in ins), writelane=sum(i.startswith("v_writelane") for i in ins),
                          barriers=sum(i.startswith("s_barrier") for i in ins), f64=sum("_f64" in i.split()[0] for i in ins),
                          ds=sum(i.startswith("ds_") for i in ins), vmem=sum(i.startswith(("global_", "buffer_", "flat_")) for i in ins))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--keep-asm", default=None)
    args = ap.parse_args()
    tmp = args.keep_asm or tempfile.mkdtemp(prefix="agbnp_asm_")
    os.makedirs(tmp, exist_ok=True)
    allrows = {}
    for src in ("tree_kernels.hip", "pair_kernels.hip", "engine.hip"):
        if not os.path.exists(os.path.join(CSRC, src)):
            continue
        out = os.path.join(tmp, src.replace(".hip", ".s"))
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-o", out, src], cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
        allrows.update(parse(out))
    pretty = demangle(list(allrows))
    cols = ["kernel", "vgpr", "agpr", "sgpr", "sgpr_spill", "vgpr_spill", "scratch_bytes", "static_lds", "instructions", "f64", "ds", "vmem",
            "barriers", "scratch_ins", "readlane", "writelane"]
    lines = []
    for name, r in sorted(allrows.items(), key=lambda kv: pretty[kv[0]]):
        short = re.sub(r"^void agbnp::", "", pretty[name])
        short = re.sub(r"\(.*$", "", short)
        lines.append([short] + [r.get(c, "") for c in cols[1:]])
    w = csv.writer(open(args.out, "w", newline="") if args.out else sys.stdout)
    w.writerow(cols)
    w.writerows(lines)


if __name__ == "__main__":
    main()
