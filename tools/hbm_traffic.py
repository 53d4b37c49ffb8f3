"""HBM traffic per kernel launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs: TCC has 4 slots,
FETCH_SIZE takes 3 and WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots").

  hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced
streaming read (MI355X_MICROARCH.md "HBM"), hence the factor 2 on the read side.  Infinity-Cache hits are counted as
traffic by these counters (they sit on the L2's fabric side).

usage: hbm_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv out.json [key=value ...]
"""
import collections
import csv
import json
import re
import sys

CLASSES = [("linear", r"linear_planes_kernel|linear_planes_grouped_kernel|linear_kernel"), ("grid_aggregate", r"grid_aggregate_kernel|grid_aggregate_pipe_kernel"),
           ("attention", r"attention_rows_kernel|attention_planes_kernel|attention_kernel"), ("transpose_v", r"transpose_v_kernel"),
           ("layernorm", r"layernorm_kernel"), ("split_rows", r"split_rows_kernel"), ("embed", r"cells_embed_kernel|node_embed_kernel"),
           ("heads", r"nav_head_rows_kernel|nav_fuse_kernel"),
           ("grid_project", r"grid_project_kernel"), ("grid_bin", r"grid_bin_sort_kernel")]


def per_class(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for cls, pat in CLASSES:
            if re.search(pat, r["Kernel_Name"]):
                agg[cls].append(float(r["Counter_Value"]))
                break
    return agg


def main():
    fetch, write = per_class(sys.argv[1], "FETCH_SIZE"), per_class(sys.argv[2], "WRITE_SIZE")
    out = {"formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes; counters in KiB, separate --pmc passes",
           "config": dict(kv.split("=", 1) for kv in sys.argv[4:]), "kernels": {}}
    for cls in fetch:
        f = sum(fetch[cls]) / len(fetch[cls])
        w = sum(write[cls]) / len(write[cls]) if write.get(cls) else 0.0
        out["kernels"][cls] = {"launches_fetch_pass": len(fetch[cls]), "launches_write_pass": len(write.get(cls, [])),
                               "FETCH_SIZE_KiB_per_launch": f, "WRITE_SIZE_KiB_per_launch": w,
                               "hbm_bytes_per_launch": (2 * f + w) * 1024}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out["kernels"].items():
        print("%-16s launches %5d  fetch %10.1f KiB  write %10.1f KiB  -> %8.2f MB/launch" % (
            k, v["launches_fetch_pass"], v["FETCH_SIZE_KiB_per_launch"], v["WRITE_SIZE_KiB_per_launch"],
            v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
