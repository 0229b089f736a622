"""Per-kernel means of a `rocprofv3 --pmc` run (counter_collection.csv), library kernels only:
python tools/pmc_summary.py <dir> [name substring]  ->  kernel, launches, mean duration, mean of each counter per launch."""
import csv
import glob
import re
import sys

csv.field_size_limit(1 << 30)
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = {}
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            name = r["Kernel_Name"]
            if "at::native" in name or "rocclr" in name or pat not in name:
                continue
            name = re.sub(r"\(anonymous namespace\)::|void ", "", name)
            name = re.sub(r"\(.*", "", name)[:90]
            d = acc.setdefault(name, {"disp": set(), "dur": 0.0, "ctr": {}, "vgpr": r["VGPR_Count"], "agpr": r["Accum_VGPR_Count"], "lds": r["LDS_Block_Size"],
                                      "wg": r["Workgroup_Size"], "grid": r["Grid_Size"]})
            if r["Dispatch_Id"] not in d["disp"]:
                d["disp"].add(r["Dispatch_Id"])
                d["dur"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            d["ctr"][r["Counter_Name"]] = d["ctr"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for name, d in sorted(acc.items(), key=lambda kv: -kv[1]["dur"]):
    n = len(d["disp"])
    print(f"{name}\n    launches {n}  mean {d['dur'] / n / 1e3:.1f} us  grid {d['grid']} wg {d['wg']} vgpr {d['vgpr']} agpr {d['agpr']} lds {d['lds']}")
    for c, v in sorted(d["ctr"].items()):
        print(f"    {c:32s} {v / n:18.0f}")
