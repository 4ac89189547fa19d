"""Condense the rocprofv3 output of tools/profile.sh into one text summary (for profiles/)."""
import collections
import csv
import glob
import os
import sys


def main(d, kernel_filter="vrc_k_raycast", variant="true, false, false"):
    out = []
    for f in glob.glob(os.path.join(d, "trace", "*", "*_kernel_stats.csv")):
        out.append("== kernel stats (rocprofv3 --kernel-trace --stats)")
        for r in csv.DictReader(open(f)):
            out.append("%-60s calls=%s avg_ns=%s min_ns=%s max_ns=%s pct=%s" % (
                r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"]))
    out.append("== PMC per dispatch of kernels matching %r (mean over dispatches; separate passes)" % kernel_filter)
    for p in sorted(glob.glob(os.path.join(d, "pmc_*"))):
        if not os.path.isdir(p):
            continue
        for f in glob.glob(os.path.join(p, "*", "*_counter_collection.csv")):
            agg = collections.defaultdict(list)
            meta = None
            for r in csv.DictReader(open(f)):
                if kernel_filter in r["Kernel_Name"] and variant in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    meta = r
            for k, v in sorted(agg.items()):
                out.append("%-12s %-40s n=%d mean=%.6g" % (os.path.basename(p), k, len(v), sum(v) / len(v)))
            if meta and os.path.basename(p) == "pmc_sq1":
                out.append("             VGPR_Count=%s SGPR_Count=%s LDS_Block_Size=%s Workgroup_Size=%s Grid_Size=%s" % (
                    meta["VGPR_Count"], meta["SGPR_Count"], meta["LDS_Block_Size"], meta["Workgroup_Size"], meta["Grid_Size"]))
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:4]))
