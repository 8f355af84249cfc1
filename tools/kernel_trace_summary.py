"""Per-(kernel symbol, grid size) summary of a rocprofv3 --kernel-trace run (lanes off: CTRL_ADAPTER_LANES=1, so kernel
durations do not overlap), in the spelling bench.py's `per_kernel` / `roofline.kernel` use:

  python tools/kernel_trace_summary.py <rocprof_out_dir> <steps_in_run> <out.csv>

The average duration of a row must agree with `avg_launch_ms` of the same kernel in the bench line (HIP events)."""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import symbol_of, kernel_class  # noqa: E402


def main():
    d, steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    acc = {}
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if kernel_class(row["Kernel_Name"]) is None:
                    continue
                if "Grid_Size" in row:
                    grid = int(row["Grid_Size"])
                else:                       # kernel-trace CSVs carry the grid per dimension (work-items)
                    grid = int(row["Grid_Size_X"]) * int(row.get("Grid_Size_Y", 1) or 1) * int(row.get("Grid_Size_Z", 1) or 1)
                k = (symbol_of(row["Kernel_Name"]), grid)
                a = acc.setdefault(k, [0, 0, 1 << 62, 0])
                ns = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                a[0] += 1; a[1] += ns; a[2] = min(a[2], ns); a[3] = max(a[3], ns)
    with open(out, "w", newline="") as fh:
        wr = csv.writer(fh)
        wr.writerow(["symbol", "grid_work_items", "calls", "calls_per_step", "avg_ms", "min_ms", "max_ms", "ms_per_step"])
        for (sym, grid), a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            wr.writerow([sym, grid, a[0], round(a[0] / steps, 2), round(a[1] / a[0] / 1e6, 5), round(a[2] / 1e6, 5),
                         round(a[3] / 1e6, 5), round(a[1] / steps / 1e6, 4)])
    print("wrote", out, len(acc), "rows")


if __name__ == "__main__":
    main()
