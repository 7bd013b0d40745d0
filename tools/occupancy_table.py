"""Per-kernel launch geometry of one profiled run: how far each launch fills the chip.

    python tools/occupancy_table.py <rocprofv3 output dir> <steps profiled> [top]

Reads the *_kernel_trace.csv of `rocprofv3 --kernel-trace` and prints, per (kernel, grid), the launches per step, workgroups, waves per
workgroup, the workgroups one CU can hold (registers: 512 per lane and SIMD shared by the waves there; LDS: 160 KB; 32 waves per CU at most),
"fill" = workgroups / (256 CUs x resident workgroups) - below 1 the launch leaves CUs or wave slots empty, a little above an integer it pays a
nearly empty last round - the average duration and ms per step, sorted by ms per step.  Written for VERDICT r4 item 7 (configs[1]: B0 at batch
16)."""
from __future__ import annotations

import collections
import csv
import glob
import math
import os
import sys


def main(d, steps, top=45):
    f = [p for p in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)][0]
    rows = list(csv.DictReader(open(f)))
    acc = collections.OrderedDict()
    for r in rows:
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        vg = int(r.get("VGPR_Count", 0) or 0) + int(r.get("Accum_VGPR_Count", 0) or 0)
        lds = int(r.get("LDS_Block_Size", 0) or 0)
        key = (r["Kernel_Name"].split("(")[0], grid // wg, wg, vg, lds)
        a = acc.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    out = []
    for (name, nwg, wg, vg, lds), (n, us) in acc.items():
        waves = (wg + 63) // 64
        per_simd = max(1, min(8, 512 // max(vg, 64)))                      # waves one SIMD holds at this register count
        by_reg = (per_simd * 4) // waves if waves <= per_simd * 4 else 0
        by_lds = (160 * 1024) // lds if lds else 99
        res = max(1, min(by_reg, by_lds, 32 // waves))
        out.append((us / steps / 1e3, name, nwg, waves, vg, lds, res, nwg / (256.0 * res), n / steps, us / n))
    out.sort(reverse=True)
    tot = sum(o[0] for o in out)
    print(f"# {os.path.basename(f)}: {len(rows)} dispatches, {steps} steps, {tot:.2f} ms of kernel time per step")
    print(f"{'kernel':58s} {'wgs':>7s} {'wv':>3s} {'vgpr':>4s} {'lds':>6s} {'res':>3s} {'fill':>6s} {'/step':>6s} {'us':>7s} {'ms/step':>7s}")
    for ms, name, nwg, waves, vg, lds, res, fill, per, us in out[:top]:
        print(f"{name[:58]:58s} {nwg:7d} {waves:3d} {vg:4d} {lds:6d} {res:3d} {fill:6.2f} {per:6.1f} {us:7.1f} {ms:7.3f}")
    low = sum(o[0] for o in out if o[7] < 1.0)
    print(f"# launches that do not fill one round of the chip (fill < 1): {low:.2f} ms per step of {tot:.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 45)
