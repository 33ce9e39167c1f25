"""Turn two rocprofv3 --pmc runs (FETCH_SIZE; WRITE_SIZE) of bench.py into profiles/roofline_traffic.json.

Collected as MI355X_MICROARCH.md prescribes: separate passes, --kernel-trace only, units of 1 KiB, and on gfx950
FETCH_SIZE of wide coalesced streams reads 1/2 of the bytes (x2 correction); WRITE_SIZE reads exactly.
usage: python scripts/pmc_traffic.py <fetch_counter_csv> <write_counter_csv> <tiles_per_launch> <out.json>
"""
import csv, json, sys

def per_launch(path, counter, kernel_prefix):
    tot, n = 0.0, 0
    for row in csv.DictReader(open(path)):
        if row.get("Counter_Name") == counter and row.get("Kernel_Name", "").startswith(kernel_prefix):
            tot += float(row["Counter_Value"]); n += 1
    return tot / max(n, 1), n

if __name__ == "__main__":
    fcsv, wcsv, tiles, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    pref = sys.argv[5] if len(sys.argv) > 5 else "void k_net_f32<2, 36, 16, 0>"
    key = sys.argv[6] if len(sys.argv) > 6 else "k_net_level0_hbm_bytes_per_launch_per_tile"
    f, nf = per_launch(fcsv, "FETCH_SIZE", pref)
    w, nw = per_launch(wcsv, "WRITE_SIZE", pref)
    hbm = (2.0 * f + w) * 1024.0
    import os
    d = json.load(open(out)) if os.path.exists(out) else {}
    d.setdefault("kernels", {})[pref] = {"launches": [nf, nw], "FETCH_SIZE_KiB_per_launch": f, "WRITE_SIZE_KiB_per_launch": w,
                                         "tiles_per_launch": tiles, "hbm_bytes_per_launch": hbm}
    d["correction"] = "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md HBM section), x1024 B"
    d[key] = hbm / tiles
    json.dump(d, open(out, "w"), indent=1)
    print(open(out).read())
