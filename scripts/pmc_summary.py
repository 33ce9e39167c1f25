"""Turn rocprofv3 --pmc passes of `bench.py` into the two committed counter summaries bench.py reads for its roofline object.

  python scripts/pmc_summary.py traffic <fetch_counter_collection.csv> <write_counter_collection.csv> <commit> profiles/roofline_traffic.json
  python scripts/pmc_summary.py mfma    <counter_collection.csv> <commit> profiles/mfma_utilisation.json

Collected as MI355X_MICROARCH.md prescribes: separate passes (FETCH_SIZE and WRITE_SIZE do not fit one), --kernel-trace only,
counter units of 1 KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced streams at 64 bytes (x2), WRITE_SIZE
reads 16-byte-per-lane stores exactly.  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs).
"""
import csv
import json
import sys
from collections import defaultdict

LEVEL0 = {   # config B, level 0 (32x32x4 tensors, n_filters 512): the kernel each arithmetic runs there
    "k_net_f32": "void k_net_f32<2, 36, 16, 0, false>(NetArgs)",
    # round 4: the headline launch is the co-resident form k_net_h3c (MODE | 16 = the coupling fused in); the short names stay -- bench.py
    # reads them -- and take the one-workgroup-per-CU kernel where a trace (GLOWK_CO_OFF=1) holds that one instead
    "k_net_h3s": ["void k_net_h3c<2, 36, 16, 16, false>(NetArgs)", "void k_net_h3s<2, 36, 16, 16, 2, false>(NetArgs)"],
    "k_net_h3s_two_term": ["void k_net_h3c<2, 36, 16, 19, false>(NetArgs)", "void k_net_h3s<2, 36, 16, 19, 2, false>(NetArgs)"],
    "k_net_h3s_one_per_cu": "void k_net_h3s<2, 36, 16, 16, 2, false>(NetArgs)",
    "k_net_h3s_unfused": "void k_net_h3s<2, 36, 16, 0, 2, false>(NetArgs)",      # (a GLOWK_NO_FUSE=1 pass, when the csv holds one)
    "k_net_h3s_two_term_unfused": "void k_net_h3s<2, 36, 16, 3, 2, false>(NetArgs)",
}


def per_kernel(path):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for row in csv.DictReader(open(path)):
        k = row.get("Kernel_Name", "")
        if k.startswith("void k_net_") or "k_wgrad" in k or k.startswith("void k_couple"):
            c = acc[k][row["Counter_Name"]]
            c[0] += float(row["Counter_Value"])
            c[1] += 1
    return {k: {c: (v[0] / max(v[1], 1), v[1]) for c, v in cs.items()} for k, cs in acc.items()}


def main():
    mode = sys.argv[1]
    if mode == "traffic":
        fcsv, wcsv, commit, out = sys.argv[2:6]
        f, w = per_kernel(fcsv), per_kernel(wcsv)
        res = {"build": commit, "tiles_per_launch": 1024,
               "recipe": "two rocprofv3 --kernel-trace --pmc passes of `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-shapes` "
                         "(FETCH_SIZE; WRITE_SIZE), per-launch averages; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 correction, MI355X_MICROARCH.md)",
               "hbm_bytes_per_level0_launch": {}, "kernels": {}}
        for k in sorted(set(f) | set(w)):
            fe, nf = f.get(k, {}).get("FETCH_SIZE", (0.0, 0))
            wr, nw = w.get(k, {}).get("WRITE_SIZE", (0.0, 0))
            res["kernels"][k] = {"launches": [nf, nw], "FETCH_SIZE_KiB_per_launch": fe, "WRITE_SIZE_KiB_per_launch": wr,
                                 "hbm_bytes_per_launch": (2.0 * fe + wr) * 1024.0}
        for short, names in LEVEL0.items():
            for name in ([names] if isinstance(names, str) else names):
                if name in res["kernels"]:
                    res["hbm_bytes_per_level0_launch"][short] = res["kernels"][name]["hbm_bytes_per_launch"]
                    res.setdefault("level0_kernel_names", {})[short] = name
                    break
    else:
        path, commit, out = sys.argv[2:5]
        res = {"build": commit, "simds": 1024, "formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)", "level0": {}, "kernels": {}}
        for k, cs in sorted(per_kernel(path).items()):
            avg = {c: v[0] for c, v in cs.items()}
            d = {"launches": max(v[1] for v in cs.values()), "per_launch": avg}
            if avg.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
                cyc = avg["GRBM_GUI_ACTIVE"] / 8.0
                d["kernel_cycles"] = cyc
                d["mfma_utilisation"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
            if avg.get("SQ_WAVE_CYCLES"):
                for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                    if c in avg:
                        d[c + "_share_of_wave_cycles"] = avg[c] / avg["SQ_WAVE_CYCLES"]
            res["kernels"][k] = d
        for short, names in LEVEL0.items():
            for name in ([names] if isinstance(names, str) else names):
                if name in res["kernels"]:
                    res["level0"][short] = dict({x: y for x, y in res["kernels"][name].items() if x != "per_launch"}, kernel=name)
                    break
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "kernels"}, indent=1))


if __name__ == "__main__":
    main()
