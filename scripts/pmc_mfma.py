"""Turn one rocprofv3 --kernel-trace --pmc run (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE) of bench.py into profiles/mfma_utilisation.json: per level-0 coupling-network kernel the
average counter values per launch and the derived MFMA utilisation

    util = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs),   kernel cycles = GRBM_GUI_ACTIVE / 8

(MI355X_MICROARCH.md: MFMA_BUSY counts cycles summed over the SIMDs; rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs).
usage: python scripts/pmc_mfma.py <counter_collection.csv> <out.json>"""
import csv, json, sys
from collections import defaultdict

if __name__ == "__main__":
    path, out = sys.argv[1], sys.argv[2]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for row in csv.DictReader(open(path)):
        k = row.get("Kernel_Name", "")
        if not k.startswith("void k_net_"):
            continue
        c = acc[k][row["Counter_Name"]]
        c[0] += float(row["Counter_Value"]); c[1] += 1
    res = {"simds": 1024, "formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)", "kernels": {}}
    for k, cs in sorted(acc.items()):
        avg = {c: v[0] / max(v[1], 1) for c, v in cs.items()}
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
    json.dump(res, open(out, "w"), indent=1)
    for k, d in res["kernels"].items():
        print(k, {x: (round(y, 4) if isinstance(y, float) else y) for x, y in d.items() if x != "per_launch"})
