"""Per-kernel sums of one rocprofv3 --pmc pass (csv output): python tools/pmc_sum.py <output dir> <COUNTER>[,<COUNTER>...]
Prints, per kernel name and grid size: launches, the counter's mean per launch (raw units: FETCH_SIZE / WRITE_SIZE are KiB; on gfx950
FETCH_SIZE reports HALF of a wide coalesced read stream - MI355X_MICROARCH.md, HBM section - the x2 is applied by the consumer)."""
import collections, csv, glob, re, sys
d, counters = sys.argv[1], sys.argv[2].split(",")
fs = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
if not fs:
    print("no counter_collection.csv under", d); sys.exit(1)
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for row in csv.DictReader(open(fs[0])):
    if row["Counter_Name"] not in counters:
        continue
    key = (re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")[:70], row["Grid_Size"])
    a = acc[key][row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
rows = sorted(acc.items(), key=lambda kv: -max(v[1] for v in kv[1].values()))
for (name, grid), cs in rows[:60]:
    print(f"{name:70s} grid={grid:>8s} " + " ".join(f"{c}: n={v[0]} mean={v[1] / max(v[0], 1):.1f} sum={v[1]:.0f}" for c, v in cs.items()))
