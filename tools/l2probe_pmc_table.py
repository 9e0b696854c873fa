"""Joins the FETCH_SIZE / WRITE_SIZE passes of `rocprofv3 --pmc ... -- tools/bin/l2probe pmc` with the probe's own configuration lines
(one two_phase_kernel launch per line, same order):   python tools/l2probe_pmc_table.py <dir with l2probe_pmc_*> > table.txt
FETCH_SIZE is doubled (MI355X_MICROARCH.md: it reports half the bytes of a wide coalesced read) and both are in KiB."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
res = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{d}/l2probe_pmc_{C}/*/*_counter_collection.csv")[0]
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "two_phase" in r["Kernel_Name"]:
            k = int(r["Dispatch_Id"])
            acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"])
    res[C] = [acc[k] for k in sorted(acc)]
lines = [l for l in open(f"{d}/l2probe_pmc_FETCH_SIZE.log") if l[:1].isdigit()]
print("# bytes that crossed the L2's fabric side per launch (GB); the algorithm reads 2.147 GB and writes 2.147 GB, a hand-over through memory adds the same again")
print(f"{'#':>3} {'plane':<9} {'group':<7} {'store':<6} {'scratch':<10} {'ahead':<5} {'read GB':>8} {'write GB':>9}")
for i, l in enumerate(lines):
    p = l.split()
    print(f"{p[0]:>3} {p[1]:<9} {p[2]:<7} {p[3]:<6} {p[4]:<10} {p[5]:<5} {res['FETCH_SIZE'][i] * 2048 / 1e9:8.3f} {res['WRITE_SIZE'][i] * 1024 / 1e9:9.3f}")
