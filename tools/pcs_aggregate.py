"""Aggregate rocprofv3 PC-sampling CSVs (host_trap / stochastic) into a histogram small enough to travel back.
usage: python tools/pcs_aggregate.py <dir> <out.json>"""
import csv, glob, json, os, sys, collections
d, out = sys.argv[1], sys.argv[2]
res = {}
for f in glob.glob(os.path.join(d, "**", "*pc_sampling*.csv"), recursive=True):
    cnt = collections.Counter()
    other = collections.Counter()
    n = 0
    with open(f, newline="") as fh:
        rd = csv.DictReader(fh)
        cols = rd.fieldnames
        head = []
        for row in rd:
            n += 1
            if len(head) < 5: head.append(row)
            cnt[(row.get("Instruction_Comment", ""), row.get("Instruction", ""))] += 1
            for k in cols:
                if k not in ("Sample_Timestamp", "Exec_Mask", "Dispatch_Id", "Instruction", "Instruction_Comment", "Correlation_Id", "Wave_Count"):
                    other[(k, row.get(k))] += 1
    res[os.path.basename(f)] = {"columns": cols, "samples": n, "head": head,
                                "by_instruction": [[k[0], k[1], v] for k, v in cnt.most_common(20000)],
                                "other": [[k[0], k[1], v] for k, v in other.most_common(300)]}
json.dump(res, open(out, "w"))
print({k: v["samples"] for k, v in res.items()})
