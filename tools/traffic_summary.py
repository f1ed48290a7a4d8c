"""Turn the four PMC passes of tools/traffic_measure.sh into profiles/traffic_<workload>.json."""
import glob, json, os, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
O, W = sys.argv[1], sys.argv[2]


def counter(name, ctr, like):
    db = glob.glob(os.path.join(O, name, "**", "*.db"), recursive=True)[0]
    con = sqlite3.connect(db)
    # (the deep launch behind every call launch — vlr_deep::vlr_call_kernel, exits at once for every locus of these workloads — is
    # its own dispatch: not part of the per-launch average of the LDS-resident kernel)
    rows = list(con.execute("select sum(value), count(distinct dispatch_id) from counters_collection where counter_name = ? and kernel_name like ? and kernel_name not like '%vlr_deep%'", (ctr, like)))
    return float(rows[0][0]), int(rows[0][1])


cal_r, nr = counter("cal_read", "FETCH_SIZE", "%selftest_stream%")
cal_w, nw = counter("cal_write", "WRITE_SIZE", "%selftest_stream%")
known_r, known_w = (1 << 30) * 4.0, (1 << 30) * 8.0
k_read = known_r / (cal_r * 1024.0 / nr)      # counters are in KB
k_write = known_w / (cal_w * 1024.0 / nw)
kern = "%vlr_realign_kernel%" if W == "realign" else "%vlr_call_kernel%"
f, nf = counter("fetch", "FETCH_SIZE", kern)
w, nwk = counter("write", "WRITE_SIZE", kern)
line = json.loads(open(os.path.join(O, "fetch.out")).read().strip().splitlines()[-1])
read_b, write_b = f * 1024.0 / nf * k_read, w * 1024.0 / nwk * k_write
out = {
    "workload": W, "n_units": int(line["config"]["workload"].split(" loci/GPU")[0].split()[-1]) if W != "realign" else int(line["config"]["workload"].split()[1]),
    "build_id": line["build_id"],
    "command": "tools/traffic_measure.sh %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, --kernel-trace only)" % W,
    "FETCH_SIZE_KB_per_launch": f / nf, "WRITE_SIZE_KB_per_launch": w / nwk,
    "calibration": {"read_4B_per_lane": {"known_bytes": known_r, "counter_KB": cal_r / nr, "factor": k_read},
                    "write_8B_per_lane": {"known_bytes": known_w, "counter_KB": cal_w / nw, "factor": k_write}},
    "read_bytes_per_launch": read_b, "write_bytes_per_launch": write_b, "hbm_bytes_per_launch": read_b + write_b,
    "algorithmic_bytes_per_launch": line["roofline"]["algorithmic_bytes_per_launch"],
    "ratio_to_algorithmic": (read_b + write_b) / line["roofline"]["algorithmic_bytes_per_launch"],
}
path = os.path.join(ROOT, "gpurun_out", "traffic_%s.json" % W)
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
