"""Every dispatch of a rocprofv3 rocpd database as one gzip CSV row (kernel, grid, workgroup, start ns, end ns, and — when the
run collected counters — one column per counter, summed over the database's per-dimension rows of that dispatch), so the
summaries under profiles/ stay reproducible after the multi-megabyte .db is gone.
    python tools/rocpd_dump.py results.db out.csv.gz"""
import csv, gzip, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
t = lambda p: ([x for x in tabs if x.startswith(p)] or [None])[0]
kd, ks, pe, ip = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol"), t("rocpd_pmc_event"), t("rocpd_info_pmc")
scols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
namecol = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else scols[-1])
pmc = {}
names = []
if pe and ip:
    for ev, ctr, val in cur.execute(f"select e.event_id, p.name, e.value from {pe} e join {ip} p on e.pmc_id = p.id"):
        d = pmc.setdefault(ev, {}); d[ctr] = d.get(ctr, 0.0) + val
    names = sorted({c for d in pmc.values() for c in d})
rows = cur.execute(f"select d.event_id, s.{namecol}, d.grid_size_x, d.workgroup_size_x, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
with gzip.open(sys.argv[2], "wt", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "workgroups", "workgroup_size", "start_ns", "end_ns"] + names)
    for ev, name, gx, wx, a, b in rows:
        name = re.sub(r"\s+", " ", str(name))[:120]
        w.writerow([name, gx // max(wx, 1), wx, a, b] + [pmc.get(ev, {}).get(c, "") for c in names])
print(f"{len(rows)} dispatches, counters {names} -> {sys.argv[2]}")
