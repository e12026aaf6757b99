"""Per-kernel PMC summary from a rocprofv3 rocpd database (one counter per run: `--pmc X --kernel-trace`).
usage: rocpd_pmc.py results.db [kernel-substring] [min-avg-us]   -> one JSON line per (kernel, grid)"""
import json, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
t = lambda p: [x for x in tabs if x.startswith(p)][0]
kd, ks, pe, ip = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol"), t("rocpd_pmc_event"), t("rocpd_info_pmc")
rows = cur.execute(f"""select s.display_name, d.grid_size_x / d.workgroup_size_x, d.workgroup_size_x, p.name, e.value, d.end - d.start
                       from {pe} e join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id join {ip} p on e.pmc_id = p.id""").fetchall()
agg = {}
for name, grid, wg, ctr, val, dur in rows:
    if pat not in name:
        continue
    a = agg.setdefault((name.split("(")[0], grid, wg, ctr), [])
    a.append((val, dur / 1e3))
for (name, grid, wg, ctr), v in sorted(agg.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    avg_us = sum(x[1] for x in v) / len(v)
    if avg_us < min_us:
        continue
    vals = [x[0] for x in v]
    print(json.dumps({"kernel": name, "grid": f"{grid}x{wg}", "counter": ctr, "launches": len(v), "avg": sum(vals) / len(vals),
                      "min": min(vals), "max": max(vals), "avg_kernel_us": avg_us}))
