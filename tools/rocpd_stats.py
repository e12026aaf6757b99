"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) as per-kernel statistics.
    python tools/rocpd_stats.py results.db [> profiles/xxx_kernel_stats.txt]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
scols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
namecol = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else scols[-1])
rows = cur.execute(f"select s.{namecol}, d.end - d.start, d.grid_size_x, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id = s.id").fetchall()
agg = {}
for name, dur, gx, wx in rows:
    name = f"[grid {gx // max(wx, 1)}x{wx}] " + re.sub(r"\s+", " ", str(name))
    a = agg.setdefault(name, [0, 0, 1 << 62, 0])
    a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
tot = sum(a[1] for a in agg.values()) or 1
print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}  kernel")
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{a[0]:7d} {a[1]/1e6:10.3f} {a[1]/a[0]/1e3:10.2f} {a[2]/1e3:10.2f} {a[3]/1e3:10.2f} {100*a[1]/tot:6.2f}  {name[:150]}")
