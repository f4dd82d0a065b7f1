#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace [+ pmc]) as text:
per-kernel launch count, avg/min/max duration, register/LDS use, and PMC counter means.
usage: tools/rocpd_stats.py <results.db> [more.db ...]"""
import sqlite3
import sys


def table(cur, prefix):
    for (n,) in cur.execute("select name from sqlite_master where type='table'"):
        if n.startswith(prefix):
            return n
    return None


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        cur = db.cursor()
        kd, ks = table(cur, "rocpd_kernel_dispatch"), table(cur, "rocpd_info_kernel_symbol")
        print("# %s" % path)
        q = ("select s.kernel_name, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3,"
             " sum(d.end-d.start)/1e6, max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count),"
             " max(d.group_segment_size), max(d.grid_size_x), max(d.workgroup_size_x)"
             " from %s d join %s s on d.kernel_id=s.id group by s.kernel_name order by 6 desc limit 32" % (kd, ks))
        print("%-64s %6s %10s %10s %10s %10s %5s %5s %5s %7s %8s %5s" %
              ("kernel", "calls", "avg_us", "min_us", "max_us", "total_ms", "vgpr", "agpr", "sgpr", "lds", "grid", "wg"))
        for r in cur.execute(q):
            name = r[0].replace("_ZN5stego", "").replace(".kd", "")[:64]
            print("%-64s %6d %10.2f %10.2f %10.2f %10.2f %5s %5s %5s %7s %8s %5s" % ((name,) + tuple(r[1:])))
        pe, pi = table(cur, "rocpd_pmc_event"), table(cur, "rocpd_info_pmc")
        if pe and pi:
            try:
                q = ("select s.kernel_name, p.name, count(*), avg(e.value) from %s e join %s p on e.pmc_id=p.id "
                     "join %s d on e.event_id=d.event_id join %s s on d.kernel_id=s.id "
                     "group by s.kernel_name, p.name order by s.kernel_name, p.name" % (pe, pi, kd, ks))
                rows = list(cur.execute(q))
                if rows:
                    print("%-64s %-28s %6s %16s" % ("kernel", "counter", "n", "mean/launch"))
                for r in rows:
                    if "stego" not in r[0]:
                        continue
                    print("%-64s %-28s %6d %16.1f" % (r[0].replace("_ZN5stego", "").replace(".kd", "")[:64], r[1], r[2], r[3]))
            except sqlite3.Error as e:
                print("pmc query failed:", e)
        print()


if __name__ == "__main__":
    main()
