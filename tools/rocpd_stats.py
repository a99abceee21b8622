"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table (markdown).
Usage: python tools/rocpd_stats.py <results.db> [--top N] [--grid]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    cur = db.cursor()
    group = "name, grid_x, workgroup_x" if "--grid" in sys.argv else "name"
    rows = cur.execute(
        f"select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
        f"max(accum_vgpr_count), max(lds_size), min(grid_x), max(grid_x), max(workgroup_x) from kernels group by {group} "
        f"order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    print("total kernel time: %.3f ms over %d dispatches, %d distinct kernels" % (total / 1e6, sum(r[1] for r in rows), len(rows)))
    print("| % | total ms | calls | avg us | min us | max us | vgpr | agpr | lds | grid | wg | kernel |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows[:top]:
        name = r[0] if len(r[0]) < 110 else r[0][:107] + "..."
        grid = str(r[9]) if r[9] == r[10] else "%d-%d" % (r[9], r[10])
        print("| %.1f | %.3f | %d | %.1f | %.1f | %.1f | %d | %d | %d | %s | %d | %s |" % (
            100.0 * r[2] / total, r[2] / 1e6, r[1], r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, r[6], r[7], r[8], grid, r[11], name))


if __name__ == "__main__":
    main()
