"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table (markdown).
Usage: python tools/rocpd_stats.py <results.db> [--top N] [--grid] [--match <substring>] [--split-b2b <kernel name substring>] [--window <ns> <ns>]
--split-b2b: for one kernel, average duration of the launches that directly follow another launch of the SAME kernel
(bench.py's roofline pass: 40 launches replayed back to back) and of all others (the launches inside the UNet workload)."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    cur = db.cursor()
    group = "name, grid_x, workgroup_x" if "--grid" in sys.argv else "name"
    where = ""
    if "--window" in sys.argv:          # only dispatches that start inside [a, b] ns (bench.py logs the timed region's CLOCK_MONOTONIC bounds)
        a, b = int(sys.argv[sys.argv.index("--window") + 1]), int(sys.argv[sys.argv.index("--window") + 2])
        n_in = cur.execute(f"select count(*) from kernels where start >= {a} and start <= {b}").fetchone()[0]
        lo, hi = cur.execute("select min(start), max(end) from kernels").fetchone()
        print("window [%d, %d] ns = %.1f ms: %d dispatches inside (trace spans [%d, %d])" % (a, b, (b - a) / 1e6, n_in, lo, hi))
        if n_in:
            where = f"where start >= {a} and start <= {b}"
    rows = cur.execute(
        f"select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
        f"max(accum_vgpr_count), max(lds_size), min(grid_x), max(grid_x), max(workgroup_x) from kernels {where} group by {group} "
        f"order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    print("total kernel time: %.3f ms over %d dispatches, %d distinct kernels" % (total / 1e6, sum(r[1] for r in rows), len(rows)))
    print("| % | total ms | calls | avg us | min us | max us | vgpr | agpr | lds | grid | wg | kernel |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else None      # only kernels whose name contains this
    if match:
        rows = [r for r in rows if match in r[0]]
    for r in rows[:top]:
        name = r[0] if len(r[0]) < 110 else r[0][:107] + "..."
        grid = str(r[9]) if r[9] == r[10] else "%d-%d" % (r[9], r[10])
        print("| %.1f | %.3f | %d | %.1f | %.1f | %.1f | %d | %d | %d | %s | %d | %s |" % (
            100.0 * r[2] / total, r[2] / 1e6, r[1], r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, r[6], r[7], r[8], grid, r[11], name))


def split_b2b(db, sub):
    rows = db.cursor().execute("select name, start, end from kernels order by start").fetchall()
    b2b, mixed = [], []
    prev = None
    for name, start, end in rows:
        if sub in name:
            (b2b if (prev is not None and sub in prev) else mixed).append((end - start) / 1e3)
        prev = name
    def stat(v):
        v = sorted(v)
        return "n=%d avg %.1f us, median %.1f, p10 %.1f, p90 %.1f" % (len(v), sum(v) / len(v), v[len(v) // 2], v[len(v) // 10], v[len(v) * 9 // 10]) if v else "n=0"
    print("\n`%s`: launches that directly follow a launch of the same kernel (back-to-back pass): %s; all other launches (inside the workload): %s"
          % (sub, stat(b2b), stat(mixed)))


if __name__ == "__main__":
    main()
    if "--split-b2b" in sys.argv:
        split_b2b(sqlite3.connect(sys.argv[1]), sys.argv[sys.argv.index("--split-b2b") + 1])
