"""Summarise a rocprofv3 --kernel-trace --stats rocpd database (SQLite) as a text table for profiles/."""
import sqlite3
import sys


def main(db_path, title):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall() \
        if "total_calls" in [r[1] for r in db.execute("pragma table_info(top_kernels)")] else None
    if rows is None:
        cols = [r[1] for r in db.execute("pragma table_info(top_kernels)")]
        rows = db.execute("select * from top_kernels").fetchall()
        print("# columns:", cols)
    print("# " + title)
    print("%-100s %8s %12s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in rows:
        print("%-100s %8d %12.0f %10.1f %7.2f" % (str(name)[:100], calls, total, avg, pct))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else sys.argv[1])
