#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite DB): per-kernel totals, and the gaps
between consecutive kernels (end of one -> start of the next) inside the busy part of the trace.
  python tools/kernel_stats.py <dir-or-db> [forwards]"""
import glob, os, sqlite3, sys

def main():
  path = sys.argv[1]
  nfwd = int(sys.argv[2]) if len(sys.argv) > 2 else 0
  dbs = [path] if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
  if not dbs:
    raise SystemExit("no rocpd .db under " + path)
  con = sqlite3.connect(dbs[-1])
  rows = con.execute("select name, start, end from kernels order by start").fetchall()
  tot = {}
  for n, s, e in rows:
    t = tot.setdefault(n, [0, 0.0]); t[0] += 1; t[1] += (e - s) / 1e6
  total = sum(v[1] for v in tot.values())
  print("# %d kernel dispatches, total kernel time %.3f ms%s" % (
      len(rows), total, (" (%.3f ms / forward over %d forwards)" % (total / nfwd, nfwd)) if nfwd else ""))
  print("%-110s %7s %12s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
  for n, (c, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-110s %7d %12.3f %10.2f %6.1f" % (n[:110], c, ms, 1e3 * ms / c, 100 * ms / total))
  gaps = [(rows[i + 1][1] - rows[i][2]) / 1e3 for i in range(len(rows) - 1)]
  small = [g for g in gaps if 0 <= g < 200]       # inter-kernel gaps inside a forward (us)
  if small:
    small.sort()
    print("# inter-kernel gaps < 200 us: n=%d  sum %.3f ms  median %.2f us  p90 %.2f us  mean %.2f us" % (
        len(small), sum(small) / 1e3, small[len(small) // 2], small[int(len(small) * 0.9)], sum(small) / len(small)))
    neg = sum(1 for g in gaps if g < 0)
    print("# overlapping dispatches (negative gap): %d" % neg)

if __name__ == "__main__":
  main()
