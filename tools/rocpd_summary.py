#!/usr/bin/env python3
"""Dump the kernel summary of a rocprofv3 (rocpd sqlite) result as CSV: name,calls,total_us,avg_us,pct."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
print("kernel,calls,total_us,avg_us,percent")
for name, calls, total, avg, pct in db.execute("select * from top_kernels"):
    print('"%s",%d,%.3f,%.3f,%.2f' % (name, calls, total, avg, pct))
