#!/usr/bin/env python3
"""which hardware queues the contexts' streams land on: python tools/queue_probe.py <results.db>"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
t = [r[0] for r in c.execute("select name from sqlite_master where type='table'") if 'kernel_dispatch' in r[0]][0]
rows = c.execute("select queue_id, stream_id, count(*), min(start), max(end) from %s group by queue_id, stream_id" % t).fetchall()
for r in rows:
    print("queue %s stream %s: %d dispatches, span %.2f ms" % (r[0], r[1], r[2], (r[4] - r[3]) / 1e6))
