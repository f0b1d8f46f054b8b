cd /root/repo; python tools/dbg/solve_prefetch_ab.py 2>&1 | tail -4
